#!/bin/bash
export TMPDIR=/tmp FASTMOT_RANDOM_WEIGHTS=1
O=gpurun_out/c45; mkdir -p $O
timeout 900 python -m pytest tests/test_e2e_parity_gpu.py tests/test_mot_gpu.py tests/test_mot_multiclass_gpu.py tests/test_tracker_gpu.py tests/test_flow_gpu.py tests/test_gallery_rccl_gpu.py -q -m gpu -x 2>&1 | tail -8 > $O/pytest.txt; tail -6 $O/pytest.txt
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-variants > $O/$tag.json 2> $O/$tag.err; python - <<P
import json
try:
    d=json.loads(open('$O/$tag.json').read().strip().splitlines()[-1]); print('$tag', d['value'], 'det ms', d['roofline']['net_ms_per_frame'])
except Exception as e: print('$tag FAILED', e)
P
grep "stage ms" $O/$tag.err | tail -1; }
for i in 1 2 3; do
run native_$i A=1
run pythread_$i FASTMOT_NATIVE_FLOW=0
done
