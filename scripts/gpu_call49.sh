#!/bin/bash
export TMPDIR=/tmp FASTMOT_RANDOM_WEIGHTS=1
O=gpurun_out/c49; mkdir -p $O
timeout 900 python -m pytest tests/test_flow_gpu.py tests/test_e2e_parity_gpu.py tests/test_fullsize_gpu.py tests/test_mot_gpu.py -q -m gpu -x 2>&1 | tail -3 > $O/pytest.txt; tail -2 $O/pytest.txt
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-variants > $O/$tag.json 2> $O/$tag.err; python - <<P
import json
try:
    d=json.loads(open('$O/$tag.json').read().strip().splitlines()[-1]); print('$tag', d['value'], 'det ms', d['roofline']['net_ms_per_frame'])
except Exception as e: print('$tag FAILED', e)
P
}
for i in 1 2 3; do
run side_$i A=1
run inline_$i FASTMOT_BG_STREAM=0
done
FASTMOT_FLOW_TIMING_VERBOSE=1 timeout 200 python scripts/profile_step.py 2>&1 | grep -E "flow_predict stages|sub-stages" | tail -2
FASTMOT_BG_STREAM=0 FASTMOT_FLOW_TIMING_VERBOSE=1 timeout 200 python scripts/profile_step.py 2>&1 | grep -E "flow_predict stages|sub-stages" | tail -2
