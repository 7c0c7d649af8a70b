#!/bin/bash
export TMPDIR=/tmp FASTMOT_RANDOM_WEIGHTS=1
O=gpurun_out/c37; mkdir -p $O
timeout 900 python -m pytest tests/test_e2e_parity_gpu.py tests/test_mot_gpu.py tests/test_flow_gpu.py tests/test_detect_gpu.py -q -m gpu -x 2>&1 | tail -4 > $O/pytest.txt; tail -2 $O/pytest.txt
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-variants > $O/$tag.json 2> $O/$tag.err; python - <<P
import json
d=json.loads(open('$O/$tag.json').read().strip().splitlines()[-1]); print('$tag', d['value'], 'det ms', d['roofline']['net_ms_per_frame'])
P
grep "stage ms" $O/$tag.err | tail -1; }
for i in 1 2; do
run new_$i A=1
run memcpy_$i FASTMOT_UPLOAD_KERNEL=0
run nopyr_$i FASTMOT_PYR_STREAM=0
done
PROFILE_H2D=1 PROFILE_PREFETCH=1 FASTMOT_FLOW_TIMING_VERBOSE=1 timeout 300 python scripts/profile_step.py > $O/tl.txt 2>&1
grep -E "ms/step|flow_predict stages" $O/tl.txt | tail -2; grep -E "sub-stages" $O/tl.txt | tail -1; grep -E "^(mot._step|trk.compute_flow|ctx.flow_predict|ext.postprocess|trk.update|ctx.frame_upload_next|ctx.detect_async_next|ext.extract_async|trk.apply_kalman|det.postprocess)" $O/tl.txt
