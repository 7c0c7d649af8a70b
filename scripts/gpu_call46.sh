#!/bin/bash
export TMPDIR=/tmp FASTMOT_RANDOM_WEIGHTS=1
O=gpurun_out/c46; mkdir -p $O
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-variants > $O/$tag.json 2> $O/$tag.err; python - <<P
import json
try:
    d=json.loads(open('$O/$tag.json').read().strip().splitlines()[-1]); print('$tag', d['value'], 'det ms', d['roofline']['net_ms_per_frame'])
except Exception as e: print('$tag FAILED', e)
P
grep "stage ms" $O/$tag.err | tail -1; }
for i in 1 2; do
run split2_$i A=1
run split1_$i FASTMOT_EXT_SPLIT=1
run split3_$i FASTMOT_EXT_SPLIT=3
run split4_$i FASTMOT_EXT_SPLIT=4
done
PROFILE_H2D=1 PROFILE_PREFETCH=1 FASTMOT_FLOW_TIMING_VERBOSE=1 timeout 300 python scripts/profile_step.py > $O/tl.txt 2>&1
grep -E "ms/step|flow_predict stages" $O/tl.txt | tail -2; grep -E "^(mot|trk|ctx|ext|det)\." $O/tl.txt | head -22
