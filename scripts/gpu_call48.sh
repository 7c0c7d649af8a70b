#!/bin/bash
export TMPDIR=/tmp FASTMOT_RANDOM_WEIGHTS=1
O=gpurun_out/c48; mkdir -p $O
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-variants > $O/$tag.json 2> $O/$tag.err; python - <<P
import json
try:
    d=json.loads(open('$O/$tag.json').read().strip().splitlines()[-1]); print('$tag', d['value'], 'det ms', d['roofline']['net_ms_per_frame'])
except Exception as e: print('$tag FAILED', e)
P
grep "stage ms" $O/$tag.err | tail -1; }
for i in 1 2; do
run default_$i A=1
run hwq8_$i GPU_MAX_HW_QUEUES=8
run hwq16_$i GPU_MAX_HW_QUEUES=16
run hwq2_$i GPU_MAX_HW_QUEUES=2
done
run split3_hwq16 GPU_MAX_HW_QUEUES=16 FASTMOT_EXT_SPLIT=3
