#!/bin/bash
# spatial partition experiment: CU masks per stream (bits are CU indices of the queue mask)
O=gpurun_out/c23; mkdir -p $O
export TMPDIR=/tmp
run() { tag=$1; shift; env "$@" timeout 200 python bench.py --steps 300 --warmup 30 --no-variants --no-cpu-baseline > $O/$tag.json 2> $O/$tag.err; python - <<P
import json
try:
    d=json.loads(open('$O/$tag.json').read().strip().splitlines()[-1]); print('$tag', d['value'], 'det net ms', d['roofline']['net_ms_per_frame'])
except Exception as e: print('$tag failed', e)
P
grep "stage ms" $O/$tag.err | tail -1; }
run base A=1
run det160_rest96 FASTMOT_CU_MASK_DET=0-160 FASTMOT_CU_MASK_EXT=160-256 FASTMOT_CU_MASK_FLOW=160-256
run det128_rest128 FASTMOT_CU_MASK_DET=0-128 FASTMOT_CU_MASK_EXT=128-256 FASTMOT_CU_MASK_FLOW=128-256
run det128_only FASTMOT_CU_MASK_DET=0-128
run det160_only FASTMOT_CU_MASK_DET=0-160
run det192_only FASTMOT_CU_MASK_DET=0-192
run det128_ext64_flow64 FASTMOT_CU_MASK_DET=0-128 FASTMOT_CU_MASK_EXT=128-192 FASTMOT_CU_MASK_FLOW=192-256 FASTMOT_LK_LDS=0
run base2 A=1
