#!/bin/bash
# round 6, call v: LK kernel with the level-independent template terms of all levels up front
. scripts/ab_lib.sh r06v
ab_tests tests/test_flow_gpu.py tests/test_e2e_parity_gpu.py tests/test_mot_gpu.py tests/test_fullsize_gpu.py
for v in 1 0 1 0; do FASTMOT_LK_PRE=$v timeout 300 python scripts/trace_pipeline.py --show 0 2>/dev/null | grep -E "^# config|lk kernel"; done
ab_bench 4 --steps 300 --warmup 10 -- "pre:" "old:FASTMOT_LK_PRE=0"
ab_bench 2 --config 4 --steps 60 --warmup 5 -- "pre:" "old:FASTMOT_LK_PRE=0"
ab_bench 2 --config 2 --steps 300 --warmup 10 -- "pre:" "old:FASTMOT_LK_PRE=0"
