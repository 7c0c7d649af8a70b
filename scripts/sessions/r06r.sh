#!/bin/bash
# round 6, call r: config[4] A/B of this round's switches; ReID in chunks of 25 (MALL footprint hypothesis) at config[1]
. scripts/ab_lib.sh r06r
ab_bench 2 --config 4 --steps 60 --warmup 5 -- "new:" "unfused:FASTMOT_FUSED_INPUT=0 FASTMOT_STEM2=0" "nocasc:FASTMOT_HOST_CASCADE=0" "r05like:FASTMOT_FUSED_INPUT=0 FASTMOT_STEM2=0 FASTMOT_HOST_CASCADE=0"
ab_bench 3 --steps 300 --warmup 10 -- "b64:" "b25:FASTMOT_BENCH_REID_BATCH=25" "b32:FASTMOT_BENCH_REID_BATCH=32"
