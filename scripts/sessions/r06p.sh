#!/bin/bash
# round 6, call p: (1) parity after the event elisions / zero-copy boxes; (2) experiment: delay the ReID network inside the step
. scripts/ab_lib.sh r06p
ab_tests tests/test_detect_gpu.py tests/test_mot_gpu.py tests/test_e2e_parity_gpu.py tests/test_fullsize_gpu.py
ab_bench 2 --steps 300 --warmup 10 -- "d0:" "d100:FASTMOT_REID_DELAY_US=100" "d200:FASTMOT_REID_DELAY_US=200" "d300:FASTMOT_REID_DELAY_US=300" "d400:FASTMOT_REID_DELAY_US=400"
