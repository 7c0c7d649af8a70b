#!/bin/bash
# round 6, call y: head decode with four class logits per load
. scripts/ab_lib.sh r06y
ab_tests tests/test_detect_gpu.py tests/test_detector_chain_gpu.py
OLD="FASTMOT_LIB_PATH=$R/fastmot_amd/libfastmot_hip_olddecode.so"
for v in "" "$OLD" "" "$OLD"; do env $v A=1 timeout 300 python scripts/trace_pipeline.py --show 0 2>/dev/null | grep -E "^# config|det: decode|period"; done
ab_bench 4 --steps 300 --warmup 10 -- "new:" "old:$OLD"
