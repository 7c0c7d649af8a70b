#!/bin/bash
# round 6, call z4: fused OSNet tail, biases ahead of the prefetches, two-pixel items at the last levels
. scripts/ab_lib.sh r06z4
ab_tests tests/test_conv_gpu.py tests/test_fullsize_gpu.py -k "osnet"
FASTMOT_LIB_PATH=$R/fastmot_amd/libfastmot_hip_timing.so python scripts/ost_timing.py 50 2>&1 | grep -v SEEDED
for v in "tail:" "layers:FASTMOT_OSTAIL=0"; do ab_trace_net "$v" 1 50 44 "ostail|head_kernel"; done
ab_bench 3 --steps 300 --warmup 10 -- "tail:" "layers:FASTMOT_OSTAIL=0"
