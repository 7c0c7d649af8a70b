#!/bin/bash
# round 6, call z2: fused OSNet tail with phase-ahead weight requests
. scripts/ab_lib.sh r06z2
ab_tests tests/test_conv_gpu.py tests/test_fullsize_gpu.py -k "osnet"
for v in "tail:" "layers:FASTMOT_OSTAIL=0"; do ab_trace_net "$v" 1 50 44 "ostail|head_kernel"; done
ab_bench 3 --steps 300 --warmup 10 -- "tail:" "layers:FASTMOT_OSTAIL=0"
