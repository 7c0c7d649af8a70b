#!/bin/bash
# round 6, call z: OSNet x0.25's last stage + head as one launch (FM_OP_OSTAIL, ostail.hip)
. scripts/ab_lib.sh r06z
ab_tests tests/test_conv_gpu.py -k "osnet"
ab_tests tests/test_fullsize_gpu.py -k "osnet or feature_extractor"
ab_tests tests/test_detect_gpu.py tests/test_torchreid_loader.py
for v in "tail:" "layers:FASTMOT_OSTAIL=0"; do ab_trace_net "$v" 1 50 44 "ostail|head_kernel"; ab_trace_net "$v" 1 300 0; done
ab_bench 3 --steps 300 --warmup 10 -- "tail:" "layers:FASTMOT_OSTAIL=0"
