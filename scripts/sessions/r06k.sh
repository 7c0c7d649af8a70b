#!/bin/bash
# round 6, call k: 8-byte pixel-pair loads + unrolled patch fill in the stems; stem pair with 4 workgroups per CU
. scripts/ab_lib.sh r06k
ab_tests tests/test_conv_gpu.py tests/test_fullsize_gpu.py tests/test_detect_gpu.py tests/test_detector_chain_gpu.py
ab_layers "stem2:" YOLOv4_608; head -4 $O/layers_YOLOv4_608_stem2.txt | tail -1 | cut -c1-150
ab_trace_net "osnet:" 1 50 40 "stem_conv"
for v in 1 0 1 0; do FASTMOT_STEM2=$v timeout 300 python scripts/trace_pipeline.py --show 0 2>/dev/null | grep -E "^# config|first layer incl|det: network  |reid: crop|period"; done
ab_bench 4 --steps 300 --warmup 10 -- "stem2:" "two:FASTMOT_STEM2=0" "unfused:FASTMOT_FUSED_INPUT=0 FASTMOT_STEM2=0"
ab_tests tests/test_mot_gpu.py tests/test_e2e_parity_gpu.py
