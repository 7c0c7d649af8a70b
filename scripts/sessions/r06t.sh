#!/bin/bash
# round 6, call t: the stem pair takes the first CSP stage's merged pointwise conv as a third stage
. scripts/ab_lib.sh r06t
ab_tests tests/test_conv_gpu.py tests/test_fullsize_gpu.py tests/test_detect_gpu.py tests/test_darknet.py tests/test_scaled_yolov4.py tests/test_onnx_reader.py
ab_layers "stem3:" YOLOv4_608; head -8 $O/layers_YOLOv4_608_stem3.txt | tail -5 | cut -c1-150
ab_bench 4 --steps 300 --warmup 10 -- "stem3:" "nofuse:FASTMOT_STEM2=0 FASTMOT_PAIR11=0"
ab_tests tests/test_mot_gpu.py tests/test_e2e_parity_gpu.py tests/test_detector_chain_gpu.py
