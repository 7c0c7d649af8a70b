#!/bin/bash
# round 6, call j: stem + first stride-2 conv as one launch (FM_OP_STEM2)
. scripts/ab_lib.sh r06j
ab_tests tests/test_conv_gpu.py tests/test_fullsize_gpu.py tests/test_detect_gpu.py tests/test_detector_chain_gpu.py tests/test_darknet.py tests/test_scaled_yolov4.py tests/test_onnx_reader.py
ab_layers "stem2:" YOLOv4_608; head -8 $O/layers_YOLOv4_608_stem2.txt | cut -c1-150
ab_layers "two:FASTMOT_STEM2=0" YOLOv4_608; head -6 $O/layers_YOLOv4_608_two.txt | cut -c1-150
ab_bench 4 --steps 300 --warmup 10 -- "stem2:" "two:FASTMOT_STEM2=0"
ab_tests tests/test_mot_gpu.py tests/test_e2e_parity_gpu.py
