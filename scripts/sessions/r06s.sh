#!/bin/bash
# round 6, call s: two pointwise convs around a concat as one launch (FM_OP_PAIR11)
. scripts/ab_lib.sh r06s
ab_tests tests/test_conv_gpu.py tests/test_fullsize_gpu.py tests/test_darknet.py tests/test_scaled_yolov4.py tests/test_onnx_reader.py
ab_layers "pair:" YOLOv4_608; head -14 $O/layers_YOLOv4_608_pair.txt | tail -11 | cut -c1-150
ab_layers "nopair:FASTMOT_PAIR11=0" YOLOv4_608
ab_bench 4 --steps 300 --warmup 10 -- "pair:" "nopair:FASTMOT_PAIR11=0"
ab_tests tests/test_mot_gpu.py tests/test_e2e_parity_gpu.py tests/test_detector_chain_gpu.py
