#!/bin/bash
# round 6, call za: residual units at 152^2 / 76^2 with the taps of a tile split over two waves
. scripts/ab_lib.sh r06za
ab_tests tests/test_conv_gpu.py tests/test_fullsize_gpu.py tests/test_detect_gpu.py -k "not osnet"
for v in "split:" "one:FASTMOT_RB_VARIANT=0"; do ab_layers "$v" YOLOv4_608; done
ab_bench 4 --steps 300 --warmup 10 -- "split:" "one:FASTMOT_RB_VARIANT=0"
