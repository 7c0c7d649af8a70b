#!/bin/bash
# round 6, call g: stem convolutions that compute their input pixels themselves (detector preprocess / ReID crops fused)
. scripts/ab_lib.sh r06g
ab_tests tests/test_detect_gpu.py tests/test_detector_chain_gpu.py tests/test_fullsize_gpu.py tests/test_mot_gpu.py tests/test_e2e_parity_gpu.py
ab_trace_net "fused:" 1 50 40 "stem_conv"
ab_bench 4 --steps 300 --warmup 10 -- "fused:" "unfused:FASTMOT_FUSED_INPUT=0"
timeout 300 python scripts/trace_pipeline.py --show 0 > $O/pipeline_trace.txt 2> $O/pipeline_trace.err; grep -E "det: |reid: |durations|period" $O/pipeline_trace.txt
