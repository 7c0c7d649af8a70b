#!/bin/bash
# round 6, call h: head decode as one wavefront per cell
. scripts/ab_lib.sh r06h
ab_tests tests/test_detect_gpu.py tests/test_detector_chain_gpu.py tests/test_mot_gpu.py
ab_bench 4 --steps 300 --warmup 10 -- "cell:" "record:FASTMOT_DECODE_PATH=1"
timeout 300 python scripts/trace_pipeline.py --show 0 > $O/pipeline_trace.txt 2> $O/pipeline_trace.err; grep -E "det: |durations|period" $O/pipeline_trace.txt
