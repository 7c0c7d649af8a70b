#!/bin/bash
# round 6, call c: r05's schedule patch (preprocess of the prefetched frame on its own stream, head decode on the post stream)
# re-measured now that the main thread's chain is 70 us shorter (host cascade)
. scripts/ab_lib.sh r06c
ab_tests tests/test_detect_gpu.py tests/test_detector_chain_gpu.py tests/test_mot_gpu.py tests/test_e2e_parity_gpu.py tests/test_two_process_gpu.py tests/test_fullsize_gpu.py
ab_bench 4 --steps 300 --warmup 10 -- "p1d1:FASTMOT_PRE_AHEAD=1 FASTMOT_DECODE_OFF=1" "p0d0:FASTMOT_PRE_AHEAD=0 FASTMOT_DECODE_OFF=0" "p1d0:FASTMOT_PRE_AHEAD=1 FASTMOT_DECODE_OFF=0" "p0d1:FASTMOT_PRE_AHEAD=0 FASTMOT_DECODE_OFF=1" "p1d1nocasc:FASTMOT_HOST_CASCADE=0"
timeout 300 python scripts/trace_pipeline.py --show 2 > $O/pipeline_trace.txt 2> $O/pipeline_trace.err; head -56 $O/pipeline_trace.txt
