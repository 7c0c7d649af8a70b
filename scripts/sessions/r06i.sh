#!/bin/bash
# round 6, call i: detector pass (fused stem + layers + head decode) as one captured graph
. scripts/ab_lib.sh r06i
ab_tests tests/test_detect_gpu.py tests/test_detector_chain_gpu.py tests/test_mot_gpu.py tests/test_e2e_parity_gpu.py tests/test_two_process_gpu.py tests/test_app_gpu.py
ab_bench 4 --steps 300 --warmup 10 -- "graph:" "eager:FASTMOT_PASS_GRAPH=0"
timeout 300 python scripts/trace_pipeline.py --show 0 > $O/pipeline_trace.txt 2> $O/pipeline_trace.err; grep -E "det: |durations|period" $O/pipeline_trace.txt
ab_bench 1 --config 4 --steps 60 --warmup 5 -- "graph:" "eager:FASTMOT_PASS_GRAPH=0"
ab_bench 1 --config 2 --steps 300 --warmup 10 -- "graph:" "eager:FASTMOT_PASS_GRAPH=0"
