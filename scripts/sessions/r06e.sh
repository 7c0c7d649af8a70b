#!/bin/bash
. scripts/ab_lib.sh r06e
ab_tests tests/test_mot_gpu.py tests/test_e2e_parity_gpu.py tests/test_detect_gpu.py
timeout 600 python scripts/interference.py 2> $O/interference.err | tee $O/interference.txt; tail -3 $O/interference.err
ab_bench 2 --steps 300 --warmup 10 -- "new:" "old:FASTMOT_HOST_CASCADE=0"
timeout 300 python scripts/trace_pipeline.py --show 0 > $O/pipeline_trace.txt 2> $O/pipeline_trace.err; grep -E "reid: ends|embeddings collected|pairwise|step ends|kalman update" $O/pipeline_trace.txt
