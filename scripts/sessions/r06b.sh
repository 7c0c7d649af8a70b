#!/bin/bash
# round 6, call b: the one-call host cascade behind an early pairwise launch -- parity, then A/B
. scripts/ab_lib.sh r06b
ab_tests tests/test_tracker_gpu.py tests/test_assoc_gpu.py tests/test_mot_gpu.py tests/test_e2e_parity_gpu.py tests/test_mot_multiclass_gpu.py tests/test_gallery_rccl_gpu.py tests/test_two_process_gpu.py tests/test_app_gpu.py tests/test_kalman_gpu.py
timeout 300 python scripts/trace_pipeline.py --show 2 > $O/pipeline_trace.txt 2> $O/pipeline_trace.err; head -50 $O/pipeline_trace.txt; tail -3 $O/pipeline_trace.err
ab_bench 3 --steps 300 --warmup 10 -- "new:" "old:FASTMOT_HOST_CASCADE=0"
