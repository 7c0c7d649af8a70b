#!/bin/bash
# round 6, call a: A/B of 027c1d9 (litechain regions clipped to the image) + this round's starting tables
. scripts/ab_lib.sh r06a
ab_tests tests/test_conv_gpu.py tests/test_fullsize_gpu.py
OLD="FASTMOT_LIB_PATH=$R/fastmot_amd/libfastmot_hip_oldlch.so"
for i in 1 2; do ab_trace_net "new$i:" 1 50 40 litechain; ab_trace_net "old$i:$OLD" 1 50 40 litechain; done
ab_trace_net "new:" 1 300 40 litechain; ab_trace_net "old:$OLD" 1 300 40 litechain
ab_bench 3 --steps 300 --warmup 10 -- "new:" "old:$OLD"
ab_layers "base:" YOLOv4_608
timeout 300 python scripts/trace_pipeline.py --show 3 > $O/pipeline_trace.txt 2> /dev/null; head -60 $O/pipeline_trace.txt
