#!/bin/bash
# round 6, call x: embeddings reach page-locked memory from the head layer itself (no export launch); buffer growth drops graphs
. scripts/ab_lib.sh r06x
ab_tests tests/test_detect_gpu.py tests/test_fullsize_gpu.py tests/test_e2e_parity_gpu.py tests/test_mot_gpu.py tests/test_mot_multiclass_gpu.py tests/test_gallery_rccl_gpu.py tests/test_two_process_gpu.py tests/test_torchreid_loader.py tests/test_onnx_reader.py
timeout 300 python scripts/trace_pipeline.py --show 0 2>/dev/null | grep -E "^# config|reid: |embeddings collected|step ends"
ab_bench 3 --steps 300 --warmup 10 -- "new:"
ab_bench 1 --config 4 --steps 60 --warmup 5 -- "new:"
