#!/bin/bash
# round 6, call w: normalisation table in the fused ReID stem
. scripts/ab_lib.sh r06w
ab_tests tests/test_detect_gpu.py tests/test_fullsize_gpu.py tests/test_e2e_parity_gpu.py
for v in 1 2; do timeout 300 python scripts/trace_pipeline.py --show 0 2>/dev/null | grep -E "^# config|reid: begins|reid: crops done|reid: crop"; done
ab_bench 3 --steps 300 --warmup 10 -- "new:"
