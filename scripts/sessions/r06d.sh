#!/bin/bash
# round 6, call d: the early pairwise launch on the ReID stream (no cross-stream wait) -- parity, A/B, trace
. scripts/ab_lib.sh r06d
ab_tests tests/test_tracker_gpu.py tests/test_assoc_gpu.py tests/test_mot_gpu.py tests/test_e2e_parity_gpu.py tests/test_mot_multiclass_gpu.py tests/test_gallery_rccl_gpu.py tests/test_two_process_gpu.py tests/test_app_gpu.py
ab_bench 4 --steps 300 --warmup 10 -- "new:" "old:FASTMOT_HOST_CASCADE=0"
timeout 300 python scripts/trace_pipeline.py --show 2 > $O/pipeline_trace.txt 2> $O/pipeline_trace.err; head -52 $O/pipeline_trace.txt
timeout 300 python bench.py --steps 300 --warmup 10 --no-cpu-baseline > $O/bench_variants.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/bench_variants.json')); print('value', d['value'], 'seq', d.get('sequential_fps'), d.get('variants'))"
FASTMOT_HOST_CASCADE=0 timeout 300 python bench.py --steps 300 --warmup 10 --no-cpu-baseline > $O/bench_variants_old.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/bench_variants_old.json')); print('old: value', d['value'], 'seq', d.get('sequential_fps'), d.get('variants'))"
