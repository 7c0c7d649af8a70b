#!/bin/bash
export TMPDIR=/tmp FASTMOT_RANDOM_WEIGHTS=1
R=$GRAFT_REPO_ROOT; [ -n "$R" ] || R=$(pwd)
O=$R/gpurun_out/r05u; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_detect_gpu.py tests/test_detector_chain_gpu.py tests/test_mot_gpu.py tests/test_e2e_parity_gpu.py tests/test_two_process_gpu.py tests/test_app_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x 2>&1 | tail -5
for i in 1 2 3 4; do for v in 11 00 10; do
  FASTMOT_PRE_AHEAD=${v:0:1} FASTMOT_DECODE_OFF=${v:1:1} timeout 300 python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-variants > $O/bench_${v}_$i.json 2> /dev/null
  python -c "
import json; d=json.load(open('$O/bench_${v}_$i.json')); print('ahead,decode_off=$v', 'fps', d['value'], 'net_ms', d['roofline']['net_ms_per_frame'], d['config']['stage_ms'])"
done; done
for v in 11 00; do FASTMOT_PRE_AHEAD=${v:0:1} FASTMOT_DECODE_OFF=${v:1:1} timeout 300 python bench.py --config 4 --steps 60 --warmup 5 --no-cpu-baseline --no-variants > $O/bench4_$v.json 2> /dev/null
  python -c "
import json; d=json.load(open('$O/bench4_$v.json')); print('config4 $v', 'fps', d['value'], 'net_ms', d['roofline']['net_ms_per_frame'])"
done
timeout 300 python scripts/trace_pipeline.py --show 2 > $O/pipeline_trace_new.txt 2> /dev/null; grep -E "^det:|^post|^# config|period|^host: detections" $O/pipeline_trace_new.txt | head -16
