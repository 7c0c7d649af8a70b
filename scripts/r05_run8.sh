#!/bin/bash
export TMPDIR=/tmp FASTMOT_RANDOM_WEIGHTS=1
R=$GRAFT_REPO_ROOT; [ -n "$R" ] || R=$(pwd)
O=$R/gpurun_out/r05h; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_flow_gpu.py tests/test_fullsize_gpu.py tests/test_torchreid_loader.py tests/test_onnx_reader.py tests/test_e2e_parity_gpu.py -m gpu -q --durations=25 2>&1 | tail -45 > $O/pytest_part.txt; tail -1 $O/pytest_part.txt; grep -E "^[0-9.]+s call" $O/pytest_part.txt | head -12
for v in "FASTMOT_LK_PATCH=1" "FASTMOT_LK_PATCH=0" "FASTMOT_CONVD_NS_MAX=2" "FASTMOT_LK_PATCH=1"; do
  env $v timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_$v.json 2> $O/bench_$v.err
  python - "$O/bench_$v.json" "$v" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
st = {r['stage']: r for r in d.get('stage_roofline', [])}
lk = [r for r in d.get('stage_roofline', []) if 'LK' in r['stage']]
print(sys.argv[2], 'fps', d['value'], 'seq', d.get('sequential_fps'), 'net_ms', d['roofline']['net_ms_per_frame'], 'frac', d['roofline']['frac'],
      'LK', [(r['stage'], r.get('us') or r.get('median_us') or r) for r in lk][:2])
PY
done
