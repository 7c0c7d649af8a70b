#!/bin/bash
export TMPDIR=/tmp FASTMOT_RANDOM_WEIGHTS=1
R=$GRAFT_REPO_ROOT; [ -n "$R" ] || R=$(pwd)
O=$R/gpurun_out/r05l; mkdir -p $O
cd $R
for i in 1 2 3; do for m in 0 1 2; do
  FASTMOT_STREAM_PRIO=$m timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-variants > $O/bench_prio${m}_$i.json 2> /dev/null
  python -c "
import json; d=json.load(open('$O/bench_prio${m}_$i.json')); print('prio mode $m', 'fps', d['value'], 'net_ms', d['roofline']['net_ms_per_frame'], d['config']['stage_ms'])"
done; done
