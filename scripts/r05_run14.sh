#!/bin/bash
# pipeline A/B of the OSNet launch shapes, one knob reverted at a time
export TMPDIR=/tmp FASTMOT_RANDOM_WEIGHTS=1
R=$GRAFT_REPO_ROOT; [ -n "$R" ] || R=$(pwd)
O=$R/gpurun_out/r05n; mkdir -p $O
cd $R
for i in 1 2 3 4; do for v in new old order0 pad1 gs256 wide800; do
  case $v in
    new) E="A=1";; old) E="FASTMOT_LCH_ORDER=0 FASTMOT_LCH_PAD=1 FASTMOT_GS2_PIX=256";;
    order0) E="FASTMOT_LCH_ORDER=0";; pad1) E="FASTMOT_LCH_PAD=1";; gs256) E="FASTMOT_GS2_PIX=256";; wide800) E="FASTMOT_LCH_WIDE_MAX=800";;
  esac
  env $E timeout 300 python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-variants > $O/bench_${v}_$i.json 2> /dev/null
  python -c "
import json; d=json.load(open('$O/bench_${v}_$i.json')); print('$v', 'fps', d['value'], 'net_ms', d['roofline']['net_ms_per_frame'], d['config']['stage_ms'])"
done; done
for i in 1 2; do for v in new old; do
  case $v in new) E="A=1";; old) E="FASTMOT_LCH_ORDER=0 FASTMOT_LCH_PAD=1 FASTMOT_GS2_PIX=256";; esac
  env $E timeout 300 python bench.py --config 4 --steps 60 --warmup 5 --no-cpu-baseline --no-variants > $O/bench4_${v}_$i.json 2> /dev/null
  python -c "
import json; d=json.load(open('$O/bench4_${v}_$i.json')); print('config4 $v', 'fps', d['value'], 'net_ms', d['roofline']['net_ms_per_frame'], d['config']['stage_ms'])"
done; done
