#!/bin/bash
# OSNet launch-shape A/B (round 5): deepest chains first, unpadded C = 16 tiles (4 workgroups / CU), one item per thread in
# the gated sum, 512-thread chains up to 800 workgroups
export TMPDIR=/tmp FASTMOT_RANDOM_WEIGHTS=1
R=$GRAFT_REPO_ROOT; [ -n "$R" ] || R=$(pwd)
O=$R/gpurun_out/r05m; mkdir -p $O
cd $R
timeout 500 python -m pytest tests/test_conv_gpu.py tests/test_fullsize_gpu.py tests/test_torchreid_loader.py -m gpu -q 2>&1 | tail -4
tr() {  # name batch env...
  local name=$1 b=$2; shift 2
  cd /tmp && rm -rf /tmp/tro_$name && env "$@" timeout 150 rocprofv3 --kernel-trace -d /tmp/tro_$name -o t -- python $R/scripts/trace_net.py 1 $b > /dev/null 2>&1
  cd $R && python scripts/rocpd_dispatches.py "$(find /tmp/tro_$name -name '*.db' | head -1)" 40 > $O/osnet_b${b}_$name.txt 2>&1; echo "b$b $name: $(tail -1 $O/osnet_b${b}_$name.txt)"
}
OLD="FASTMOT_LCH_ORDER=0 FASTMOT_LCH_PAD=1 FASTMOT_GS2_PIX=256"
for i in 1 2; do
  tr new$i 50 A=1
  tr old$i 50 $OLD
done
tr order0 50 FASTMOT_LCH_ORDER=0
tr pad1 50 FASTMOT_LCH_PAD=1
tr gs256 50 FASTMOT_GS2_PIX=256
tr wide800 50 FASTMOT_LCH_WIDE_MAX=800
tr new 300 A=1
tr old 300 $OLD
tr wide4800 300 FASTMOT_LCH_WIDE_MAX=4800
for i in 1 2; do for v in new old; do
  if [ $v = old ]; then E="$OLD"; else E="A=1"; fi
  env $E timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-variants > $O/bench_${v}_$i.json 2> /dev/null
  python -c "
import json; d=json.load(open('$O/bench_${v}_$i.json')); print('$v', 'fps', d['value'], 'net_ms', d['roofline']['net_ms_per_frame'], d['config']['stage_ms'])"
done; done
