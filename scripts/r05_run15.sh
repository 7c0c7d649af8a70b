#!/bin/bash
export TMPDIR=/tmp FASTMOT_RANDOM_WEIGHTS=1
R=$GRAFT_REPO_ROOT; [ -n "$R" ] || R=$(pwd)
O=$R/gpurun_out/r05o; mkdir -p $O
cd $R
timeout 500 python -m pytest tests/test_conv_gpu.py tests/test_fullsize_gpu.py tests/test_torchreid_loader.py -m gpu -q 2>&1 | tail -4
tr() {  # name batch model env...
  local name=$1 b=$2 m=$3; shift 3
  cd /tmp && rm -rf /tmp/tro_$name && env "$@" timeout 150 rocprofv3 --kernel-trace -d /tmp/tro_$name -o t -- python $R/scripts/trace_net.py 1 $b > /dev/null 2>&1
  cd $R && python scripts/rocpd_dispatches.py "$(find /tmp/tro_$name -name '*.db' | head -1)" 40 > $O/osnet_b${b}_$name.txt 2>&1; echo "b$b $name: $(tail -1 $O/osnet_b${b}_$name.txt)"
}
tr new1 50 "" A=1
tr new2 50 "" A=1
tr new300 300 "" A=1
for i in 1 2 3; do
  timeout 300 python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-variants > $O/bench_$i.json 2> /dev/null
  python -c "
import json; d=json.load(open('$O/bench_$i.json')); print('fps', d['value'], 'net_ms', d['roofline']['net_ms_per_frame'], d['config']['stage_ms'])"
done
