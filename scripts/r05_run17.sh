#!/bin/bash
export TMPDIR=/tmp FASTMOT_RANDOM_WEIGHTS=1
R=$GRAFT_REPO_ROOT; [ -n "$R" ] || R=$(pwd)
O=$R/gpurun_out/r05q; mkdir -p $O
cd $R
timeout 500 python -m pytest tests/test_conv_gpu.py tests/test_fullsize_gpu.py tests/test_torchreid_loader.py -m gpu -q -x 2>&1 | tail -6
for b in 50; do for l in 0 2 4; do
  FASTMOT_LIB_PATH=$R/fastmot_amd/libfastmot_hip_lch.so FASTMOT_LCH_TIMING_LAUNCH=$l timeout 120 python scripts/lch_timing.py $b 2>&1 | tail -6
done; done | tee $O/lch_phase_cycles.txt
tr() {  # name batch env...
  local name=$1 b=$2; shift 2
  cd /tmp && rm -rf /tmp/tro_$name && env "$@" timeout 150 rocprofv3 --kernel-trace -d /tmp/tro_$name -o t -- python $R/scripts/trace_net.py 1 $b > /dev/null 2>&1
  cd $R && python scripts/rocpd_dispatches.py "$(find /tmp/tro_$name -name '*.db' | head -1)" 40 > $O/osnet_b${b}_$name.txt 2>&1; echo "b$b $name: $(tail -1 $O/osnet_b${b}_$name.txt)"
}
tr new1 50 A=1
tr new2 50 A=1
tr new300 300 A=1
for i in 1 2 3; do
  timeout 300 python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-variants > $O/bench_$i.json 2> /dev/null
  python -c "
import json; d=json.load(open('$O/bench_$i.json')); print('fps', d['value'], 'net_ms', d['roofline']['net_ms_per_frame'], d['config']['stage_ms'])"
done
