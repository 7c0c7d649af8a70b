#!/bin/bash
export TMPDIR=/tmp FASTMOT_RANDOM_WEIGHTS=1
R=$GRAFT_REPO_ROOT; [ -n "$R" ] || R=$(pwd)
O=$R/gpurun_out/r05ab; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_conv_gpu.py tests/test_fullsize_gpu.py tests/test_torchreid_loader.py -m gpu -q -x 2>&1 | tail -4
OLD=$R/fastmot_amd/libfastmot_hip_oldlch.so
trace_osnet() {  # name batch env... (NOT "tr": the function below pipes through tr(1) -- the first version of this script recursed into itself and burnt the round's last 20 GPU minutes)
  local name=$1 b=$2; shift 2
  cd /tmp && rm -rf /tmp/tro_$name && env "$@" timeout 150 rocprofv3 --kernel-trace -d /tmp/tro_$name -o t -- python $R/scripts/trace_net.py 1 $b > /dev/null 2>&1
  cd $R && python scripts/rocpd_dispatches.py "$(find /tmp/tro_$name -name '*.db' | head -1)" 40 > $O/osnet_b${b}_$name.txt 2>&1; echo "b$b $name: $(tail -1 $O/osnet_b${b}_$name.txt) | chains: $(grep litechain $O/osnet_b${b}_$name.txt | awk '{print $3}' | sed 's/dur=//' | tr '\n' ' ')"
}
for i in 1 2; do trace_osnet new$i 50 A=1; trace_osnet old$i 50 FASTMOT_LIB_PATH=$OLD; done
trace_osnet new 300 A=1; trace_osnet old 300 FASTMOT_LIB_PATH=$OLD
for i in 1 2 3 4; do for v in new old; do
  if [ $v = old ]; then E="FASTMOT_LIB_PATH=$OLD"; else E="A=1"; fi
  env $E timeout 300 python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-variants > $O/bench_${v}_$i.json 2> /dev/null
  python -c "
import json; d=json.load(open('$O/bench_${v}_$i.json')); print('$v', 'fps', d['value'], 'net_ms', d['roofline']['net_ms_per_frame'], d['config']['stage_ms'])"
done; done
for v in new old; do if [ $v = old ]; then E="FASTMOT_LIB_PATH=$OLD"; else E="A=1"; fi
  env $E timeout 300 python bench.py --config 4 --steps 60 --warmup 5 --no-cpu-baseline --no-variants > $O/bench4_$v.json 2> /dev/null
  python -c "
import json; d=json.load(open('$O/bench4_$v.json')); print('config4 $v', 'fps', d['value'], d['config']['stage_ms'])"
done
