#!/bin/bash
export TMPDIR=/tmp FASTMOT_RANDOM_WEIGHTS=1
R=$GRAFT_REPO_ROOT; [ -n "$R" ] || R=$(pwd)
O=$R/gpurun_out/r05z; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_conv_gpu.py tests/test_fullsize_gpu.py tests/test_torchreid_loader.py -m gpu -q -x 2>&1 | tail -4
OLD=$R/fastmot_amd/libfastmot_hip_oldstem.so
tr() {  # name which model/batch env...
  local name=$1 which=$2 arg=$3; shift 3
  cd /tmp && rm -rf /tmp/trs_$name && env "$@" timeout 150 rocprofv3 --kernel-trace -d /tmp/trs_$name -o t -- python $R/scripts/trace_net.py $which $arg > /dev/null 2>&1
  cd $R && python scripts/rocpd_dispatches.py "$(find /tmp/trs_$name -name '*.db' | head -1)" 400 > $O/disp_$name.txt 2>&1
  echo "$name: $(grep -m1 stem_conv $O/disp_$name.txt | cut -c1-60)  | $(tail -1 $O/disp_$name.txt)"
}
for i in 1 2; do
tr y608_new$i 0 YOLOv4_608 A=1
tr y608_old$i 0 YOLOv4_608 FASTMOT_LIB_PATH=$OLD
done
tr p6_new 0 YOLOv4P6_1280 A=1
tr p6_old 0 YOLOv4P6_1280 FASTMOT_LIB_PATH=$OLD
tr osnet_new 1 50 A=1
tr osnet_old 1 50 FASTMOT_LIB_PATH=$OLD
for i in 1 2 3; do for v in new old; do
  if [ $v = old ]; then E="FASTMOT_LIB_PATH=$OLD"; else E="A=1"; fi
  env $E timeout 300 python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-variants > $O/bench_${v}_$i.json 2> /dev/null
  python -c "
import json; d=json.load(open('$O/bench_${v}_$i.json')); print('$v', 'fps', d['value'], 'net_ms', d['roofline']['net_ms_per_frame'], d['config']['stage_ms'])"
done; done
