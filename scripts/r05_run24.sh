#!/bin/bash
export TMPDIR=/tmp FASTMOT_RANDOM_WEIGHTS=1
R=$GRAFT_REPO_ROOT; [ -n "$R" ] || R=$(pwd)
O=$R/gpurun_out/r05x; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_conv_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x 2>&1 | tail -4
for m in YOLOv4_608 YOLOv4P6_1280; do for v in 1 0; do
  cd /tmp && rm -rf /tmp/tr_${m}_$v && FASTMOT_CONVD_CIN32=$v rocprofv3 --kernel-trace -d /tmp/tr_${m}_$v -o t -- python $R/scripts/trace_net.py 0 $m > /dev/null 2>&1
  cd $R && FASTMOT_CONVD_CIN32=$v python scripts/layer_roofline.py /tmp/tr_${m}_$v $m > $O/layers_${m}_cin32_$v.txt 2>&1; echo "$m cin32=$v: $(tail -2 $O/layers_${m}_cin32_$v.txt | head -1)"; grep -E "x32 ->|x32->" $O/layers_${m}_cin32_$v.txt | cut -c1-130
done; done
for i in 1 2 3; do for v in 1 0; do
  FASTMOT_CONVD_CIN32=$v timeout 300 python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-variants > $O/bench_${v}_$i.json 2> /dev/null
  python -c "
import json; d=json.load(open('$O/bench_${v}_$i.json')); print('cin32=$v', 'fps', d['value'], 'net_ms', d['roofline']['net_ms_per_frame'])"
done; done
for v in 1 0; do FASTMOT_CONVD_CIN32=$v timeout 300 python bench.py --config 4 --steps 60 --warmup 5 --no-cpu-baseline --no-variants > $O/bench4_$v.json 2> /dev/null
  python -c "
import json; d=json.load(open('$O/bench4_$v.json')); print('config4 cin32=$v', 'fps', d['value'], 'net_ms', d['roofline']['net_ms_per_frame'])"
done
