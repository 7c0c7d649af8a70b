#!/bin/bash
export TMPDIR=/tmp FASTMOT_RANDOM_WEIGHTS=1
R=$GRAFT_REPO_ROOT; [ -n "$R" ] || R=$(pwd)
O=$R/gpurun_out/r05j; mkdir -p $O
cd $R
timeout 200 python -m pytest tests/test_conv_gpu.py -k "streamed" -m gpu -q 2>&1 | tail -1
for i in 1 2; do for ord in 1 0; do
  cd /tmp && rm -rf /tmp/tr_$ord$i && FASTMOT_CONVS_ORDER=$ord timeout 150 rocprofv3 --kernel-trace -d /tmp/tr_$ord$i -o t -- python $R/scripts/trace_net.py 0 YOLOv4_608 > /dev/null 2>&1
  cd $R && python scripts/layer_roofline.py /tmp/tr_$ord$i YOLOv4_608 > $O/layers_608_order${ord}_$i.txt 2>&1; echo "order=$ord run $i: $(tail -2 $O/layers_608_order${ord}_$i.txt | head -1)"
done; done
for i in 1 2; do for ord in 1 0; do
  FASTMOT_CONVS_ORDER=$ord timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-variants > $O/bench_order${ord}_$i.json 2> /dev/null
  python -c "
import json; d=json.load(open('$O/bench_order${ord}_$i.json')); print('order=$ord', 'fps', d['value'], 'net_ms', d['roofline']['net_ms_per_frame'])"
done; done
