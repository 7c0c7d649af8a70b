#!/bin/bash
# round 5, first GPU call: parity of convd.hip, single-layer sweep, per-layer tables with and without it
export TMPDIR=/tmp FASTMOT_RANDOM_WEIGHTS=1
R=$GRAFT_REPO_ROOT; [ -n "$R" ] || R=$(pwd)
O=$R/gpurun_out/r05a; mkdir -p $O
cd $R
timeout 400 python -m pytest tests/test_conv_gpu.py -k "convd" -q --maxfail=40 2>&1 | tail -60 > $O/pytest_convd.txt; tail -3 $O/pytest_convd.txt
timeout 600 python scripts/convd_sweep.py all > $O/sweep.txt 2> $O/sweep.err; tail -5 $O/sweep.txt
for lvl in 2 0; do
  cd /tmp && rm -rf /tmp/tr_$lvl && FASTMOT_CONVD=$lvl timeout 150 rocprofv3 --kernel-trace -d /tmp/tr_$lvl -o t -- python $R/scripts/trace_net.py 0 > /dev/null 2>&1
  cd $R && FASTMOT_CONVD=$lvl python scripts/layer_roofline.py /tmp/tr_$lvl > $O/yolo_layer_roofline_convd$lvl.txt 2>&1; tail -2 $O/yolo_layer_roofline_convd$lvl.txt
done
for lvl in 2 0; do
  cd /tmp && rm -rf /tmp/tp_$lvl && FASTMOT_CONVD=$lvl timeout 150 rocprofv3 --kernel-trace -d /tmp/tp_$lvl -o t -- python $R/scripts/trace_net.py 0 YOLOv4P6_1280 > /dev/null 2>&1
  cd $R && FASTMOT_CONVD=$lvl python scripts/layer_roofline.py /tmp/tp_$lvl YOLOv4P6_1280 > $O/p6_layer_roofline_convd$lvl.txt 2>&1; tail -2 $O/p6_layer_roofline_convd$lvl.txt
done
