#!/bin/bash
export TMPDIR=/tmp FASTMOT_RANDOM_WEIGHTS=1
R=$GRAFT_REPO_ROOT; [ -n "$R" ] || R=$(pwd)
O=$R/gpurun_out/r05f; mkdir -p $O
cd $R
timeout 400 python -m pytest tests/test_conv_gpu.py -k "convd" -q --maxfail=20 2>&1 | tail -25 > $O/pytest_convd.txt; tail -1 $O/pytest_convd.txt
timeout 500 python scripts/convd_sweep.py all > $O/sweep.txt 2> $O/sweep.err; tail -1 $O/sweep.txt
for m in YOLOv4_608 YOLOv4P6_1280 YOLOv4CSP_640; do for lvl in 1 0; do
  cd /tmp && rm -rf /tmp/tr_$m$lvl && FASTMOT_CONVD=$lvl timeout 150 rocprofv3 --kernel-trace -d /tmp/tr_$m$lvl -o t -- python $R/scripts/trace_net.py 0 $m > /dev/null 2>&1
  cd $R && FASTMOT_CONVD=$lvl python scripts/layer_roofline.py /tmp/tr_$m$lvl $m > $O/layers_${m}_convd$lvl.txt 2>&1; echo "$m convd=$lvl: $(tail -2 $O/layers_${m}_convd$lvl.txt | head -1)"
done; done
