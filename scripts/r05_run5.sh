#!/bin/bash
export TMPDIR=/tmp FASTMOT_RANDOM_WEIGHTS=1
R=$GRAFT_REPO_ROOT; [ -n "$R" ] || R=$(pwd)
O=$R/gpurun_out/r05e; mkdir -p $O
cd $R
timeout 500 python -m pytest tests/test_conv_gpu.py tests/test_fullsize_gpu.py tests/test_darknet.py -m gpu -q --maxfail=20 2>&1 | tail -25 > $O/pytest_conv.txt; tail -2 $O/pytest_conv.txt
timeout 400 python scripts/convd_sweep.py all > $O/sweep.txt 2> $O/sweep.err; tail -1 $O/sweep.txt
for m in YOLOv4_608 YOLOv4P6_1280 YOLOv4CSP_640; do for lvl in 1 0; do
  cd /tmp && rm -rf /tmp/tr_$m$lvl && FASTMOT_CONVD=$lvl timeout 150 rocprofv3 --kernel-trace -d /tmp/tr_$m$lvl -o t -- python $R/scripts/trace_net.py 0 $m > /dev/null 2>&1
  cd $R && FASTMOT_CONVD=$lvl python scripts/layer_roofline.py /tmp/tr_$m$lvl $m > $O/layers_${m}_convd$lvl.txt 2>&1; echo "$m convd=$lvl: $(tail -2 $O/layers_${m}_convd$lvl.txt | head -1)"
done; done
for lvl in 1 0; do
  FASTMOT_CONVD=$lvl timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-variants > $O/bench_convd$lvl.json 2> $O/bench_convd$lvl.err; cut -c1-160 $O/bench_convd$lvl.json
  FASTMOT_CONVD=$lvl timeout 300 python bench.py --config 4 --steps 40 --warmup 10 --no-cpu-baseline --no-variants > $O/bench4_convd$lvl.json 2> $O/bench4_convd$lvl.err; cut -c1-160 $O/bench4_convd$lvl.json
done
