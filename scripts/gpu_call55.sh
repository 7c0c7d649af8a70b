#!/bin/bash
export TMPDIR=/tmp FASTMOT_RANDOM_WEIGHTS=1
O=$GRAFT_REPO_ROOT/gpurun_out/c55; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_fullsize_gpu.py -q -m gpu -x 2>&1 | tail -3 > $O/pytest.txt; tail -2 $O/pytest.txt
for m in 1; do
cd /tmp && rm -rf /tmp/tr55_$m && FASTMOT_CONVS_PT2=$m rocprofv3 --kernel-trace --stats -d /tmp/tr55_$m -o t -- python $GRAFT_REPO_ROOT/scripts/trace_net.py 0 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python scripts/layer_roofline.py /tmp/tr55_$m > $O/yolo_layer_roofline_pt2_$m.txt 2>&1; tail -2 $O/yolo_layer_roofline_pt2_$m.txt | head -1
done
grep -E "^ *(24|58|60|73|75|77) " $O/yolo_layer_roofline_pt2_1.txt | cut -c1-120
grep -E "^ *(24|58|60|73|75|77) " $O/yolo_layer_roofline_pt2_0.txt | cut -c1-120
