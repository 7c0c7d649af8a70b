#!/bin/bash
export TMPDIR=/tmp FASTMOT_RANDOM_WEIGHTS=1
O=$GRAFT_REPO_ROOT/gpurun_out/c58; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_fullsize_gpu.py -q -m gpu -x 2>&1 | tail -12 > $O/pytest.txt; tail -4 $O/pytest.txt
for m in 1; do
cd /tmp && rm -rf /tmp/tr58_$m && FASTMOT_CONVS_HALO=$m rocprofv3 --kernel-trace --stats -d /tmp/tr58_$m -o t -- python $GRAFT_REPO_ROOT/scripts/trace_net.py 0 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python scripts/layer_roofline.py /tmp/tr58_$m > $O/yolo_layer_roofline_halo_$m.txt 2>&1; tail -2 $O/yolo_layer_roofline_halo_$m.txt | head -1
done
grep -E "^ *(39|41|49|53|58|60|73|81) " $O/yolo_layer_roofline_halo_1.txt | cut -c1-120
grep -E "^ *(39|41|49|53|58|60|73|81) " $O/yolo_layer_roofline_halo_0.txt | cut -c1-120
