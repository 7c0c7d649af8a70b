#!/bin/bash
export TMPDIR=/tmp FASTMOT_RANDOM_WEIGHTS=1
O=$GRAFT_REPO_ROOT/gpurun_out/c33; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_fullsize_gpu.py tests/test_detect_gpu.py tests/test_e2e_parity_gpu.py -q -m gpu -x 2>&1 | tail -4 > $O/pytest.txt; tail -2 $O/pytest.txt
cd /tmp && rm -rf /tmp/tr33 && rocprofv3 --kernel-trace --stats -d /tmp/tr33 -o t -- python $GRAFT_REPO_ROOT/scripts/trace_net.py 0 > /tmp/tr33.log 2>&1
cd $GRAFT_REPO_ROOT && python scripts/layer_roofline.py /tmp/tr33 > $O/yolo_layer_roofline.txt 2>&1; tail -3 $O/yolo_layer_roofline.txt
cd /tmp && rm -rf /tmp/tr33b && rocprofv3 --kernel-trace --stats -d /tmp/tr33b -o t -- python $GRAFT_REPO_ROOT/scripts/trace_net.py 1 25 > /tmp/tr33.log 2>&1
python $GRAFT_REPO_ROOT/scripts/rocpd_dispatches.py $(find /tmp/tr33b -name '*.db' | head -1) 32 > $O/osnet_b25.txt 2>&1; tail -1 $O/osnet_b25.txt
cd $GRAFT_REPO_ROOT && timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; cut -c1-200 $O/bench.json; grep -o '"variants.*' $O/bench.json | cut -c1-200; grep -o '"roofline.*' $O/bench.json | cut -c1-400
