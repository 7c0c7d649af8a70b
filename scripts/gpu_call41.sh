#!/bin/bash
export TMPDIR=/tmp FASTMOT_RANDOM_WEIGHTS=1
O=$GRAFT_REPO_ROOT/gpurun_out/c41; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_fullsize_gpu.py tests/test_darknet_gpu.py tests/test_flow_gpu.py -q -m gpu -x 2>&1 | tail -3 > $O/pytest.txt; tail -2 $O/pytest.txt
cd /tmp && rm -rf /tmp/tr41 && rocprofv3 --kernel-trace --stats -d /tmp/tr41 -o t -- python $GRAFT_REPO_ROOT/scripts/trace_net.py 0 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python scripts/layer_roofline.py /tmp/tr41 > $O/yolo_layer_roofline.txt 2>&1; tail -2 $O/yolo_layer_roofline.txt
cd /tmp && rm -rf /tmp/tr41b && rocprofv3 --kernel-trace --stats -d /tmp/tr41b -o t -- python $GRAFT_REPO_ROOT/scripts/trace_net.py 1 25 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/scripts/rocpd_dispatches.py $(find /tmp/tr41b -name '*.db' | head -1) 32 > $O/osnet_b25.txt 2>&1; grep -E "gated|span" $O/osnet_b25.txt | cut -c1-80
cd /tmp && rm -rf /tmp/prof41 && rocprofv3 --kernel-trace --stats -d /tmp/prof41 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-variants > $O/bench_prof.json 2>/dev/null
cd $GRAFT_REPO_ROOT && python scripts/rocpd_summary.py "$(find /tmp/prof41 -name '*.db' | head -1)" > $O/kernel_stats.txt 2>&1; grep -E "prepare_kernel|gftt|gated_sum" $O/kernel_stats.txt | cut -c1-150
