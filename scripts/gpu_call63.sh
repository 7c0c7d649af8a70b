#!/bin/bash
export TMPDIR=/tmp FASTMOT_RANDOM_WEIGHTS=1
O=$GRAFT_REPO_ROOT/gpurun_out/c63; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/pytest.txt; tail -3 $O/pytest.txt
timeout 300 python -m pytest tests/test_e2e_parity_gpu.py tests/test_flow_gpu.py -m gpu -q 2>&1 | tail -2 > $O/pytest2.txt; tail -1 $O/pytest2.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; cut -c1-160 $O/bench_n1.json
cd /tmp && rm -rf /tmp/tr63 && rocprofv3 --kernel-trace --stats -d /tmp/tr63 -o t -- python $GRAFT_REPO_ROOT/scripts/trace_net.py 0 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python scripts/layer_roofline.py /tmp/tr63 > $O/yolo_layer_roofline.txt 2>&1; tail -2 $O/yolo_layer_roofline.txt
