#!/bin/bash
# litechain: chain parameters staged once, fragments prefetched; LK scans.  Exactness + per-layer timing
O=gpurun_out/c25; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_fullsize_gpu.py tests/test_flow_gpu.py -q -m gpu -x 2>&1 | tail -5 > $O/pytest.txt
tail -3 $O/pytest.txt
timeout 200 python scripts/profile_layers.py 1 > $O/osnet_layers.txt 2>&1; grep -E "total|op16|op11" $O/osnet_layers.txt
FASTMOT_FLOW_TIMING_VERBOSE=1 timeout 200 python scripts/profile_step.py > $O/profile_step.txt 2>&1; grep -E "ms/step|flow_predict stages|sub-stages" $O/profile_step.txt | tail -3
