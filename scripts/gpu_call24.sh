#!/bin/bash
# LK window sums as DPP scans: exactness, disturbance (no CU isolation), timing
O=gpurun_out/c24; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_flow_gpu.py tests/test_e2e_parity_gpu.py -q -m gpu -x 2>&1 | tail -5 > $O/pytest.txt
tail -3 $O/pytest.txt
FASTMOT_LK_LDS=0 timeout 400 python scripts/stress_lk4.py 600 > $O/stress4_lds0.txt 2>&1; grep -i "hammer" $O/stress4_lds0.txt
FASTMOT_LK_LDS=0 timeout 400 python scripts/stress_lk6.py 400 > $O/stress6_lds0.txt 2>&1; grep -i "only\|hammer" $O/stress6_lds0.txt | head -12
for v in 150000 0; do echo "== LK_LDS=$v"; FASTMOT_LK_LDS=$v FASTMOT_FLOW_TIMING_VERBOSE=1 timeout 200 python scripts/profile_step.py 2>&1 | grep -E "ms/step|flow_predict stages|sub-stages" | tail -3; done
