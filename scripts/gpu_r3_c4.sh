#!/bin/bash
# round 3, call 4: the fixed build -- LK tests un-isolated, e2e parity (all configs) repeated, bench
export TMPDIR=/tmp FASTMOT_RANDOM_WEIGHTS=1
O=$GRAFT_REPO_ROOT/gpurun_out/r3c4; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 300 python scripts/lk_bisect.py 300 litechain production,diag-dpp+checks,diag-lds+checks > $O/bisect_fixed.txt 2>&1; grep "^hammer" $O/bisect_fixed.txt | cut -c1-330
timeout 900 python -m pytest tests/test_flow_gpu.py tests/test_mot_gpu.py -m gpu -q -x 2>&1 | tail -4 > $O/pytest_flow.txt; tail -2 $O/pytest_flow.txt
timeout 1500 python -m pytest tests/test_e2e_parity_gpu.py -m gpu -q 2>&1 | tail -15 > $O/pytest_e2e.txt; tail -6 $O/pytest_e2e.txt
for i in 1 2 3 4 5 6 7 8 9 10; do timeout 300 python -m pytest "tests/test_e2e_parity_gpu.py::test_mot_step_equals_oracle_1080p_50" -m gpu -q 2>&1 | tail -1; done > $O/e2e_x10.txt; cat $O/e2e_x10.txt
timeout 600 python bench.py --no-cpu-baseline > $O/bench_n1.json 2> $O/bench_n1.err; cut -c1-200 $O/bench_n1.json
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver.json 2> $O/bench_driver.err; cut -c1-200 $O/bench_driver.json
