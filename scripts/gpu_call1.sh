#!/bin/bash
# round-2 GPU call 1: full GPU suite, first bench line with H2D, profiles
export TMPDIR=/tmp FASTMOT_RANDOM_WEIGHTS=1
mkdir -p gpurun_out
O=gpurun_out
python -m pytest tests -m gpu -q -x --timeout=900 > $O/c1_pytest.log 2>&1; echo "pytest rc=$?" >> $O/c1_pytest.log
tail -5 $O/c1_pytest.log
python -m pytest tests/test_e2e_parity_gpu.py tests/test_fullsize_gpu.py -m gpu -q -s --timeout=900 > $O/c1_parity.log 2>&1
python bench.py > $O/c1_bench.json 2> $O/c1_bench.err; tail -c 3000 $O/c1_bench.json
FASTMOT_FLOW_TIMING_VERBOSE=1 python scripts/profile_step.py > $O/c1_profile_step.txt 2>&1
python scripts/profile_layers.py 0 > $O/c1_yolo_layers.txt 2>&1
python scripts/profile_layers.py 1 > $O/c1_osnet_layers.txt 2>&1
python scripts/boundary_cost.py > $O/c1_boundary.txt 2>&1
cd /tmp && rm -rf /tmp/prof1 && rocprofv3 --kernel-trace --stats -d /tmp/prof1 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-variants > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python scripts/rocpd_summary.py "$(find /tmp/prof1 -name '*.db' | head -1)" > $O/c1_kernel_stats.txt 2>&1
echo done
