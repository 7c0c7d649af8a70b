#!/bin/bash
export TMPDIR=/tmp FASTMOT_RANDOM_WEIGHTS=1
mkdir -p gpurun_out
O=gpurun_out
python -m pytest tests -m gpu -q -x --timeout=900 > $O/c5_pytest.log 2>&1; tail -3 $O/c5_pytest.log
python bench.py > $O/c5_bench.json 2> $O/c5_bench.err; tail -c 2200 $O/c5_bench.json; tail -2 $O/c5_bench.err
FASTMOT_FLOW_TIMING_VERBOSE=1 python scripts/profile_step.py > $O/c5_profile_step.txt 2>&1; grep -E "ms/step|flow_predict|sub-stages|compute_flow|apply_kalman|trk.update|detect_sync|extract_sync" $O/c5_profile_step.txt
cd /tmp && rm -rf /tmp/prof5 && rocprofv3 --kernel-trace --stats -d /tmp/prof5 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-variants > $GRAFT_REPO_ROOT/$O/c5_bench_prof.json 2>/dev/null
cd $GRAFT_REPO_ROOT && python scripts/rocpd_summary.py "$(find /tmp/prof5 -name '*.db' | head -1)" > $O/c5_kernel_stats.txt 2>&1
head -12 $O/c5_kernel_stats.txt | cut -c1-140
