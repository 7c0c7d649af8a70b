#!/bin/bash
export TMPDIR=/tmp FASTMOT_RANDOM_WEIGHTS=1
mkdir -p gpurun_out; O=gpurun_out
python -m pytest tests -m gpu -q --timeout=900 > $O/c18_pytest.log 2>&1; tail -3 $O/c18_pytest.log
python bench.py > $O/c18_bench.json 2> $O/c18_bench.err; tail -c 1800 $O/c18_bench.json | head -c 900; echo; tail -1 $O/c18_bench.err
FASTMOT_FLOW_TIMING_VERBOSE=1 python scripts/profile_step.py > $O/c18_profile_step.txt 2>&1; grep -E "ms/step|flow_predict stages|sub-stages|compute_flow" $O/c18_profile_step.txt
cd /tmp && rm -rf /tmp/tr18 && rocprofv3 --kernel-trace --stats -d /tmp/tr18 -o t -- python $GRAFT_REPO_ROOT/scripts/trace_net.py 0 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python scripts/layer_roofline.py "$(find /tmp/tr18 -name '*.db' | head -1)" > $O/c18_yolo_layer_roofline.txt 2>&1; tail -4 $O/c18_yolo_layer_roofline.txt
cd /tmp && rm -rf /tmp/prof18 && rocprofv3 --kernel-trace --stats -d /tmp/prof18 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-variants > $GRAFT_REPO_ROOT/$O/c18_bench_prof.json 2>/dev/null
cd $GRAFT_REPO_ROOT && python scripts/rocpd_summary.py "$(find /tmp/prof18 -name '*.db' | head -1)" > $O/c18_kernel_stats.txt 2>&1
grep -E "lk_wave|gftt|prepare_kernel" $O/c18_kernel_stats.txt | cut -c1-150
