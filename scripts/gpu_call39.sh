#!/bin/bash
# consolidated round-2 evidence run: full GPU suite, smoke, bench (configs 1/2/4), timelines, per-layer roofline, kernel stats, PMC
export TMPDIR=/tmp FASTMOT_RANDOM_WEIGHTS=1
O=$GRAFT_REPO_ROOT/gpurun_out/c59; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/pytest.txt; tail -2 $O/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; cut -c1-160 $O/bench_n1.json
timeout 400 python bench.py --config 2 --no-variants > $O/bench_config2.json 2> $O/bench_config2.err; cut -c1-160 $O/bench_config2.json
timeout 400 python bench.py --config 4 --steps 60 --warmup 6 --no-variants > $O/bench_config4.json 2> $O/bench_config4.err; cut -c1-160 $O/bench_config4.json
FASTMOT_FLOW_TIMING_VERBOSE=1 timeout 200 python scripts/profile_step.py > $O/profile_step_sequential.txt 2>&1
PROFILE_H2D=1 PROFILE_PREFETCH=1 FASTMOT_FLOW_TIMING_VERBOSE=1 timeout 200 python scripts/profile_step.py > $O/profile_step_pipelined.txt 2>&1
grep -E "ms/step|flow_predict stages" $O/profile_step_sequential.txt $O/profile_step_pipelined.txt
cd /tmp && rm -rf /tmp/tr39 && rocprofv3 --kernel-trace --stats -d /tmp/tr39 -o t -- python $GRAFT_REPO_ROOT/scripts/trace_net.py 0 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python scripts/layer_roofline.py /tmp/tr39 > $O/yolo_layer_roofline.txt 2>&1; tail -2 $O/yolo_layer_roofline.txt
cd /tmp && rm -rf /tmp/tr39b && rocprofv3 --kernel-trace --stats -d /tmp/tr39b -o t -- python $GRAFT_REPO_ROOT/scripts/trace_net.py 1 25 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/scripts/rocpd_dispatches.py $(find /tmp/tr39b -name '*.db' | head -1) 32 > $O/osnet_b25_dispatches.txt 2>&1; tail -1 $O/osnet_b25_dispatches.txt
cd /tmp && rm -rf /tmp/prof39 && rocprofv3 --kernel-trace --stats -d /tmp/prof39 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-variants > $O/bench_prof.json 2>/dev/null
cd $GRAFT_REPO_ROOT && python scripts/rocpd_summary.py "$(find /tmp/prof39 -name '*.db' | head -1)" > $O/bench_kernel_stats.txt 2>&1; head -12 $O/bench_kernel_stats.txt | cut -c1-150
bash scripts/collect_pmc.sh > $O/pmc.log 2>&1; cp gpurun_out/pmc_*.txt $O/ 2>/dev/null; cp gpurun_out/r02_pmc_conv.json $O/ 2>/dev/null; tail -3 $O/pmc.log
