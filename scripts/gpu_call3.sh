#!/bin/bash
export TMPDIR=/tmp FASTMOT_RANDOM_WEIGHTS=1
mkdir -p gpurun_out
O=gpurun_out
rm -f $O/c3_*.txt
python -m pytest tests/test_flow_gpu.py tests/test_e2e_parity_gpu.py tests/test_fullsize_gpu.py::test_flow_predict_1080p_50_tracks tests/test_mot_gpu.py -m gpu -q --timeout=900 > $O/c3_parity.log 2>&1; tail -3 $O/c3_parity.log
FASTMOT_LK_PTS=64 python -m pytest tests/test_flow_gpu.py tests/test_fullsize_gpu.py::test_flow_predict_1080p_50_tracks -m gpu -q --timeout=900 > $O/c3_parity_lane.log 2>&1; tail -1 $O/c3_parity_lane.log
run() { echo "== $*" >> $O/c3_lk.txt; env "$@" FASTMOT_FLOW_TIMING_VERBOSE=1 python scripts/profile_step.py 2>&1 | grep -E "ms/step|flow_predict stages|flow_estimate:|trk.compute_flow|trk.apply_kalman|trk.update|ctx.detect_sync|ctx.extract_sync" >> $O/c3_lk.txt; }
run FASTMOT_LK_PTS=0
run FASTMOT_LK_PTS=16
run FASTMOT_LK_PTS=64
run FASTMOT_LK_PTS=0 FASTMOT_FLOW_THREADS=1
run FASTMOT_LK_PTS=0 FASTMOT_FLOW_THREADS=3
run FASTMOT_LK_PTS=0 FASTMOT_FLOW_THREADS=12
cat $O/c3_lk.txt
python bench.py --no-cpu-baseline > $O/c3_bench.json 2> $O/c3_bench.err; tail -c 1200 $O/c3_bench.json
cd /tmp && rm -rf /tmp/prof3 && rocprofv3 --kernel-trace --stats -d /tmp/prof3 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-variants > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python scripts/rocpd_summary.py "$(find /tmp/prof3 -name '*.db' | head -1)" > $O/c3_kernel_stats.txt 2>&1
grep -E "lk_|gftt|prepare_kernel|eig_kernel|pyr_|gray_half|fast_|resize_linear|copyBuffer" $O/c3_kernel_stats.txt | cut -c1-150
