#!/bin/bash
export TMPDIR=/tmp FASTMOT_RANDOM_WEIGHTS=1
mkdir -p gpurun_out
O=gpurun_out
python -m pytest tests/test_flow_gpu.py tests/test_e2e_parity_gpu.py tests/test_fullsize_gpu.py tests/test_detect_gpu.py tests/test_mot_gpu.py -m gpu -q -s --timeout=900 > $O/c2_parity.log 2>&1; tail -3 $O/c2_parity.log
for pts in 8 16 32 64; do
  echo "== LK_PTS=$pts" >> $O/c2_lk.txt
  FASTMOT_LK_PTS=$pts FASTMOT_FLOW_TIMING_VERBOSE=1 python scripts/profile_step.py 2>&1 | grep -E "ms/step|flow_predict stages|flow_estimate|trk.compute_flow|ctx.flow_predict" >> $O/c2_lk.txt
done
for th in 1 4 8; do
  echo "== FLOW_THREADS=$th" >> $O/c2_lk.txt
  FASTMOT_FLOW_THREADS=$th FASTMOT_FLOW_TIMING_VERBOSE=1 python scripts/profile_step.py 2>&1 | grep -E "ms/step|flow_predict stages|flow_estimate" >> $O/c2_lk.txt
done
echo "== PYR_FUSED=0" >> $O/c2_lk.txt
FASTMOT_PYR_FUSED=0 python scripts/profile_step.py 2>&1 | grep -E "ms/step|flow_predict stages" >> $O/c2_lk.txt
cat $O/c2_lk.txt
python bench.py --no-cpu-baseline > $O/c2_bench.json 2> $O/c2_bench.err; tail -c 1500 $O/c2_bench.json
cd /tmp && rm -rf /tmp/prof2 && rocprofv3 --kernel-trace --stats -d /tmp/prof2 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-variants > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python scripts/rocpd_summary.py "$(find /tmp/prof2 -name '*.db' | head -1)" > $O/c2_kernel_stats.txt 2>&1
grep -E "lk_kernel|gftt|prepare_kernel|eig_kernel|pyr_|gray_half|fast_|resize_linear" $O/c2_kernel_stats.txt | cut -c1-150
