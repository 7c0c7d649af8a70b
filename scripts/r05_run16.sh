#!/bin/bash
export TMPDIR=/tmp FASTMOT_RANDOM_WEIGHTS=1
R=$GRAFT_REPO_ROOT; [ -n "$R" ] || R=$(pwd)
O=$R/gpurun_out/r05p; mkdir -p $O
cd $R
for b in 50 300; do for l in 0 2 4; do
  FASTMOT_LIB_PATH=$R/fastmot_amd/libfastmot_hip_lch.so FASTMOT_LCH_TIMING_LAUNCH=$l timeout 120 python scripts/lch_timing.py $b 2>&1 | tail -6
done; done | tee $O/lch_phase_cycles.txt
