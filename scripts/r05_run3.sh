#!/bin/bash
export TMPDIR=/tmp FASTMOT_RANDOM_WEIGHTS=1
R=$GRAFT_REPO_ROOT; [ -n "$R" ] || R=$(pwd)
O=$R/gpurun_out/r05c; mkdir -p $O
cd $R
for v in base role ilv; do
  L=$R/fastmot_amd/libfastmot_hip_$v.so; [ $v = base ] && L=$R/fastmot_amd/libfastmot_hip.so
  FASTMOT_LIB_PATH=$L timeout 300 python -m pytest tests/test_conv_gpu.py -k "convd" -q --maxfail=20 2>&1 | tail -25 > $O/pytest_$v.txt; echo "$v: $(tail -1 $O/pytest_$v.txt)"
  FASTMOT_LIB_PATH=$L timeout 400 python scripts/convd_sweep.py all > $O/sweep_$v.txt 2> $O/sweep_$v.err; tail -2 $O/sweep_$v.txt
done
FASTMOT_LIB_PATH=$R/fastmot_amd/libfastmot_hip_timing.so timeout 200 python scripts/convd_timing.py > $O/timing_base.txt 2> $O/timing_base.err
FASTMOT_LIB_PATH=$R/fastmot_amd/libfastmot_hip_roletiming.so timeout 200 python scripts/convd_timing.py > $O/timing_role.txt 2> $O/timing_role.err
grep -A2 "  full" $O/timing_role.txt | grep -v periods | head -40
