#!/bin/bash
export TMPDIR=/tmp FASTMOT_RANDOM_WEIGHTS=1
R=$GRAFT_REPO_ROOT; [ -n "$R" ] || R=$(pwd)
O=$R/gpurun_out/r05d; mkdir -p $O
cd $R
lib() { [ $1 = base ] && echo $R/fastmot_amd/libfastmot_hip.so || echo $R/fastmot_amd/libfastmot_hip_$1.so; }
for v in base role; do
  FASTMOT_LIB_PATH=$(lib $v) timeout 300 python -m pytest tests/test_conv_gpu.py -k "convd" -q --maxfail=20 2>&1 | tail -25 > $O/pytest_$v.txt; echo "$v: $(tail -1 $O/pytest_$v.txt)"
done
for v in base epi0 role roleepi0; do
  FASTMOT_LIB_PATH=$(lib $v) timeout 300 python scripts/convd_sweep.py kscan > $O/kscan_$v.txt 2> $O/kscan_$v.err; echo "== $v"; cat $O/kscan_$v.txt | cut -c1-200
done
for v in base role; do
  FASTMOT_LIB_PATH=$(lib $v) timeout 400 python scripts/convd_sweep.py all > $O/sweep_$v.txt 2> $O/sweep_$v.err
done
FASTMOT_LIB_PATH=$(lib timing) timeout 200 python scripts/convd_timing.py > $O/timing_base.txt 2> $O/timing_base.err
FASTMOT_LIB_PATH=$(lib roletiming) timeout 200 python scripts/convd_timing.py > $O/timing_role.txt 2> $O/timing_role.err
grep -B1 -A3 "  full  " $O/timing_role.txt | grep -v periods | head -60
