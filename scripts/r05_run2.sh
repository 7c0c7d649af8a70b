#!/bin/bash
export TMPDIR=/tmp FASTMOT_RANDOM_WEIGHTS=1
R=$GRAFT_REPO_ROOT; [ -n "$R" ] || R=$(pwd)
O=$R/gpurun_out/r05b; mkdir -p $O
cd $R
timeout 300 scripts/bin/dma_bench > $O/dma_bench.txt 2>&1; tail -12 $O/dma_bench.txt
FASTMOT_LIB_PATH=$R/fastmot_amd/libfastmot_hip_timing.so timeout 300 python scripts/convd_timing.py > $O/convd_timing.txt 2> $O/convd_timing.err; head -40 $O/convd_timing.txt; tail -3 $O/convd_timing.err
