#!/bin/bash
# round 3, call 2: which instruction goes wrong in lanes 48-63?  extended capture + library without packed-fp32 (no SLP)
export TMPDIR=/tmp FASTMOT_RANDOM_WEIGHTS=1
O=$GRAFT_REPO_ROOT/gpurun_out/r3c2; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 400 python scripts/lk_bisect.py 300 litechain production,diag-dpp+capture,diag-dpp+checks,diag-lds+checks,diag-both+checks > $O/bisect_litechain.txt 2>&1; grep -A12 "^    point" $O/bisect_litechain.txt | head -60 | cut -c1-330; grep "^hammer" $O/bisect_litechain.txt | cut -c1-400
FASTMOT_LIB_PATH=$GRAFT_REPO_ROOT/fastmot_amd/build/libfastmot_hip_noslp.so timeout 400 python scripts/lk_bisect.py 300 litechain production,diag-dpp,diag-dpp+checks,diag-lds+checks,production > $O/bisect_noslp.txt 2>&1; echo NOSLP; grep "^hammer" $O/bisect_noslp.txt | cut -c1-400
