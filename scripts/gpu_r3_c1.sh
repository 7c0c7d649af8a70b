#!/bin/bash
# round 3, call 1: bisect of the LK disturbance
export TMPDIR=/tmp FASTMOT_RANDOM_WEIGHTS=1
O=$GRAFT_REPO_ROOT/gpurun_out/r3c1; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 400 python scripts/lk_bisect.py 300 litechain > $O/bisect_litechain.txt 2>&1; tail -25 $O/bisect_litechain.txt | cut -c1-400
timeout 300 python scripts/lk_bisect.py 300 osnet production,diag-lds,diag-both+checks,diag-dpp+capture > $O/bisect_osnet.txt 2>&1; grep "^hammer" $O/bisect_osnet.txt | cut -c1-300
timeout 300 python scripts/lk_bisect.py 300 yolo production,diag-dpp+capture,diag-lds,diag-both+checks > $O/bisect_yolo.txt 2>&1; grep "^hammer" $O/bisect_yolo.txt | cut -c1-300
FASTMOT_FLOW_PRIO=0 timeout 300 python scripts/lk_bisect.py 300 litechain production,diag-lds > $O/bisect_prio0.txt 2>&1; grep "^hammer" $O/bisect_prio0.txt | cut -c1-300
