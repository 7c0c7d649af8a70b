#!/bin/bash
export TMPDIR=/tmp FASTMOT_RANDOM_WEIGHTS=1
O=$GRAFT_REPO_ROOT/gpurun_out/r3c3; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python scripts/pkhaz2.py 24 > $O/pkhaz2.txt 2>&1; cat $O/pkhaz2.txt | cut -c1-200
