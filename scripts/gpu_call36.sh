#!/bin/bash
export TMPDIR=/tmp FASTMOT_RANDOM_WEIGHTS=1
O=gpurun_out/c36; mkdir -p $O
PROFILE_H2D=1 PROFILE_PREFETCH=1 FASTMOT_FLOW_TIMING_VERBOSE=1 timeout 300 python scripts/profile_step.py > $O/tl.txt 2>&1
grep -E "ms/step|flow_predict stages" $O/tl.txt | tail -2; grep -E "sub-stages" $O/tl.txt | tail -1; grep -E "^(mot|trk|ctx|ext|det)\." $O/tl.txt | head -24
