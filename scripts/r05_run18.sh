#!/bin/bash
export TMPDIR=/tmp FASTMOT_RANDOM_WEIGHTS=1
R=$GRAFT_REPO_ROOT; [ -n "$R" ] || R=$(pwd)
O=$R/gpurun_out/r05s; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_conv_gpu.py -m gpu -q -x -k "convd" 2>&1 | tail -6
SWEEP_SPLIT=1 timeout 900 python scripts/convd_sweep.py small > $O/convd_sweep_split.txt 2>&1
grep -E "^##|tiled|streamed" $O/convd_sweep_split.txt | cut -c1-120
for s in $(grep "^## " $O/convd_sweep_split.txt | awk '{print $2}' | tr -d ':'); do echo "best $s: $(awk -v s="## $s:" '$0 ~ s {f=1; next} /^##/ {f=0} f && /convd/' $O/convd_sweep_split.txt | sort -t'v' -k2 -n | sort -k6 -n | head -3 | cut -c1-90 | tr '\n' '|')"; done
