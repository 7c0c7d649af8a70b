#!/bin/bash
export TMPDIR=/tmp FASTMOT_RANDOM_WEIGHTS=1
R=$GRAFT_REPO_ROOT; [ -n "$R" ] || R=$(pwd)
O=$R/gpurun_out/r05g; mkdir -p $O
cd $R
timeout 1300 python -m pytest tests -m gpu -q --durations=70 2>&1 | tail -110 > $O/pytest_gpu_durations.txt; tail -2 $O/pytest_gpu_durations.txt
bash scripts/collect_pmc.sh > $O/pmc.log 2>&1; cp gpurun_out/pmc_conv.json gpurun_out/pmc_FETCH_SIZE.txt gpurun_out/pmc_WRITE_SIZE.txt gpurun_out/pmc_sq_yolo.txt gpurun_out/pmc_layers.txt $O/ 2>/dev/null; tail -3 $O/pmc.log | cut -c1-300
