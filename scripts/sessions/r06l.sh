#!/bin/bash
# round 6, call l: stream priority levels per group (experiment)
. scripts/ab_lib.sh r06l
python -c "
import ctypes
" ; ab_bench 3 --steps 300 --warmup 10 -- "base:" "flow_mid:FASTMOT_PRIO_FLOW=1" "flow_lo:FASTMOT_PRIO_FLOW=2" "ext_mid:FASTMOT_PRIO_EXT=1" "ext_mid_flow_lo:FASTMOT_PRIO_EXT=1 FASTMOT_PRIO_FLOW=2" "det_mid:FASTMOT_PRIO_DET=1"
