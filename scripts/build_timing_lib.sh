#!/bin/bash
# libfastmot_hip_timing.so = the in-tree library with ONE source rebuilt with profiling flags (cycle stamps / ablations):
#   bash scripts/build_timing_lib.sh convd.hip -DFM_CONVD_TIMING      then run with FASTMOT_LIB_PATH=fastmot_amd/libfastmot_hip_timing.so
set -e
cd "$(dirname "$0")/.."
python -m fastmot_amd.build > /dev/null
SRC=$1; shift
B=fastmot_amd/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result "$@" -c fastmot_amd/csrc/$SRC -o $B/timing_${SRC%.hip}.o
OBJS=$(ls $B/*.o | grep -v "/timing_" | grep -v "/${SRC%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o fastmot_amd/libfastmot_hip_timing.so $OBJS $B/timing_${SRC%.hip}.o
echo built fastmot_amd/libfastmot_hip_timing.so
