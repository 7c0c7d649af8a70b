#!/bin/bash
# libfastmot_hip_<NAME>.so = the in-tree library with ONE source rebuilt with extra flags (cycle stamps, ablations, A/B variants):
#   bash scripts/build_timing_lib.sh convd.hip -DFM_CONVD_TIMING      then run with FASTMOT_LIB_PATH=fastmot_amd/libfastmot_hip_timing.so
#   NAME=role bash scripts/build_timing_lib.sh convd.hip -DFM_CONVD_ROLE=1   -> fastmot_amd/libfastmot_hip_role.so
set -e
cd "$(dirname "$0")/.."
python -m fastmot_amd.build > /dev/null
SRC=$1; shift
NAME=${NAME:-timing}
B=fastmot_amd/build
mkdir -p $B/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result "$@" -c fastmot_amd/csrc/$SRC -o $B/variants/${NAME}_${SRC%.hip}.o
OBJS=$(ls $B/*.o | grep -v "/${SRC%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o fastmot_amd/libfastmot_hip_${NAME}.so $OBJS $B/variants/${NAME}_${SRC%.hip}.o
echo built fastmot_amd/libfastmot_hip_${NAME}.so
