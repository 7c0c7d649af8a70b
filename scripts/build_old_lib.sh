#!/bin/bash
# libfastmot_hip_<NAME>.so = the in-tree library with ONE source taken from another commit (the "before" side of an A/B):
#   NAME=oldlch bash scripts/build_old_lib.sh 027c1d9^ litechain.hip
set -e
cd "$(dirname "$0")/.."
python -m fastmot_amd.build > /dev/null
REV=$1; SRC=$2; NAME=${NAME:-old}
B=fastmot_amd/build; mkdir -p $B/variants
git show $REV:fastmot_amd/csrc/$SRC > fastmot_amd/csrc/_old_$SRC
trap "rm -f fastmot_amd/csrc/_old_$SRC" EXIT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -c fastmot_amd/csrc/_old_$SRC -o $B/variants/${NAME}_${SRC%.hip}.o
OBJS=$(ls $B/*.o | grep -v "/${SRC%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o fastmot_amd/libfastmot_hip_${NAME}.so $OBJS $B/variants/${NAME}_${SRC%.hip}.o
echo built fastmot_amd/libfastmot_hip_${NAME}.so
