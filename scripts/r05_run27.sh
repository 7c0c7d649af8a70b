#!/bin/bash
export TMPDIR=/tmp FASTMOT_RANDOM_WEIGHTS=1
R=$GRAFT_REPO_ROOT; [ -n "$R" ] || R=$(pwd)
O=$R/gpurun_out/r05aa; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_flow_gpu.py tests/test_fullsize_gpu.py tests/test_e2e_parity_gpu.py -m gpu -q -x 2>&1 | tail -4
OLD=$R/fastmot_amd/libfastmot_hip_oldflow.so
for v in new old new old; do
  if [ $v = old ]; then E="FASTMOT_LIB_PATH=$OLD"; else E="A=1"; fi
  cd /tmp && rm -rf /tmp/kt4_$v && env $E FASTMOT_GRAPHS=0 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt4_$v -o b -- python $R/bench.py --config 4 --steps 20 --warmup 5 --no-cpu-baseline --no-variants > /dev/null 2>&1
  cd $R && python scripts/rocpd_summary.py "$(find /tmp/kt4_$v -name '*.db' | head -1)" > $O/config4_kernel_stats_$v.txt 2>&1; echo "== $v"; grep -E "eig_cand|gftt_select" $O/config4_kernel_stats_$v.txt | cut -c1-150
  cd /tmp && rm -rf /tmp/kt1_$v && env $E FASTMOT_GRAPHS=0 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt1_$v -o b -- python $R/bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-variants > /dev/null 2>&1
  cd $R && python scripts/rocpd_summary.py "$(find /tmp/kt1_$v -name '*.db' | head -1)" > $O/config1_kernel_stats_$v.txt 2>&1; grep -E "eig_cand" $O/config1_kernel_stats_$v.txt | cut -c1-150
done
