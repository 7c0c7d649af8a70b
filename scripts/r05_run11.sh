#!/bin/bash
export TMPDIR=/tmp FASTMOT_RANDOM_WEIGHTS=1
R=$GRAFT_REPO_ROOT; [ -n "$R" ] || R=$(pwd)
O=$R/gpurun_out/r05k; mkdir -p $O
cd $R
timeout 400 python -m pytest tests/test_conv_gpu.py tests/test_fullsize_gpu.py tests/test_torchreid_loader.py -m gpu -q 2>&1 | tail -4
for i in 1 2; do for c in 16 64; do
  cd /tmp && rm -rf /tmp/tro_$c$i && FASTMOT_CONVD_MIN_CIN1=$c timeout 150 rocprofv3 --kernel-trace -d /tmp/tro_$c$i -o t -- python $R/scripts/trace_net.py 1 50 > /dev/null 2>&1
  cd $R && python scripts/rocpd_dispatches.py "$(find /tmp/tro_$c$i -name '*.db' | head -1)" 40 > $O/osnet_b50_mincin${c}_$i.txt 2>&1; echo "min cin $c run $i: $(tail -1 $O/osnet_b50_mincin${c}_$i.txt)"
done; done
for i in 1 2; do for c in 16 64; do
  FASTMOT_CONVD_MIN_CIN1=$c timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-variants > $O/bench_mincin${c}_$i.json 2> /dev/null
  python -c "
import json; d=json.load(open('$O/bench_mincin${c}_$i.json')); print('min cin $c', 'fps', d['value'], 'net_ms', d['roofline']['net_ms_per_frame'], d['config']['stage_ms'])"
done; done
