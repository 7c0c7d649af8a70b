#!/bin/bash
export TMPDIR=/tmp FASTMOT_RANDOM_WEIGHTS=1
R=$GRAFT_REPO_ROOT; [ -n "$R" ] || R=$(pwd)
O=$R/gpurun_out/r05i; mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_conv_gpu.py -k "streamed or yolov4 or convd" -m gpu -q 2>&1 | tail -3
for i in 1 2 3; do for v in "FASTMOT_CONVD_NS_MAX=0" "FASTMOT_CONVD_NS_MAX=2" "FASTMOT_CONVD_NS_MAX=3"; do
  env $v timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-variants > $O/bench_${v}_$i.json 2> /dev/null
  python - "$O/bench_${v}_$i.json" "$v" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[2], 'fps', d['value'], 'net_ms', d['roofline']['net_ms_per_frame'])
PY
done; done
cd /tmp && rm -rf /tmp/tr_x && timeout 150 rocprofv3 --kernel-trace -d /tmp/tr_x -o t -- python $R/scripts/trace_net.py 0 YOLOv4_608 > /dev/null 2>&1
cd $R && python scripts/layer_roofline.py /tmp/tr_x YOLOv4_608 > $O/layers_608.txt 2>&1; tail -2 $O/layers_608.txt | head -1
