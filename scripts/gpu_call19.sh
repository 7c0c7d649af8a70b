#!/bin/bash
export TMPDIR=/tmp FASTMOT_RANDOM_WEIGHTS=1
mkdir -p gpurun_out; O=gpurun_out
cd /tmp && rm -rf /tmp/tr19 && rocprofv3 --kernel-trace --stats -d /tmp/tr19 -o t -- python $GRAFT_REPO_ROOT/scripts/trace_net.py 0 > /tmp/tr19.log 2>&1; find /tmp/tr19 -name '*.db' | head; tail -3 /tmp/tr19.log
cd $GRAFT_REPO_ROOT && python scripts/layer_roofline.py /tmp/tr19 > $O/c19_yolo_layer_roofline.txt 2>&1; tail -5 $O/c19_yolo_layer_roofline.txt
