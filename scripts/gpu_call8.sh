#!/bin/bash
export TMPDIR=/tmp FASTMOT_RANDOM_WEIGHTS=1
mkdir -p gpurun_out; O=gpurun_out
python -m pytest tests -m gpu -q --timeout=900 > $O/c8_pytest.log 2>&1; tail -3 $O/c8_pytest.log
python bench.py > $O/c8_bench.json 2> $O/c8_bench.err; tail -c 2600 $O/c8_bench.json; tail -1 $O/c8_bench.err
python scripts/lap_crossover.py > $O/c8_lap_crossover.txt 2>&1; cat $O/c8_lap_crossover.txt
cd /tmp && rm -rf /tmp/tr8 && rocprofv3 --kernel-trace -d /tmp/tr8 -o t -- python $GRAFT_REPO_ROOT/scripts/trace_net.py 0 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python scripts/layer_roofline.py "$(find /tmp/tr8 -name '*.db' | head -1)" > $O/c8_yolo_layer_roofline.txt 2>&1; tail -4 $O/c8_yolo_layer_roofline.txt
bash scripts/collect_pmc.sh > $O/c8_pmc.log 2>&1; tail -2 $O/c8_pmc.log
