#!/bin/bash
export TMPDIR=/tmp FASTMOT_RANDOM_WEIGHTS=1
mkdir -p gpurun_out
O=gpurun_out
rm -f $O/c4_*.txt
run() { echo "== $*" >> $O/c4_io.txt; env "$@" FASTMOT_FLOW_TIMING_VERBOSE=1 python scripts/profile_step.py 2>&1 | grep -E "ms/step|flow_predict stages|flow sub-stages|flow_estimate:|trk.compute_flow" >> $O/c4_io.txt; }
run FASTMOT_LK_IO=1 FASTMOT_PREP_OUT=0
run FASTMOT_LK_IO=0 FASTMOT_PREP_OUT=0
run FASTMOT_LK_IO=2 FASTMOT_PREP_OUT=1
run FASTMOT_LK_IO=0 FASTMOT_PREP_OUT=1
cat $O/c4_io.txt
python -m pytest tests -m gpu -q -x --timeout=900 > $O/c4_pytest.log 2>&1; tail -4 $O/c4_pytest.log
timeout 600 python bench.py --config 2 --steps 100 --warmup 10 --no-variants > $O/c4_bench_cfg2.json 2> $O/c4_bench_cfg2.err; tail -c 2500 $O/c4_bench_cfg2.json; tail -3 $O/c4_bench_cfg2.err
timeout 900 python bench.py --config 4 --steps 40 --warmup 5 --no-variants > $O/c4_bench_cfg4.json 2> $O/c4_bench_cfg4.err; tail -c 2500 $O/c4_bench_cfg4.json; tail -3 $O/c4_bench_cfg4.err
