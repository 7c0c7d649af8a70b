#!/bin/bash
# full GPU suite + smoke + bench (config 1 and 4) after the host-LAP default / Python glue changes
mkdir -p gpurun_out/c20
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > gpurun_out/c20/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/c20/smoke.txt 2>&1
timeout 600 python bench.py > gpurun_out/c20/bench1.json 2> gpurun_out/c20/bench1.err
timeout 400 python bench.py --config 4 --steps 40 --warmup 5 --no-variants > gpurun_out/c20/bench4.json 2> gpurun_out/c20/bench4.err
FASTMOT_FLOW_TIMING=1 timeout 300 python bench.py --steps 200 --warmup 20 --no-variants > gpurun_out/c20/bench1_t.json 2> gpurun_out/c20/bench1_t.err
tail -5 gpurun_out/c20/pytest.txt; cat gpurun_out/c20/smoke.txt | tail -2; cat gpurun_out/c20/bench1.json | cut -c1-600; cat gpurun_out/c20/bench4.json | cut -c1-400
