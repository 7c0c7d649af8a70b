#!/bin/bash
export TMPDIR=/tmp FASTMOT_RANDOM_WEIGHTS=1
mkdir -p gpurun_out; O=gpurun_out
python -m pytest tests/test_mot_gpu.py tests/test_detect_gpu.py tests/test_flow_gpu.py tests/test_e2e_parity_gpu.py tests/test_app_gpu.py tests/test_mot_multiclass_gpu.py -m gpu -q -x --timeout=900 > $O/c7_pytest.log 2>&1; tail -3 $O/c7_pytest.log
python bench.py --no-cpu-baseline > $O/c7_bench.json 2> $O/c7_bench.err; tail -c 1500 $O/c7_bench.json; tail -1 $O/c7_bench.err
python bench.py --no-cpu-baseline --no-variants --steps 600 --warmup 50 > $O/c7_bench600.json 2>/dev/null; python -c "import json;d=json.loads(open('$O/c7_bench600.json').read().strip().splitlines()[-1]);print('600 steps:',d['value'],d['config']['stage_ms'],d['roofline']['net_ms_per_frame'])"
