#!/bin/bash
# round 6, call z8: fused OSNet tail against the eleven launches: config[1] pipeline / sequential, driver command line, config[4]
. scripts/ab_lib.sh r06z8
ab_tests tests/test_conv_gpu.py tests/test_fullsize_gpu.py tests/test_detect_gpu.py tests/test_e2e_parity_gpu.py
ab_bench 4 --steps 300 --warmup 10 -- "tail:" "layers:FASTMOT_OSTAIL=0"
for i in 1 2; do for v in "tail:" "layers:FASTMOT_OSTAIL=0"; do n=$(ab_name "$v")
  env $(ab_env "$v") timeout 600 python bench.py --steps 300 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$n full', d['value'], 'seq', d.get('sequential_fps'), d.get('variants'))"
  env $(ab_env "$v") timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-variants 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$n driver-cmdline', d['value'])"
  env $(ab_env "$v") timeout 600 python bench.py --config 4 --steps 60 --warmup 10 --no-cpu-baseline --no-variants 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$n config4', d['value'], d['config'].get('stage_ms'))"
done; done
