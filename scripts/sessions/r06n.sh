#!/bin/bash
# round 6, call n: timing events on every 4th pass only; slot wait elided when the host sees the event complete
. scripts/ab_lib.sh r06n
ab_tests tests/test_detect_gpu.py tests/test_mot_gpu.py tests/test_e2e_parity_gpu.py tests/test_two_process_gpu.py tests/test_app_gpu.py
ab_bench 4 --steps 300 --warmup 10 -- "new:" 
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/bench_driver.json')); print('driver cmdline: value', d['value'], 'seq', d.get('sequential_fps'), 'frac', d['roofline']['frac'], 'samples', d['roofline']['net_ms_samples'])"
