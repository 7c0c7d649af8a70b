#!/bin/bash
export TMPDIR=/tmp FASTMOT_RANDOM_WEIGHTS=1
O=gpurun_out/c35; mkdir -p $O
timeout 900 python -m pytest tests/test_e2e_parity_gpu.py tests/test_mot_gpu.py tests/test_flow_gpu.py tests/test_mot_multiclass_gpu.py -q -m gpu -x 2>&1 | tail -4 > $O/pytest.txt; tail -2 $O/pytest.txt
for i in 1 2; do for x in 1 0; do
FASTMOT_LK_EXCLUSION=$x timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-variants > $O/bench_x${x}_$i.json 2> $O/bench_x${x}_$i.err
python - <<P
import json
d=json.loads(open('$O/bench_x${x}_$i.json').read().strip().splitlines()[-1]); print('exclusion=$x run $i', d['value'], 'det ms', d['roofline']['net_ms_per_frame'])
P
grep "stage ms" $O/bench_x${x}_$i.err | tail -1
done; done
timeout 300 python bench.py > $O/bench_full.json 2> $O/bench_full.err; python - <<P
import json
d=json.loads(open('$O/bench_full.json').read().strip().splitlines()[-1]); print('full', d['value'], d['variants'], d['parity'].get('all_identical'), d['cpu_baseline']['value'])
P
timeout 300 python scripts/stress_determinism.py > $O/stress_det.txt 2>&1; tail -3 $O/stress_det.txt
