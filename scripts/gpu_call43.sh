#!/bin/bash
export TMPDIR=/tmp FASTMOT_RANDOM_WEIGHTS=1
O=gpurun_out/c43; mkdir -p $O
python -c "import os; print('cpus', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))"; nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-variants > $O/$tag.json 2> $O/$tag.err; python - <<P
import json
d=json.loads(open('$O/$tag.json').read().strip().splitlines()[-1]); print('$tag', d['value'], 'det ms', d['roofline']['net_ms_per_frame'])
P
grep "stage ms" $O/$tag.err | tail -1; }
for i in 1 2 3; do
run default_$i A=1
run threads1_$i FASTMOT_FLOW_THREADS=1
run threads3_$i FASTMOT_FLOW_THREADS=3
done
