#!/bin/bash
# A/B of one environment knob on the end-to-end bench, alternating runs inside ONE process group on ONE box
# (box-to-box and run-to-run spread is +-10 %: never compare numbers from different gpurun calls).
#   bash scripts/ab.sh FASTMOT_LITECHAIN 0 1 [runs=4] [steps=600]
# prints one "value FPS ms" line per run and the mean per setting.
var=$1; a=$2; b=$3; runs=${4:-4}; steps=${5:-600}
for i in $(seq $runs); do
    for v in $a $b; do
        env $var=$v python bench.py --steps $steps --warmup 50 --no-cpu-baseline 2>/dev/null |
            python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'])"
    done
done | tee /tmp/ab_$$.txt
python - /tmp/ab_$$.txt <<'PY'
import sys, collections
acc = collections.defaultdict(list)
for line in open(sys.argv[1]):
    k, fps, ms = line.split()
    acc[k].append(float(fps))
for k, v in acc.items():
    print(f'{k}: mean {sum(v) / len(v):.1f} FPS over {len(v)} runs (min {min(v):.1f}, max {max(v):.1f})')
PY
