#!/bin/bash
export TMPDIR=/tmp FASTMOT_RANDOM_WEIGHTS=1
O=$GRAFT_REPO_ROOT/gpurun_out/c26; mkdir -p $O
cd /tmp
for b in 50 25; do
rm -rf /tmp/tr26_$b && rocprofv3 --kernel-trace --stats -d /tmp/tr26_$b -o t -- python $GRAFT_REPO_ROOT/scripts/trace_net.py 1 $b > /tmp/tr26.log 2>&1
db=$(find /tmp/tr26_$b -name '*.db' | head -1)
python $GRAFT_REPO_ROOT/scripts/rocpd_dispatches.py $db 32 > $O/osnet_b$b.txt 2>&1
done
rm -rf /tmp/prof26 && rocprofv3 --kernel-trace --stats -d /tmp/prof26 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-variants > $O/bench_prof.json 2>/dev/null
cd $GRAFT_REPO_ROOT && python scripts/rocpd_summary.py "$(find /tmp/prof26 -name '*.db' | head -1)" > $O/kernel_stats.txt 2>&1
cat $O/osnet_b25.txt; head -30 $O/kernel_stats.txt
