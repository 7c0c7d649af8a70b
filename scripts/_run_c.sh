export TMPDIR=/tmp FASTMOT_RANDOM_WEIGHTS=1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04z; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_fullsize_gpu.py tests/test_e2e_parity_gpu.py -m gpu -q 2>&1 | tail -8 > $O/tests.txt; cat $O/tests.txt
timeout 600 python bench.py --no-cpu-baseline --no-variants > $O/bench_n1.json 2> $O/bench_n1.err; cut -c1-200 $O/bench_n1.json
FASTMOT_GATEDCONV=0 timeout 600 python bench.py --no-cpu-baseline --no-variants > $O/bench_n1_off.json 2> $O/bench_n1_off.err; cut -c1-200 $O/bench_n1_off.json
timeout 600 python bench.py --no-cpu-baseline --no-variants > $O/bench_n1_b.json 2> $O/bench_n1.err; cut -c1-200 $O/bench_n1_b.json
cd /tmp && rm -rf /tmp/tro && rocprofv3 --kernel-trace -d /tmp/tro -o t -- python $R/scripts/trace_net.py 1 50 > /dev/null 2>&1
cd $R && python scripts/rocpd_dispatches.py "$(find /tmp/tro -name '*.db' | head -1)" 40 > $O/osnet_b50_dispatches.txt 2>&1; grep -E "gatedconv|span" $O/osnet_b50_dispatches.txt | tail -8 | cut -c1-150
