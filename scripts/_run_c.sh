export TMPDIR=/tmp FASTMOT_RANDOM_WEIGHTS=1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04w; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8 > $O/pytest_gpu.txt; tail -3 $O/pytest_gpu.txt
timeout 600 python bench.py --no-cpu-baseline > $O/bench_n1.json 2> $O/bench_n1.err; cut -c1-200 $O/bench_n1.json
timeout 600 python bench.py --config 4 --steps 60 --warmup 10 --no-cpu-baseline > $O/bench_config4.json 2> $O/bench_config4.err; cut -c1-200 $O/bench_config4.json
