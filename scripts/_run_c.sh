export TMPDIR=/tmp FASTMOT_RANDOM_WEIGHTS=1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04x; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_detect_gpu.py tests/test_detector_chain_gpu.py -m gpu -q -x 2>&1 | tail -5 > $O/tests.txt; cat $O/tests.txt
timeout 600 python bench.py --no-cpu-baseline > $O/bench_n1.json 2> $O/bench_n1.err; cut -c1-200 $O/bench_n1.json
