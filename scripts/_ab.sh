export FASTMOT_RANDOM_WEIGHTS=1
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_conv_gpu.py tests/test_assoc_gpu.py tests/test_detect_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x 2>&1 | tail -2
for i in 1 2; do
for v in eager new; do
  echo "$v: $(FASTMOT_LIB_PATH=$GRAFT_REPO_ROOT/fastmot_amd/build/libfastmot_hip_$v.so timeout 120 python bench.py --no-cpu-baseline --no-variants 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['net_ms_per_frame'], d['config']['stage_ms'])")"
done
done
