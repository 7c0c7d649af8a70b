export FASTMOT_RANDOM_WEIGHTS=1
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
for v in old new; do
  echo "$v: $(FASTMOT_LIB_PATH=$GRAFT_REPO_ROOT/fastmot_amd/build/libfastmot_hip_$v.so timeout 120 python bench.py --no-cpu-baseline --no-variants 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['net_ms_per_frame'], d['config']['stage_ms'])")"
done
done
