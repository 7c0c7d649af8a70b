export TMPDIR=/tmp FASTMOT_RANDOM_WEIGHTS=1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04aa; mkdir -p $O; cd $R
timeout 300 python -m pytest tests/test_conv_gpu.py -m gpu -q -x -k "osnet_fused_and_arena" 2>&1 | tail -15 > $O/tests.txt; cat $O/tests.txt
