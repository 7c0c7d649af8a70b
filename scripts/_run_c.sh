export TMPDIR=/tmp FASTMOT_RANDOM_WEIGHTS=1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04c; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_detector_chain_gpu.py tests/test_onnx_reader.py -m gpu -q -s 2>&1 | grep "^config\|^E  \|passed\|failed\|^tests.*Error" | cut -c1-1200 > $O/chain.txt; cat $O/chain.txt
