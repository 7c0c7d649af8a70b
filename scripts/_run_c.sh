export TMPDIR=/tmp FASTMOT_RANDOM_WEIGHTS=1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04m; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_conv_gpu.py tests/test_detect_gpu.py -m gpu -q -x -k "fused_csp or filter_dets" 2>&1 | tail -15 > $O/tests.txt; cat $O/tests.txt
cd /tmp && rm -rf /tmp/tr && rocprofv3 --kernel-trace -d /tmp/tr -o t -- python $R/scripts/trace_net.py 0 > /dev/null 2>&1
cd $R && python scripts/layer_roofline.py /tmp/tr > $O/yolo_layer_roofline.txt 2>&1; head -12 $O/yolo_layer_roofline.txt | cut -c1-150; tail -3 $O/yolo_layer_roofline.txt
timeout 600 python bench.py --no-cpu-baseline > $O/bench_n1.json 2> $O/bench_n1.err; tail -2 $O/bench_n1.err; cut -c1-200 $O/bench_n1.json
FASTMOT_CSPSTAGE=0 timeout 600 python bench.py --no-cpu-baseline > $O/bench_nocsp.json 2> /dev/null; cut -c1-200 $O/bench_nocsp.json
