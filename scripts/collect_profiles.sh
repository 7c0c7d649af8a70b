#!/bin/bash
# Regenerates the evidence under profiles/ for one round on the GPU box (run through gpurun from the repo root):
#   bash scripts/collect_profiles.sh r03 [quick]      (quick: no stand-alone network replays, no PMC passes, CPU baseline on config[1] only)
# -> gpurun_out/<tag>/: GPU test summary, bench lines (default command line, the driver's --steps 20 --warmup 5, configs 2
#    and 4), rocprofv3 kernel stats of the bench command, per-layer roofline table of the detector, OSNet dispatch list,
#    PMC traffic passes (separate rocprofv3 --pmc runs, MI355X_MICROARCH.md).  Copy what is to be judged into profiles/.
TAG=${1:-r06}; QUICK=${2:-}
export TMPDIR=/tmp FASTMOT_RANDOM_WEIGHTS=1
R=$GRAFT_REPO_ROOT; [ -n "$R" ] || R=$(pwd)
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q --durations=15 2>&1 | tail -30 > $O/pytest_gpu.txt; tail -3 $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; cut -c1-180 $O/bench_n1.json
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_cmdline.json 2> $O/bench_driver.err; cut -c1-180 $O/bench_driver_cmdline.json
NOCPU=; [ -z "$QUICK" ] || NOCPU=--no-cpu-baseline
timeout 900 python bench.py --config 2 $NOCPU > $O/bench_config2.json 2> $O/bench_config2.err; cut -c1-230 $O/bench_config2.json
timeout 900 python bench.py --config 4 --steps 60 --warmup 10 $NOCPU > $O/bench_config4.json 2> $O/bench_config4.err; cut -c1-230 $O/bench_config4.json
# stage boundaries of the pipelined step on the GPU's clock, hipGraph replays included (fm_trace_*)
timeout 300 python scripts/trace_pipeline.py --show 3 > $O/pipeline_trace.txt 2> /dev/null; head -1 $O/pipeline_trace.txt
# kernel stats of the bench command itself (the durations roofline.achieved must agree with).  FASTMOT_GRAPHS=0: the layers
# are launched one by one -- rocprofv3's tool crashes inside hipGraphLaunch of this pipeline (ROCm 7.2), and per-kernel
# durations are what is wanted here anyway
cd /tmp && rm -rf /tmp/kt_$TAG && FASTMOT_GRAPHS=0 timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt_$TAG -o b -- python $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-variants > $O/bench_under_rocprof.json 2> /dev/null
cd $R && python scripts/rocpd_summary.py "$(find /tmp/kt_$TAG -name '*.db' | head -1)" > $O/bench_kernel_stats.txt 2>&1; head -30 $O/bench_kernel_stats.txt | cut -c1-160
if [ -z "$QUICK" ]; then
# the same for config[4] (4K, 300 tracks): the KLT keypoint kernels at the size DESIGN 3c is about
cd /tmp && rm -rf /tmp/kt4_$TAG && FASTMOT_GRAPHS=0 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt4_$TAG -o b -- python $R/bench.py --config 4 --steps 20 --warmup 5 --no-cpu-baseline --no-variants > /dev/null 2>&1
cd $R && python scripts/rocpd_summary.py "$(find /tmp/kt4_$TAG -name '*.db' | head -1)" > $O/bench_config4_kernel_stats.txt 2>&1; grep -E "gftt|eig|prepare|fast_|lk_pair" $O/bench_config4_kernel_stats.txt | cut -c1-160
fi
if [ -z "$QUICK" ]; then
# stand-alone replays: per-layer roofline of the detector, dispatch list of the ReID network
cd /tmp && rm -rf /tmp/tr_$TAG && rocprofv3 --kernel-trace -d /tmp/tr_$TAG -o t -- python $R/scripts/trace_net.py 0 > /dev/null 2>&1
cd $R && python scripts/layer_roofline.py /tmp/tr_$TAG > $O/yolo_layer_roofline.txt 2>&1; tail -3 $O/yolo_layer_roofline.txt
# the same for the detectors of configs [2] and [4], and for YOLOv4 @ 608 without the DMA-fed kernel (the A/B of DESIGN 4c)
for m in YOLOv4P6_1280 YOLOv4CSP_640; do
  cd /tmp && rm -rf /tmp/tr_${TAG}_$m && rocprofv3 --kernel-trace -d /tmp/tr_${TAG}_$m -o t -- python $R/scripts/trace_net.py 0 $m > /dev/null 2>&1
  cd $R && python scripts/layer_roofline.py /tmp/tr_${TAG}_$m $m > $O/layers_$m.txt 2>&1; echo "$m: $(tail -2 $O/layers_$m.txt | head -1)"
done
for m in YOLOv4_608 YOLOv4P6_1280 YOLOv4CSP_640; do
  cd /tmp && rm -rf /tmp/tr0_${TAG}_$m && FASTMOT_CONVD=0 rocprofv3 --kernel-trace -d /tmp/tr0_${TAG}_$m -o t -- python $R/scripts/trace_net.py 0 $m > /dev/null 2>&1
  cd $R && FASTMOT_CONVD=0 python scripts/layer_roofline.py /tmp/tr0_${TAG}_$m $m > $O/layers_${m}_convd0.txt 2>&1; echo "$m without convd: $(tail -2 $O/layers_${m}_convd0.txt | head -1)"
done
cd /tmp && rm -rf /tmp/tro_$TAG && rocprofv3 --kernel-trace -d /tmp/tro_$TAG -o t -- python $R/scripts/trace_net.py 1 50 > /dev/null 2>&1
cd $R && python scripts/rocpd_dispatches.py "$(find /tmp/tro_$TAG -name '*.db' | head -1)" 44 > $O/osnet_b50_dispatches.txt 2>&1; tail -3 $O/osnet_b50_dispatches.txt
# phases of the fused OSNet tail (profiling build of ostail.hip: bash scripts/build_timing_lib.sh ostail.hip -DFM_OST_TIMING before the call)
[ -f fastmot_amd/libfastmot_hip_timing.so ] && FASTMOT_LIB_PATH=$R/fastmot_amd/libfastmot_hip_timing.so timeout 200 python scripts/ost_timing.py 50 2>&1 | grep -v SEEDED > $O/osnet_tail_phases.txt
fi
if [ -z "$QUICK" ]; then
    bash scripts/collect_pmc.sh > $O/pmc.log 2>&1; cp gpurun_out/pmc_conv.json gpurun_out/pmc_FETCH_SIZE.txt gpurun_out/pmc_WRITE_SIZE.txt gpurun_out/pmc_sq_yolo.txt gpurun_out/pmc_layers.txt $O/ 2>/dev/null; tail -3 $O/pmc.log | cut -c1-400
fi
