#!/bin/bash
# The scaling run the driver performs at round end on an 8-GPU node, as one command per N (bench.py contract):
#   scripts/scale_rehearsal.sh [steps] [warmup]
# One rank per GPU over RCCL; every rank tracks its own 1080p stream, the opt-out ReID-gallery all-gather is the only
# collective (fm_gallery_*, csrc/gallery.hip).  Prints one JSON line per N.  No curve has been measured on hardware
# yet (gpurun boxes have one GPU): this script exists so that the day one is, nothing needs editing.
set -e
cd "$(dirname "$0")/.."
STEPS=${1:-300}; WARMUP=${2:-30}
export HSA_ENABLE_IPC_MODE_LEGACY=0
NGPU=$(python -c "import torch; print(torch.cuda.device_count())")
for N in 1 2 4 8; do
    [ "$N" -le "$NGPU" ] || break
    if [ "$N" = 1 ]; then
        python bench.py --gpus 1 --steps "$STEPS" --warmup "$WARMUP" --no-cpu-baseline
    else
        python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port $((29500 + N)) \
            bench.py --gpus "$N" --steps "$STEPS" --warmup "$WARMUP"
    fi
done
