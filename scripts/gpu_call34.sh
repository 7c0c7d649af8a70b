#!/bin/bash
export TMPDIR=/tmp FASTMOT_RANDOM_WEIGHTS=1
O=gpurun_out/c34; mkdir -p $O
for h in 0 1; do
PROFILE_H2D=$h PROFILE_PREFETCH=1 FASTMOT_FLOW_TIMING_VERBOSE=1 timeout 300 python scripts/profile_step.py > $O/tl_h2d$h.txt 2>&1
echo "== H2D=$h"; grep -E "ms/step|flow_predict stages" $O/tl_h2d$h.txt | tail -2; grep -E "sub-stages" $O/tl_h2d$h.txt | tail -1; grep -E "^(mot._step|trk.compute_flow|ctx.flow_predict|ext.postprocess|trk.update|ctx.detect_async_next|ctx.frame_upload_next|ctx.frame_promote_next|det.postprocess|trk.apply_kalman)" $O/tl_h2d$h.txt
done
