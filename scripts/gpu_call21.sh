#!/bin/bash
# timeline of a pipelined step (prefetch) vs sequential, then the rest of the GPU suite after the flow.py fix
O=gpurun_out/c21; mkdir -p $O
export TMPDIR=/tmp
PROFILE_PREFETCH=1 FASTMOT_FLOW_TIMING_VERBOSE=1 timeout 300 python scripts/profile_step.py > $O/tl_prefetch.txt 2>&1
PROFILE_PREFETCH=0 FASTMOT_FLOW_TIMING_VERBOSE=1 timeout 300 python scripts/profile_step.py > $O/tl_seq.txt 2>&1
PROFILE_PREFETCH=1 FASTMOT_LK_LDS=0 FASTMOT_FLOW_TIMING_VERBOSE=1 timeout 300 python scripts/profile_step.py > $O/tl_prefetch_nolds.txt 2>&1
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 > $O/pytest.txt
grep -E "ms/step|flow_predict stages|sub-stages" $O/tl_prefetch.txt | tail -4; grep -E "^(mot|det|trk|ext|ctx)" $O/tl_prefetch.txt
echo ---; grep -E "ms/step|flow_predict stages|sub-stages" $O/tl_prefetch_nolds.txt | tail -4
tail -3 $O/pytest.txt
