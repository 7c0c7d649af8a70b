#!/bin/bash
export TMPDIR=/tmp FASTMOT_RANDOM_WEIGHTS=1
mkdir -p gpurun_out; O=gpurun_out; rm -f $O/c6_*.txt
for io in 0 1 2 0 2; do FASTMOT_LK_IO=$io python scripts/stress_lk3.py 600 2>&1 | grep "LK_IO" >> $O/c6_stress.txt; done
cat $O/c6_stress.txt
python -m pytest tests/test_mot_multiclass_gpu.py -m gpu -q --timeout=600 2>&1 | tail -3
python scripts/profile_host.py 2>/dev/null | tail -45 > $O/c6_host_profile.txt; head -40 $O/c6_host_profile.txt | cut -c1-150
