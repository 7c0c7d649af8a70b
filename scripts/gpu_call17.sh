export FASTMOT_RANDOM_WEIGHTS=1
python scripts/stress_lk6.py 600 2>&1 | grep -E "hammer=(prefix\[:32\]|only litechain)" 
FASTMOT_LK_LDS=0 python scripts/stress_lk6.py 600 2>&1 | grep -E "hammer=(prefix\[:32\]|only litechain)" | sed 's/^/LDS=0 /'
for v in 150000 0; do echo "== LK_LDS=$v"; FASTMOT_LK_LDS=$v FASTMOT_FLOW_TIMING_VERBOSE=1 python scripts/profile_step.py 2>&1 | grep -E "ms/step|flow_predict stages|sub-stages" | tail -3; done
