#!/bin/bash
# Shell helpers for A/B measurements inside ONE gpurun call (boxes of the pool differ by ~10 %: only numbers of the same
# call compare).  Source it:   . scripts/ab_lib.sh <tag>      -> results under gpurun_out/<tag>/
# A variant is "name:ENV=.. ENV=.." (empty env list allowed); a second build of the library is selected with
# FASTMOT_LIB_PATH=$R/fastmot_amd/libfastmot_hip_<name>.so (scripts/build_timing_lib.sh / build_old_lib.sh).
# No function here is named like a coreutils tool (round 5 lost 20 GPU minutes to a function called tr).
export TMPDIR=/tmp FASTMOT_RANDOM_WEIGHTS=1
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-ab}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R

ab_env() { local spec=${1#*:}; [ -n "$spec" ] && echo $spec || echo AB_NONE=1; }
ab_name() { echo ${1%%:*}; }

# ab_tests <pytest args...>: the parity gate in front of a measurement
ab_tests() { timeout 1200 python -m pytest "$@" -m gpu -q -x 2>&1 | tail -4; }

# ab_trace_net <variant> <which: 0 detector | 1 osnet> <model or batch> <dispatches to list> [grep pattern]
ab_trace_net() {
  local v=$1 which=$2 arg=$3 n=$4 pat=${5:-}; local name; name=$(ab_name "$v")
  cd /tmp && rm -rf /tmp/abt_$name && env $(ab_env "$v") timeout 200 rocprofv3 --kernel-trace -d /tmp/abt_$name -o t -- python $R/scripts/trace_net.py $which $arg > /dev/null 2>&1
  cd $R && python scripts/rocpd_dispatches.py "$(find /tmp/abt_$name -name '*.db' | head -1)" $n > $O/disp_${name}_$arg.txt 2>&1
  local extra=""; [ -n "$pat" ] && extra=" | $pat: $(grep -E "$pat" $O/disp_${name}_$arg.txt | sed 's/.*dur= *\([0-9.]*\).*/\1/' | paste -sd' ')"
  echo "$name $arg: $(tail -1 $O/disp_${name}_$arg.txt)$extra"
}

# ab_layers <variant> <model>: per-layer roofline table of a detector (scripts/layer_roofline.py)
ab_layers() {
  local v=$1 m=$2; local name; name=$(ab_name "$v")
  cd /tmp && rm -rf /tmp/abl_$name && env $(ab_env "$v") timeout 200 rocprofv3 --kernel-trace -d /tmp/abl_$name -o t -- python $R/scripts/trace_net.py 0 $m > /dev/null 2>&1
  cd $R && env $(ab_env "$v") python scripts/layer_roofline.py /tmp/abl_$name $m > $O/layers_${m}_$name.txt 2>&1; echo "$name $m: $(tail -2 $O/layers_${m}_$name.txt | head -1)"
}

# ab_bench <rounds> <bench args...> -- <variant> <variant> ...: the variants alternate, <rounds> times
ab_bench() {
  local rounds=$1; shift; local args=(); while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
  for i in $(seq $rounds); do for v in "$@"; do
    local name; name=$(ab_name "$v")
    env $(ab_env "$v") timeout 400 python bench.py "${args[@]}" --no-cpu-baseline --no-variants > $O/bench_${name}_$i.json 2> $O/bench_${name}_$i.err
    python - "$O/bench_${name}_$i.json" "$name" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[2], 'fps', d['value'], 'seq', d.get('sequential_fps'), 'net_ms', d['roofline'].get('net_ms_per_frame'), d['config'].get('stage_ms'))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
  done; done
}
