#!/bin/bash
# usage (GPU box): tools/ab.sh <out tag> "<ENV=.. of leg a>" "<ENV=.. of leg b>" [repeats] -- <bench.py args>
# Two settings of an environment knob on the same box and build, interleaved -> gpurun_out/ab_<tag>.txt
tag=$1; ea=$2; eb=$3; reps=${4:-2}; shift 4; [ "$1" = "--" ] && shift
cd $GRAFT_REPO_ROOT
out=gpurun_out/ab_$tag.txt
: > $out
run() { # label, env, args
  label=$1; shift; envv=$1; shift
  line=$(env $envv timeout -k 10 280 python bench.py --no-cpu-baseline "$@" 2>/dev/null | grep '"metric"') || { echo "FAILED: $label $*" >> $out; return; }
  python - "$label" <<PY >> $out
import json, sys
j = json.loads('''$line''')
a = j["one_frame_at_a_time"]
print("%-28s %-52s %8.0f Mrays/s %8.3f ms/frame (uninstrumented %8.0f) | alone %8.0f Mrays/s %8.3f ms  %s" % (sys.argv[1], j["config"]["workload"][:52], j["value"], j["ms_per_step"], j.get("uninstrumented", {}).get("value", 0), a["value"], a["ms_per_step"], {k: round(v, 3) for k, v in a["kernel_ms_per_step"].items()}))
PY
}
for rep in $(seq $reps); do
  run "$ea" "$ea" "$@"
  run "$eb" "$eb" "$@"
done
cat $out
