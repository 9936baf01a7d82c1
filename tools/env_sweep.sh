#!/bin/bash
# usage (GPU box): tools/env_sweep.sh <tag> "<ENV=a>" "<ENV=b>" ... -- <bench.py args>   one bench line per setting, twice -> gpurun_out/sweep_<tag>.txt
tag=$1; shift
envs=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do envs+=("$1"); shift; done
[ "$1" = "--" ] && shift
cd $GRAFT_REPO_ROOT
out=gpurun_out/sweep_$tag.txt
: > $out
for rep in 1 2; do
for e in "${envs[@]}"; do
  line=$(env $e timeout -k 10 280 python bench.py --no-cpu-baseline "$@" 2>/dev/null | grep '"metric"') || { echo "FAILED: $e" >> $out; continue; }
  python - "$e" <<PY >> $out
import json, sys
j = json.loads('''$line''')
a = j["one_frame_at_a_time"]
print("%-28s %8.0f Mrays/s %8.3f ms/frame (uninstrumented %8.0f) | alone %8.3f ms  %s" % (sys.argv[1], j["value"], j["ms_per_step"], j.get("uninstrumented", {}).get("value", 0), a["ms_per_step"], {k: round(v, 3) for k, v in a["kernel_ms_per_step"].items()}))
PY
done
done
cat $out
