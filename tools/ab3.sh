#!/bin/bash
# usage (GPU box): tools/ab3.sh <tag> <repeats> "<lib a> <lib b> ..." -- <bench args>   (a library "main" = the main build)
tag=$1; reps=$2; libs=$3; shift 3; [ "$1" = "--" ] && shift
cd $GRAFT_REPO_ROOT
out=gpurun_out/ab_$tag.txt
: > $out
for rep in $(seq $reps); do
  for lib in $libs; do
    if [ $lib = main ]; then e="RT_MAIN=1"; else e="RT_AMD_LIB=$PWD/variants/librt_$lib.so"; fi
    line=$(env $e timeout -k 10 280 python bench.py --no-cpu-baseline "$@" 2>/dev/null | grep '"metric"') || { echo "FAILED: $lib" >> $out; continue; }
    python - "$lib" <<PY >> $out
import json, sys
j = json.loads('''$line''')
a = j["one_frame_at_a_time"]
print("%-28s %-52s %8.0f Mrays/s %8.3f ms/frame (uninstrumented %8.0f) | alone %8.0f Mrays/s %8.3f ms  %s" % (sys.argv[1], j["config"]["workload"][:52].replace(" ", "_"), j["value"], j["ms_per_step"], j.get("uninstrumented", {}).get("value", 0), a["value"], a["ms_per_step"], {k: round(v, 3) for k, v in a["kernel_ms_per_step"].items()}))
PY
  done
done
python tools/ab_table.py $out
