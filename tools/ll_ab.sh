#!/bin/bash
# interval lists in LDS against lists in scratch memory (RT_LDS_LISTS=0), same box, same build -> gpurun_out/ll_ab.txt
cd $GRAFT_REPO_ROOT
out=gpurun_out/ll_ab.txt
: > $out
run() { # label, env, args
  label=$1; shift; envv=$1; shift
  line=$(env $envv timeout -k 10 280 python bench.py --no-cpu-baseline "$@" 2>/dev/null | grep '"metric"') || { echo "FAILED: $label $*" >> $out; return; }
  python - "$label" <<PY >> $out
import json, sys
j = json.loads('''$line''')
a = j["one_frame_at_a_time"]
print("%-10s %-62s %8.0f Mrays/s %9.2f ms/frame | alone %8.0f Mrays/s %9.2f ms  %s" % (sys.argv[1], j["config"]["workload"], j["value"], j["ms_per_step"], a["value"], a["ms_per_step"], {k: round(v, 2) for k, v in a["kernel_ms_per_step"].items()}))
PY
}
for rep in 1 2; do
run scratch RT_LDS_LISTS=0 --scene csg --width 3840 --height 2160 --steps 16 --warmup 2
run lds     RT_LDS_LISTS=1 --scene csg --width 3840 --height 2160 --steps 16 --warmup 2
done
run scratch RT_LDS_LISTS=0 --scene room --width 3840 --height 2160 --steps 4 --warmup 1
run lds     RT_LDS_LISTS=1 --scene room --width 3840 --height 2160 --steps 4 --warmup 1
run scratch RT_LDS_LISTS=0 --scene fog_light --width 3840 --height 2160 --samples 4 --steps 4 --warmup 1
run lds     RT_LDS_LISTS=1 --scene fog_light --width 3840 --height 2160 --samples 4 --steps 4 --warmup 1
cat $out
