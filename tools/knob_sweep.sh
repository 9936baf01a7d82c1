#!/bin/bash
# usage (on the GPU box): tools/knob_sweep.sh "VAR=a VAR=b 'VAR=c OTHER=d' ..." [bench args]
# One bench line per setting of the library's tuning knobs (environment variables read once per process).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
settings=$1; shift
eval "set_list=($settings)"
for s in "" "${set_list[@]}"; do
  out=$(env $s timeout -k 10 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null) || { echo "$s failed"; exit 1; }
  python - "${s:-default}" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
a = d["one_frame_at_a_time"]
print("%-36s %7.1f Mrays/s  %.3f ms (plain %.3f)  alone %.3f ms: %s" % (
    sys.argv[1], d["value"], d["ms_per_step"], d["uninstrumented"]["ms_per_step"], a["ms_per_step"],
    " ".join("%s %.3f" % kv for kv in a["kernel_ms_per_step"].items())))
PY
done
