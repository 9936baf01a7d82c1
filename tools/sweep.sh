#!/bin/bash
# usage: tools/sweep.sh VAR v1 v2 ...   -> one short bench line per value (run on the GPU box)
var=$1; shift
for v in "$@"; do
  export $var=$v
  timeout -k 10 120 python bench.py --no-cpu-baseline --steps 8 --warmup 2 > gpurun_out/sweep.log 2>&1 || { echo "$var=$v FAILED"; tail -3 gpurun_out/sweep.log; exit 1; }
  python - <<PY
import json
l=[x for x in open("gpurun_out/sweep.log") if x.startswith("{")][-1]
j=json.loads(l)
print("$var=$v", "Mrays/s %.0f" % j["value"], "ms/step %.3f" % j["ms_per_step"], " ".join("%s %.3f" % kv for kv in j["one_frame_at_a_time"]["kernel_ms_per_step"].items()))
PY
done
