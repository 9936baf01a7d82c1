#!/bin/bash
# the BASELINE configurations that run the CSG variant of k_intersect -> gpurun_out/configs_csg.jsonl
cd $GRAFT_REPO_ROOT
: > gpurun_out/configs_csg.jsonl
run() { timeout -k 10 280 python bench.py --no-cpu-baseline "$@" 2>/dev/null | grep '"metric"' >> gpurun_out/configs_csg.jsonl || echo "FAILED: $*"; }
run --scene csg --width 3840 --height 2160 --steps 5 --warmup 1
run --scene room --width 1920 --height 1080 --steps 3 --warmup 1
run --scene fog_light --width 3840 --height 2160 --samples 4 --steps 2 --warmup 1
python - <<'PY'
import json
for l in open("gpurun_out/configs_csg.jsonl"):
    j = json.loads(l)
    a = j["one_frame_at_a_time"]
    print("%-70s %8.0f Mrays/s %9.2f ms/frame | one frame at a time %8.0f Mrays/s %9.2f ms  kernels %s" % (j["config"]["workload"], j["value"], j["ms_per_step"], a["value"], a["ms_per_step"], {k: round(v, 2) for k, v in a["kernel_ms_per_step"].items()}))
PY
