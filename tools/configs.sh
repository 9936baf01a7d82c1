#!/bin/bash
# BASELINE.json's other configurations on one GPU: one bench line each -> gpurun_out/configs.jsonl
cd $GRAFT_REPO_ROOT
: > gpurun_out/configs.jsonl
run() { timeout -k 10 280 python bench.py --no-cpu-baseline "$@" 2>/dev/null | grep '"metric"' >> gpurun_out/configs.jsonl || echo "FAILED: $*"; }
run --scene cornell_box --width 1920 --height 1080 --steps 16 --warmup 2
run --scene mesh_monkey --width 3840 --height 2160 --steps 16 --warmup 2
run --scene torus --width 3840 --height 2160 --steps 16 --warmup 2
run --scene csg --width 3840 --height 2160 --steps 16 --warmup 2
run --scene three-spheres --width 3840 --height 2160 --steps 16 --warmup 2
run --scene room --width 3840 --height 2160 --steps 8 --warmup 1
run --scene complex_v2 --width 3840 --height 2160 --steps 8 --warmup 1
run --scene fog_light --width 7680 --height 4320 --samples 16 --steps 4 --warmup 1
python - <<'PY'
import json
for l in open("gpurun_out/configs.jsonl"):
    j = json.loads(l)
    a = j["one_frame_at_a_time"]
    print("%-70s %8.0f Mrays/s %9.2f ms/frame | one frame at a time %8.0f Mrays/s %9.2f ms  kernels %s" % (j["config"]["workload"], j["value"], j["ms_per_step"], a["value"], a["ms_per_step"], {k: round(v, 2) for k, v in a["kernel_ms_per_step"].items()}))
PY
