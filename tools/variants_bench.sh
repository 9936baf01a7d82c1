#!/bin/bash
# usage (on the GPU box): tools/variants_bench.sh [bench args]   -> one line per library under variants/ and the main build
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for lib in raytracer_amd/librt_amd.so variants/*.so; do
  out=$(RT_AMD_LIB=$PWD/$lib timeout -k 10 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null) || { echo "$lib failed"; exit 1; }
  python - "$lib" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
s = d["stage_ms_per_step"]
print("%-28s %7.1f Mrays/s  %.3f ms (plain %.3f)  int %.3f mesh %.3f shade %.3f tail %.3f" % (
    sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], d["uninstrumented"]["ms_per_step"], s["intersect"], s["mesh"], s["shade"], s["tail"]))
PY
done
