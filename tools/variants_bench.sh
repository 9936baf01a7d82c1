#!/bin/bash
# usage (on the GPU box): tools/variants_bench.sh [bench args]   -> one line per library under variants/ and the main build
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for lib in raytracer_amd/librt_amd.so variants/*.so; do
  out=$(RT_AMD_LIB=$PWD/$lib timeout -k 10 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null) || { echo "$lib failed"; exit 1; }
  python - "$lib" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
a = d["one_frame_at_a_time"]
print("%-28s %7.1f Mrays/s  %.3f ms (plain %.3f) | alone %.3f ms: %s" % (
    sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], d["uninstrumented"]["ms_per_step"], a["ms_per_step"],
    " ".join("%s %.3f" % kv for kv in a["kernel_ms_per_step"].items())))
PY
done
