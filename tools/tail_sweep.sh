#!/bin/bash
# usage (GPU box): tools/tail_sweep.sh "<env assignments>" ...   one bench line per setting: in flight / one frame at a time
cd $GRAFT_REPO_ROOT
for setting in "$@"; do
  out=$(env $setting timeout -k 10 150 python bench.py --no-cpu-baseline --steps 12 --warmup 3 $BENCH_ARGS 2>/dev/null) || { echo "$setting failed"; exit 1; }
  python - "$setting" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[2]); a = d["one_frame_at_a_time"]
print("%-44s in flight %7.0f Mrays/s %.3f ms | alone %.3f ms %6.0f Mrays/s | alone, per kernel: %s" % (
    sys.argv[1], d["value"], d["ms_per_step"], a["ms_per_step"], a["value"], " ".join("%s %.3f" % kv for kv in a["kernel_ms_per_step"].items())), flush=True)
PY
done
