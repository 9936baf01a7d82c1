#!/bin/bash
# usage (GPU box): tools/flight_sweep.sh <scene> <width> <height>   -> Mrays/s by hardware queues x frames in flight
cd $GRAFT_REPO_ROOT
for q in 4 8; do for f in 2 3 4; do
  out=$(BENCH_HW_QUEUES=$q timeout -k 10 120 python bench.py --no-cpu-baseline --scene $1 --width $2 --height $3 --steps 8 --warmup 2 --in-flight $f 2>/dev/null) || { echo failed; exit 1; }
  python - "$1 queues $q in flight $f" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[2]); print("%-40s %7.1f Mrays/s %.3f ms" % (sys.argv[1], d["value"], d["ms_per_step"]))
PY
done; done
