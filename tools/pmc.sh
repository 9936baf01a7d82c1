#!/bin/bash
# usage: tools/pmc.sh <tag> "<counters of pass 1>" ["<counters of pass 2>" ...]
# One rocprofv3 --pmc run per counter set (no trace flags beside it), each under its own hard limit.
# Output: gpurun_out/pmc_<tag>_<n>.csv (per-dispatch counter values).  Run on the GPU box.
tag=$1; shift
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=${BENCH_HW_QUEUES:-8} # as bench.py sets it: the profiler's preload may start the HIP runtime before bench.py runs
cd $GRAFT_REPO_ROOT
n=0
for set in "$@"; do
  n=$((n+1))
  out=gpurun_out/pmc_${tag}_$n
  timeout -s KILL 150 rocprofv3 --pmc $set --output-format csv -d $out -- python3 bench.py --no-cpu-baseline --timed-only --in-flight 1 --steps 1 --warmup 1 $PMC_ARGS > $out.log 2>&1 || { echo "pass $n failed"; tail -5 $out.log; exit 1; }
  f=$(find $out -name "*counter_collection.csv" | head -1)
  cp $f $out.csv
  echo "pass $n: $(wc -l < $out.csv) rows -> $out.csv"
done
