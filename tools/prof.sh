#!/bin/bash
# usage: tools/prof.sh <tag> [bench args...]  -> gpurun_out/prof_<tag>/ kernel stats (run on the GPU box)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=${BENCH_HW_QUEUES:-8} # as bench.py sets it: the profiler's preload may start the HIP runtime before bench.py runs
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py --no-cpu-baseline --timed-only "$@" > gpurun_out/prof_$tag.log 2>&1
rc=$?
f=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/prof_${tag}_kernel_stats.csv
cut -d, -f1-4 $f | cut -c1-60,200-400 | head -8
grep '"metric"' gpurun_out/prof_$tag.log | cut -c1-200
exit $rc
