#!/bin/bash
# usage (GPU box): tools/shard_kernels.sh <lib> [N] -> per-kernel average durations of shard 0 of N rendered alone (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
lib=$1; n=${2:-8}; tag=$(basename $lib .so)
export GPU_MAX_HW_QUEUES=8 SHARD_N=$n SHARD_F=1 RT_AMD_LIB=$PWD/$lib
rm -rf gpurun_out/prof_sk_$tag
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_sk_$tag -- python3 tools/shard_overlap.py > gpurun_out/sk_$tag.log 2>&1 || { echo failed; exit 1; }
f=$(find gpurun_out/prof_sk_$tag -name "*kernel_trace.csv" | head -1)
echo "== $tag: $(grep shard gpurun_out/sk_$tag.log)"
python tools/pass_times.py $f
