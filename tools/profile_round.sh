#!/bin/bash
# usage (GPU box): tools/profile_round.sh <tag>
# bench line, rocprofv3 kernel stats of its warmup + timed region (bench.py --timed-only), the same one frame at a
# time (--in-flight 1: kernels with the GPU to themselves), then one --pmc pass per counter set (HBM fetch, HBM write,
# two SQ sets), each its own run.  Results under gpurun_out/<tag>_*.
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || { echo bench failed; exit 1; }
tools/prof.sh $tag --steps 20 --warmup 5 || { echo prof failed; exit 1; }
tools/prof.sh ${tag}_alone --steps 20 --warmup 5 --in-flight 1 || { echo prof alone failed; exit 1; }
tools/pmc.sh $tag "FETCH_SIZE" "WRITE_SIZE" \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
  "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA" || exit 1
python tools/pmc_sum.py gpurun_out/pmc_${tag}_3.csv gpurun_out/pmc_${tag}_4.csv > gpurun_out/${tag}_pmc_sq_summary.txt
python tools/traffic.py gpurun_out/pmc_${tag}_1.csv gpurun_out/pmc_${tag}_2.csv gpurun_out/${tag}_traffic.json gpurun_out/pmc_${tag}_3.csv > /dev/null
echo done
