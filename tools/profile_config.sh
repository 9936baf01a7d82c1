#!/bin/bash
# usage (GPU box): tools/profile_config.sh <tag> <bench args: --scene S --width W --height H [--samples N] --steps K --warmup W>
# One BASELINE configuration, one frame at a time (--in-flight 1): the bench line, rocprofv3 kernel stats and trace
# (VGPRs / scratch / LDS per dispatch), then one --pmc pass per counter set (HBM fetch, HBM write, two SQ sets), each its
# own run.  Results under gpurun_out/<tag>_*; tools/profile_config_summary.py turns them into profiles/<tag>_summary.txt.
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=${BENCH_HW_QUEUES:-8}
timeout -k 10 300 python bench.py --no-cpu-baseline --in-flight 1 "$@" > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || { echo bench failed; tail -3 gpurun_out/${tag}_bench.err; exit 1; }
timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > gpurun_out/${tag}_bench_in_flight.json 2>> gpurun_out/${tag}_bench.err || { echo bench in flight failed; exit 1; }
rm -rf gpurun_out/prof_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py --no-cpu-baseline --timed-only --in-flight 1 "$@" > gpurun_out/prof_$tag.log 2>&1 || { echo prof failed; exit 1; }
cp $(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_alone_kernel_stats.csv
python - $(find gpurun_out/prof_$tag -name "*kernel_trace.csv" | head -1) > gpurun_out/${tag}_kernel_resources.txt <<'PY'
import csv, sys, collections
seen = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if not k.lstrip("void ").startswith("k_"):
        continue
    key = k.split("(")[0].replace("void ", "")
    seen.setdefault(key, (r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["Scratch_Size"], r["LDS_Block_Size"], r["Workgroup_Size_X"]))
print("kernel: arch VGPRs, accum VGPRs, SGPRs, scratch bytes per lane, LDS bytes per workgroup, workgroup size")
for k, v in seen.items():
    print("%-50s %s" % (k, ", ".join(v)))
PY
PMC_ARGS="$(echo "$@" | sed -E 's/--steps [0-9]+//; s/--warmup [0-9]+//')" tools/pmc.sh $tag "FETCH_SIZE" "WRITE_SIZE" \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
  "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA" || exit 1
python tools/pmc_sum.py gpurun_out/pmc_${tag}_3.csv gpurun_out/pmc_${tag}_4.csv > gpurun_out/${tag}_pmc_sq_summary.txt
python tools/traffic.py gpurun_out/pmc_${tag}_1.csv gpurun_out/pmc_${tag}_2.csv gpurun_out/${tag}_traffic.json gpurun_out/pmc_${tag}_3.csv > /dev/null
rm -rf gpurun_out/prof_$tag gpurun_out/pmc_${tag}_[1-4]
echo "$tag done"
