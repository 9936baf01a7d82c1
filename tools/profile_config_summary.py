"""One configuration's profile as text: python tools/profile_config_summary.py <tag> [gpurun_out] > profiles/<tag>_summary.txt
Reads what tools/profile_config.sh left: the bench lines (one frame at a time, frames in flight), rocprofv3 kernel stats,
per-dispatch resources, HBM traffic and SQ counters; prints a roofline line per kernel (bench.py's byte models)."""
import csv, json, os, sys
tag = sys.argv[1]; d = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out"
P = lambda s: os.path.join(d, "%s_%s" % (tag, s))
b = json.loads([l for l in open(P("bench.json")) if l.startswith("{")][-1])
f = json.loads([l for l in open(P("bench_in_flight.json")) if l.startswith("{")][-1])
tr = json.load(open(P("traffic.json")))
print("%s   (tools/profile_config.sh; one MI355X)" % b["config"]["workload"])
print("one frame at a time: %.3f ms per frame, %.0f Mrays/s;  %d frames in flight: %.3f ms, %.0f Mrays/s;  %.2f M rays per frame" % (
    b["ms_per_step"], b["value"], f["config"]["frames_in_flight"], f["ms_per_step"], f["value"], b["config"]["rays_per_frame"] / 1e6))
print("whole frame (SURVEY 8d: 264 B per ray + 24 B per pixel sample): %.1f %% of 8 TB/s one frame at a time, %.1f %% in flight" % (
    100 * b["frame_hbm"]["frac"], 100 * f["frame_hbm"]["frac"]))
print()
print("rocprofv3 --kernel-trace --stats of bench.py --timed-only --in-flight 1 (warmup + timed frames):")
for r in csv.DictReader(open(P("alone_kernel_stats.csv"))):
    n = r["Name"].replace("void ", "").split("(")[0]
    if n.startswith("k_"):
        print("  %-46s calls %5s  average %9.1f us  total %8.2f ms  %5.1f %%" % (n[:46], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, float(r["Percentage"])))
print()
print(open(P("kernel_resources.txt")).read())
print("roofline per kernel (HIP events of the one-frame-at-a-time bench leg; algorithmic bytes: bench.py's models; traffic: FETCH_SIZE x 2 + WRITE_SIZE of separate --pmc passes, per launch):")
for k in [b["roofline"]] + b["kernels"]:
    if not k["launches"] or not k["ms_per_step"]:
        continue
    name = k["kernel"]
    t = tr.get(name + "_hbm_bytes_per_launch") or tr.get(name + "_grid_hbm_bytes_per_launch")
    alg = k["algorithmic_bytes_per_ray"] * (k["rays_per_launch"] or 0)
    lanes = tr.get(name + "_valu_lanes_per_instruction") or tr.get(name + "_grid_valu_lanes_per_instruction")
    insts = tr.get(name + "_valu_wave_instructions_per_frame") or tr.get(name + "_grid_valu_wave_instructions_per_frame")
    busy = tr.get(name + "_valu_busy_per_wave_cycle") or tr.get(name + "_grid_valu_busy_per_wave_cycle")
    line = "  %-11s %6.3f ms per frame, %3d launches of %7.1f us, %8.0f rays per launch x %3d B = %6.1f MB algorithmic -> %6.0f GB/s = %.3f of HBM peak" % (
        name, k["ms_per_step"], k["launches"] // b["steps"], k["avg_launch_ms"] * 1e3, k["rays_per_launch"], k["algorithmic_bytes_per_ray"], alg / 1e6, k["achieved"], k["frac"])
    if t:
        line += "; counters %.1f MB per launch (x%.2f)" % (t / 1e6, t / alg if alg else 0)
    print(line)
    if insts:
        print("              VALU: %.0f M wave instructions per frame = %.0f %% of the 614 G/s issue peak while it runs, %.1f of 64 lanes per instruction, VALU busy %.0f %% of wave cycles" % (
            insts / 1e6, 100 * insts / (k["ms_per_step"] * 1e-3) / 614e9, lanes or 0, 100 * (busy or 0)))
print()
print("SQ counters (sums over the profiled launches):")
keep = ("SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAIT_ANY", "SQ_INSTS_SALU", "SQ_INSTS_LDS",
        "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_WAIT_INST_LDS")
for l in open(P("pmc_sq_summary.txt")):
    if any((" " + c + " ") in l for c in keep):
        print(l.rstrip())
