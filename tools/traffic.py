"""HBM traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE), written to profiles/traffic.json.

    python tools/traffic.py <fetch.csv> <write.csv> [out.json [sq.csv]]

FETCH_SIZE and WRITE_SIZE are reported in KiB-like units of 1 KB by rocprofv3; on gfx950
FETCH_SIZE counts 128-byte requests as 64 bytes, so it is doubled (MI355X_MICROARCH.md, HBM)."""
import csv, json, sys, collections

def per_kernel(path, counter):
    tot = collections.defaultdict(float); n = collections.Counter()
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
            tot[k] += float(r["Counter_Value"]); n[k] += 1
    return tot, n

fetch, nf = per_kernel(sys.argv[1], "FETCH_SIZE")
write, nw = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"units": "bytes per launch, averaged over the launches of one bench run (warmup included)",
       "correction": "FETCH_SIZE x 1024 x 2 (gfx950 half-count) + WRITE_SIZE x 1024"}
for k in sorted(fetch):
    if not k.startswith("k_"):
        continue
    f = fetch[k] / nf[k] * 1024.0 * 2.0
    w = write.get(k, 0.0) / max(1, nw.get(k, 1)) * 1024.0
    out[k + "_hbm_bytes_per_launch"] = f + w
    out[k + "_launches"] = nf[k]
    out[k + "_fetch_bytes_per_launch"] = f
    out[k + "_write_bytes_per_launch"] = w
# optional 4th argument: the SQ pass (SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU, SQ_THREAD_CYCLES_VALU, SQ_WAVE_CYCLES, SQ_WAVES per
# launch) -> wave-level VALU instructions per frame and lanes active per instruction, for the FP64-VALU side of the roofline
if len(sys.argv) > 4:
    sq = {}
    for c in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVE_CYCLES"):
        sq[c], _ = per_kernel(sys.argv[4], c)
    # tools/pmc.sh profiles `bench.py --timed-only --in-flight 1 --steps 1 --warmup 1`: two frames
    frames = int(__import__("os").environ.get("PMC_FRAMES", "2"))
    out["valu_frames_profiled"] = frames
    for k in sorted(sq["SQ_INSTS_VALU"]):
        if not k.startswith("k_") or sq["SQ_ACTIVE_INST_VALU"].get(k, 0.0) <= 0.0:
            continue
        out[k + "_valu_wave_instructions_per_frame"] = sq["SQ_INSTS_VALU"][k] / frames
        out[k + "_valu_lanes_per_instruction"] = sq["SQ_THREAD_CYCLES_VALU"][k] / sq["SQ_ACTIVE_INST_VALU"][k]
        out[k + "_valu_busy_per_wave_cycle"] = sq["SQ_ACTIVE_INST_VALU"][k] / sq["SQ_WAVE_CYCLES"][k]
sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), ".."))
import bench
out["sources_sha256"] = bench.kernel_sources_sha()
json.dump(out, open(sys.argv[3] if len(sys.argv) > 3 and sys.argv[3] else "profiles/traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
