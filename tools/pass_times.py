"""Per-launch durations of one frame from a rocprofv3 kernel trace: python tools/pass_times.py <kernel_trace.csv>
(the last frame of the trace: the launches after the last k_tail but one)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ours = [r for r in rows if r["Kernel_Name"].lstrip("void ").startswith("k_")]
tails = [i for i, r in enumerate(ours) if "k_tail" in r["Kernel_Name"]]
frame = ours[tails[-2] + 1:tails[-1] + 1] if len(tails) >= 2 else ours
t0 = int(frame[0]["Start_Timestamp"])
line = []
for r in frame:
    name = r["Kernel_Name"].replace("void ", "").split("<")[0].split("(")[0]
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    line.append("%s %.0f" % (name, us))
    if name in ("k_shade", "k_tail", "k_tail_grid"):
        print("  ".join(line))
        line = []
if line:
    print("  ".join(line))
print("frame: %.0f us from the first launch to the end of the last" % ((int(frame[-1]["End_Timestamp"]) - t0) / 1e3))
