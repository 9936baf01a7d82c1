"""Durations of the last k_intersect launches in a rocprofv3 kernel trace, in launch order:
python tools/isect_dispatches.py <kernel_trace.csv> [how many]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
isect = [r for r in rows if "k_intersect" in r["Kernel_Name"]]
print(" ".join("%.0f" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in isect[-n:]))
