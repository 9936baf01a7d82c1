"""Per-dispatch durations (us) of one kernel from a rocprofv3 kernel trace: python tools/dispatches.py <dir> <kernel prefix>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith(sys.argv[2])]
print([round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows][:int(sys.argv[3]) if len(sys.argv) > 3 else 16])
