"""Means per leg of the files tools/ab.sh wrote: python tools/ab_table.py gpurun_out/ab_<tag>.txt ..."""
import re, sys
for f in sys.argv[1:]:
    legs = {}
    for line in open(f):
        m = re.match(r"(\S+)\s+(.*?)\s+(\d+) Mrays/s\s+([\d.]+) ms/frame \(uninstrumented\s+(\d+)\) \| alone\s+(\d+) Mrays/s\s+([\d.]+) ms\s+(\{.*\})", line)
        if not m:
            continue
        legs.setdefault(m.group(1), []).append((float(m.group(3)), float(m.group(5)), float(m.group(7)), eval(m.group(8))))
    for k, v in legs.items():
        n = len(v)
        kern = {kk: round(sum(x[3][kk] for x in v) / n, 3) for kk in v[0][3]}
        print("%-14s %-40s in flight %7.0f (plain %7.0f) Mrays/s | alone %8.3f ms %s" % (f.split("ab_")[-1][:-4], k[-40:], sum(x[0] for x in v) / n, sum(x[1] for x in v) / n, sum(x[2] for x in v) / n, kern))
