"""Sum rocprofv3 counter_collection rows per (kernel, counter): python tools/pmc_sum.py file.csv [...]"""
import csv, sys, collections
for path in sys.argv[1:]:
    tot = collections.defaultdict(float); cnt = collections.Counter()
    with open(path) as f:
        for r in csv.DictReader(f):
            k = r["Kernel_Name"].split("(")[0][:40]
            tot[(k, r["Counter_Name"])] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    print(path)
    for (k, c), v in sorted(tot.items()):
        print("  %-42s %-28s n=%-5d sum=%.6g  avg=%.6g" % (k, c, cnt[(k, c)], v, v / cnt[(k, c)]))
