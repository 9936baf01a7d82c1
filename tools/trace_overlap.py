"""How well concurrent frames overlap, from a rocprofv3 kernel trace: python tools/trace_overlap.py <kernel_trace.csv> [skip_fraction]
Per queue: kernels, busy time, span.  Overall: time with >= 1 / >= 2 kernels running, wall span."""
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kernel_Name"].lstrip("void ").startswith("k_")]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = rows[int(len(rows) * skip):]  # the timed part
t0 = int(rows[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in rows)
byq = collections.defaultdict(list)
for r in rows:
    byq[r.get("Queue_Id", "?")].append(r)
for q, rs in sorted(byq.items()):
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs)
    span = max(int(r["End_Timestamp"]) for r in rs) - int(rs[0]["Start_Timestamp"])
    print("queue %s: %d kernels, busy %.2f ms of span %.2f ms (%.0f %%)" % (q, len(rs), busy / 1e6, span / 1e6, 100.0 * busy / span))
ev = []
for r in rows:
    ev.append((int(r["Start_Timestamp"]), 1)); ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
n = 0; last = ev[0][0]; hist = collections.Counter()
for t, d in ev:
    hist[n] += t - last; last = t; n += d
tot = t1 - t0
print("wall %.2f ms; kernels running at once: " % (tot / 1e6) + ", ".join("%d: %.0f %%" % (k, 100.0 * v / tot) for k, v in sorted(hist.items())))
names = collections.defaultdict(lambda: [0, 0])
for r in rows:
    k = r["Kernel_Name"].replace("void ", "").split("<")[0].split("(")[0]
    names[k][0] += 1; names[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, (c, ns) in sorted(names.items(), key=lambda kv: -kv[1][1]):
    print("  %-14s %5d launches, %8.1f us on average, %.2f ms in all" % (k, c, ns / c / 1e3, ns / 1e6))
