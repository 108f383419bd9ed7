"""Kernels between the last two AdamW launches of a rocprofv3 kernel trace, per queue: what one replayed step launches
and on which queue.  python tools/step_window.py <r_kernel_trace.csv> [--list]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ad = [i for i, r in enumerate(rows) if "adamw_table_kernel" in r["Kernel_Name"]]
a, b = ad[-2], ad[-1]
t0, t1 = int(rows[a]["End_Timestamp"]), int(rows[b]["End_Timestamp"])
print("window %.3f ms, %d launches" % ((t1 - t0) / 1e6, b - a))
agg = collections.defaultdict(lambda: [0, 0.0])
busy = collections.defaultdict(float)
for r in rows[a + 1:b + 1]:
    q = r["Queue_Id"]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    k = r["Kernel_Name"]
    k = k[:90]
    agg[(q, k)][0] += 1
    agg[(q, k)][1] += d
    busy[q] += d
for q in sorted(busy):
    print("queue %s: busy %.3f ms" % (q, busy[q] / 1e3))
    items = sorted(((v[1], v[0], k[1]) for k, v in agg.items() if k[0] == q), reverse=True)
    for us, n, k in items[: (400 if "--list" in sys.argv else 25)]:
        print("   %8.1f us %4d  %s" % (us, n, k))
if "--gaps" in sys.argv:
    main = max(busy, key=lambda q: busy[q])
    prev = None
    gaps = []
    for r in rows[a + 1:b + 1]:
        if r["Queue_Id"] != main: continue
        if prev is not None:
            gaps.append(((int(r["Start_Timestamp"]) - int(prev["End_Timestamp"])) / 1e3, prev["Kernel_Name"][:50], r["Kernel_Name"][:50]))
        prev = r
    print("main queue %s: idle between launches %.3f ms; largest gaps:" % (main, sum(g[0] for g in gaps if g[0] > 0) / 1e3))
    for g in sorted(gaps, reverse=True)[:15]:
        print("   %8.1f us  %s -> %s" % g)
if "--sequence" in sys.argv:
    print("launch order (us since the previous AdamW's end, duration, name):")
    for r in rows[a + 1:b + 1]:
        print("   %9.1f %7.1f  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                                      r["Kernel_Name"][:110]))
