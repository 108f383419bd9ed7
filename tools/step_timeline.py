"""Timeline of ONE replayed step from a rocprofv3 --kernel-trace CSV: which kernels of the main branch run
while the geometry branch (FPS / gather_xyz / ball query / compaction of batch i+1) is active, and the idle
gaps between consecutive main-branch kernels.

python tools/step_timeline.py <kernel_trace.csv> [step_from_end=1] [--list]
"""
import csv
import sys

SIDE = ("fps_", "gather_xyz", "ball_query", "grid_", "compact", "bq_", "hash_")

path = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 1
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"], r["Stream_Id"])
        for r in csv.DictReader(open(path))]
rows.sort()
marks = [i for i, r in enumerate(rows) if "fps_coop" in r[2]]
# one cooperative FPS launch per step (B = 8 scenes = one launch)
lo = marks[-back - 1]
hi = marks[-back]
step = rows[lo:hi]
t0 = step[0][0]
is_side = lambda n: any(n.startswith(p) or ("::" + p) in n or (" " + p) in n for p in SIDE)
side = [r for r in step if is_side(r[2])]
main = [r for r in step if not is_side(r[2])]
print("step span %.3f ms, %d launches (%d side)" % ((step[-1][1] - t0) / 1e6, len(step), len(side)))
print("side branch: busy %.3f ms, ends at +%.3f ms" % (sum(e - s for s, e, *_ in side) / 1e6, (max(e for s, e, *_ in side) - t0) / 1e6))
for s, e, n, q, st in side:
    print("   side +%.3f  %.1f us  %s" % ((s - t0) / 1e6, (e - s) / 1e3, n[:60]))
busy = sum(e - s for s, e, *_ in main)
gaps = 0
prev = None
for s, e, n, q, st in main:
    if prev is not None and s > prev:
        gaps += s - prev
    prev = max(prev or 0, e)
print("main branch: %d launches, busy %.3f ms, gaps %.3f ms" % (len(main), busy / 1e6, gaps / 1e6))
if "--list" in sys.argv:
    prev = None
    for s, e, n, q, st in main:
        print("+%.3f gap %.1f dur %.1f q%s %s" % ((s - t0) / 1e6, (s - prev) / 1e3 if prev else 0, (e - s) / 1e3, q, n[:70]))
        prev = e
