"""Turn a rocprofv3 --kernel-trace --stats CSV directory into a committed summary under profiles/.

python tools/summarize_rocprof.py gpurun_out/prof_x profiles/r01_name.md "command line" [steps]
"""
import csv
import glob
import os
import sys

src, dst, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
steps = int(sys.argv[4]) if len(sys.argv) > 4 else None
stats = glob.glob(os.path.join(src, "*kernel_stats.csv"))[0]
rows = list(csv.DictReader(open(stats)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
calls = sum(int(r["Calls"]) for r in rows)
with open(dst, "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats summary\n\n")
    f.write("command: `%s`\n\n" % cmd)
    f.write("total kernel time %.3f ms over %d launches" % (tot / 1e6, calls))
    if steps:
        f.write(" (%d steps incl. warm-up => %.3f ms and %d launches per step)" % (steps, tot / 1e6 / steps, calls // steps))
    f.write("\n\n| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|\n")
    for r in rows[:45]:
        f.write("| `%s` | %s | %.3f | %.1f | %.1f | %.1f | %.1f |\n" % (
            r["Name"][:110].replace("|", "/"), r["Calls"], float(r["TotalDurationNs"]) / 1e6,
            float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3,
            float(r["Percentage"])))
print("wrote", dst)
