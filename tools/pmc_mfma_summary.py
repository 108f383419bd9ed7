"""MFMA-pipe busy fraction per kernel from a `rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE`
pass: busy / (GRBM_GUI_ACTIVE summed over the 8 XCDs / 8 x 1024 SIMDs).   python tools/pmc_mfma_summary.py <dir> [pattern]"""
import csv, glob, os, re, sys
d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "attention"
acc = {}
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if pat not in r["Kernel_Name"]:
            continue
        key = (re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0], r["Grid_Size"])
        a = acc.setdefault(key, {"n": 0})
        a[r["Counter_Name"]] = a.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            a["n"] += 1
print("| kernel | grid | launches | GRBM_GUI_ACTIVE / launch (8 XCDs) | MFMA busy cycles / launch | MFMA pipe busy |\n|---|---|---|---|---|---|")
for (name, grid), a in sorted(acc.items()):
    n = max(a["n"], 1)
    gui, busy = a.get("GRBM_GUI_ACTIVE", 0.0) / n, a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / n
    frac = busy / (gui / 8.0 * 1024.0) if gui else 0.0
    print("| `%s` | %s | %d | %.4g | %.4g | %.1f %% |" % (name[:72], grid, n, gui, busy, 100.0 * frac))
