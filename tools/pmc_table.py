"""Per-kernel averages of every counter found under a tools/pmc_kernel.sh output directory.
python tools/pmc_table.py gpurun_out/pmc_<name> [kernel-name substring]"""
import csv, glob, os, re, sys
d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
acc = {}
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if pat not in r["Kernel_Name"]:
            continue
        name = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0]
        a = acc.setdefault((name, r["Grid_Size"]), {})
        c = a.setdefault(r["Counter_Name"], [0.0, 0])
        c[0] += float(r["Counter_Value"]); c[1] += 1
for (name, grid), a in sorted(acc.items()):
    print("%s  grid %s" % (name[:100], grid))
    v = {k: s / max(n, 1) for k, (s, n) in a.items()}
    for k in sorted(v):
        print("    %-28s %14.1f" % (k, v[k]))
    if "GRBM_GUI_ACTIVE" in v and "SQ_VALU_MFMA_BUSY_CYCLES" in v:
        print("    MFMA pipe busy               %13.1f %%" % (100 * v["SQ_VALU_MFMA_BUSY_CYCLES"] / (v["GRBM_GUI_ACTIVE"] / 8 * 1024)))
    if "SQ_WAVE_CYCLES" in v:
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
            if k in v:
                print("    %-20s / WAVE_CYCLES %8.1f %%" % (k, 100 * v[k] / v["SQ_WAVE_CYCLES"]))
