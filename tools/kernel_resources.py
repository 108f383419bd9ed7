"""Registers, scratch, LDS and occupancy of every kernel of libsig3d_hip.so as the compiler reports them
(-Rpass-analysis=kernel-resource-usage; no GPU needed): `python tools/kernel_resources.py > profiles/rNN_kernel_resources.md`.
A kernel with scratch or spills is flagged; dynamic LDS (set at launch) is not in the static figure."""
import os, re, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
from situation3d_amd.build import CSRC, FLAGS, SOURCES, _hipcc

FIELDS = ["TotalSGPRs", "VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "SGPRs Spill", "VGPRs Spill",
          "LDS Size [bytes/block]"]


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return out.stdout.split("\n") if out.returncode == 0 else names


def one(src):
    cmd = [_hipcc(), "-x", "hip", "-c", os.path.join(CSRC, src), "-o", os.devnull, "-Rpass-analysis=kernel-resource-usage"] + FLAGS
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in err.split("\n"):
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1), "file": src}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\S+) \[-Rpass", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = m.group(2)
    return rows


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    depth, out = 0, []
    for ch in name:            # drop the argument list, keep template arguments
        if ch == "(" and depth == 0:
            break
        out.append(ch)
    return "".join(out)


def main():
    with ThreadPoolExecutor(max_workers=min(os.cpu_count() or 4, 8)) as ex:
        rows = [r for rs in ex.map(one, SOURCES) for r in rs]
    for r, d in zip(rows, demangle([r["name"] for r in rows])):
        r["short"] = short(d)
    print("# Kernel resources as compiled for gfx950 (`tools/kernel_resources.py`, hipcc %s)\n" % " ".join(FLAGS))
    flagged = [r for r in rows if int(r.get("ScratchSize [bytes/lane]", 0)) or int(r.get("VGPRs Spill", 0))]
    print("%d kernels in %d translation units; %d with scratch (VGPRs spilled to memory)%s\n" %
          (len(rows), len(SOURCES), len(flagged), (": " + ", ".join("`%s` (%s B/lane)" % (r["short"], r["ScratchSize [bytes/lane]"])
                                                                    for r in flagged)) if flagged else ""))
    print("| file | kernel | SGPR | VGPR | AGPR | scratch B/lane | VGPR spills | SGPR spills (to VGPR lanes) | static LDS B | waves/SIMD |\n|---|---|---|---|---|---|---|---|---|---|")
    for r in sorted(rows, key=lambda r: (r["file"], r["short"])):
        print("| %s | `%s` | %s | %s | %s | %s | %s | %s | %s | %s |" % (r["file"], r["short"], r.get("TotalSGPRs"), r.get("VGPRs"), r.get("AGPRs"),
              r.get("ScratchSize [bytes/lane]"), r.get("VGPRs Spill"), r.get("SGPRs Spill"), r.get("LDS Size [bytes/block]"),
              r.get("Occupancy [waves/SIMD]")))


if __name__ == "__main__":
    main()
