"""Compact per-kernel view of the order of global loads (L), stores (S), LDS ops (d) and vmcnt waits
(W<n>) in the gfx950 ISA of one csrc/*.hip file -- to spot loads serialised by exec-masked blocks
(`if (c < cols) x = p[c]` inside an unrolled loop compiles to load; s_waitcnt per iteration).

python tools/isa_memory_pattern.py rowops [kernel-substring]
"""
import os, re, subprocess, sys, tempfile
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
from situation3d_amd.build import FLAGS, CSRC
name = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else ""
src = os.path.join(CSRC, name if name.endswith(".hip") else name + ".hip")
with tempfile.TemporaryDirectory() as td:
    subprocess.run(["/opt/rocm/bin/hipcc", "-x", "hip", "-c", src, "-o", os.path.join(td, "o.o"), "-save-temps=obj"] + FLAGS,
                   cwd=td, capture_output=True, text=True)
    asm = [f for f in os.listdir(td) if f.endswith("gfx950.s")][0]
    lines = open(os.path.join(td, asm)).read().splitlines()
cur, out = None, []
for ln in lines:
    m = re.match(r"^(_Z\w+):", ln)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        out = []
        continue
    if cur is None:
        continue
    t = ln.strip()
    if t.startswith("global_load") or t.startswith("buffer_load"): out.append("L")
    elif t.startswith("global_store") or t.startswith("buffer_store"): out.append("S")
    elif t.startswith("global_atomic"): out.append("A")
    elif t.startswith("ds_"): out.append("d")
    elif t.startswith("v_mfma"): out.append("M")
    elif t.startswith("s_waitcnt") and "vmcnt" in t: out.append("W" + re.search(r"vmcnt\((\d+)\)", t).group(1))
    elif t.startswith("s_cbranch"): out.append("|")
    elif t.startswith(".Lfunc_end"):
        if filt in cur:
            s = "".join(out)
            s = re.sub(r"(L+)", lambda m: "L%d " % len(m.group(1)), s)
            s = re.sub(r"(S+)", lambda m: "S%d " % len(m.group(1)), s)
            s = re.sub(r"(d+)", lambda m: "d%d " % len(m.group(1)), s)
            s = re.sub(r"(M+)", lambda m: "M%d " % len(m.group(1)), s)
            print("==", cur[:100]); print(s); print()
        cur = None
