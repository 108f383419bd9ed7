"""Print VGPR/SGPR/LDS/scratch/occupancy per kernel of one csrc/*.hip file (hipcc remarks)."""
import re, subprocess, sys, os
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
from situation3d_amd.build import FLAGS, CSRC
for name in sys.argv[1:]:
    src = os.path.join(CSRC, name if name.endswith(".hip") else name + ".hip")
    p = subprocess.run(["/opt/rocm/bin/hipcc", "-x", "hip", "-c", src, "-o", "/dev/null"] + FLAGS +
                       ["-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
    cur = {}
    for line in p.stderr.splitlines():
        m = re.search(r"remark: [^:]+:\d+:\d+:\s+(.*?): (.*?) \[-Rpass", line) or re.search(r":\s+([A-Za-z][A-Za-z \[\]/]*?): (\S+) \[-Rpass", line)
        if not m: continue
        k, v = m.group(1).strip(), m.group(2).strip()
        if k == "Function Name" or k == "Name":
            if cur: print(cur)
            cur = {"kernel": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()[:70]}
        elif k in ("VGPRs", "AGPRs", "TotalSGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]", "VGPRs Spill"):
            cur[k.split(" ")[0]] = v
    if cur: print(cur)
