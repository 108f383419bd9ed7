"""HBM traffic of the roofline pair (ball_query + query_group + point-major transposes, all four SA levels) per
training step, from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; counter values are KiB).  FETCH_SIZE is
doubled as MI355X_MICROARCH.md prescribes for gfx950 (wide coalesced reads are tallied at half their bytes).

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d F -- python3 bench.py --steps 2 --warmup 1 \
        --no-graph --no-variants --no-cpu-baseline          (and the same with WRITE_SIZE into W)
    python tools/pmc_traffic.py F W profiles/r02_pmc_group_pair.json "<command>" <commit>
"""
import csv
import glob
import hashlib
import json
import os
import re
import sys

fetch_dir, write_dir, dst, cmd, commit = sys.argv[1:6]
PATS = ("query_group", "ball_query", "bqg_", "bqc_", "transpose_cn_kernel")
STEPS = 1 + 2 + 3   # warm-up + timed + event-bracketed eager steps of the command above


def collect(d, counter):
    per_kernel = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if r["Counter_Name"] == counter and any(p in name for p in PATS) and "grad" not in name:
                key = re.sub(r"\(anonymous namespace\)::|void ", "", name).split("(")[0]
                per_kernel.setdefault(key, []).append(float(r["Counter_Value"]) * 1024.0)
    return per_kernel


fe, wr = collect(fetch_dir, "FETCH_SIZE"), collect(write_dir, "WRITE_SIZE")
assert fe and set(fe) == set(wr), (sorted(fe), sorted(wr))
rows = {}
for k in sorted(fe):
    rows[k] = {"launches_per_step": len(fe[k]) / STEPS, "fetch_bytes_per_step": 2.0 * sum(fe[k]) / STEPS,
               "write_bytes_per_step": sum(wr[k]) / STEPS}
fetch = sum(r["fetch_bytes_per_step"] for r in rows.values())
write = sum(r["write_bytes_per_step"] for r in rows.values())
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
# every source is hashed; bench.pair_traffic_is_stale compares the ones that decide the pair (bench.PAIR_SOURCES)
WATCHED = ("situation3d_amd/csrc/*.hip", "situation3d_amd/csrc/*.h", "situation3d_amd/pointnet2/*.py", "situation3d_amd/geometry.py")


def source_hashes(root=ROOT):
    """sha256 of every source that decides which launches the pair consists of and what they move (bench.py compares
    them with the tree it runs in: no git on the GPU box)."""
    h = {}
    for pat in WATCHED:
        for f in sorted(glob.glob(os.path.join(root, pat))):
            h[os.path.relpath(f, root)] = hashlib.sha256(open(f, "rb").read()).hexdigest()[:16]
    return h


out = {"kernels": rows, "source_hashes": source_hashes(), "fetch_bytes_per_step": fetch, "write_bytes_per_step": write,
       "traffic_bytes_per_step": fetch + write, "commit": commit, "command": cmd,
       "note": "FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE, separate --pmc passes, eager steps (geometry inline)"}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "kernels"}, indent=1))
