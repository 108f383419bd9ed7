"""HBM traffic per launch of the grouping kernels from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE;
counter values are KiB).  FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950.

    python tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json> "<command>"
"""
import csv
import glob
import json
import os
import re
import sys

fetch_dir, write_dir, dst, cmd = sys.argv[1:5]
PAT = "query_group_fused"


def collect(d, counter):
    vals = {}
    for f in glob.glob(os.path.join(d, "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            # dense launches only: the compact-mode instance <true> moves a data-dependent fraction
            if r["Counter_Name"] == counter and PAT in name and "grad" not in name and "pm_kernel<true>" not in name:
                vals.setdefault(re.search(r"query_group_fused\w*", name).group(0), []).append(float(r["Counter_Value"]) * 1024.0)
    return vals


fe, wr = collect(fetch_dir, "FETCH_SIZE"), collect(write_dir, "WRITE_SIZE")
n = sum(len(v) for v in fe.values())
assert n and n == sum(len(v) for v in wr.values()), (n, {k: len(v) for k, v in wr.items()})
fetch = 2.0 * sum(sum(v) for v in fe.values()) / n
write = sum(sum(v) for v in wr.values()) / n
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
# levels that form the dense grouped tensor (the others run in compact mode)
from situation3d_amd.pointnet2 import fused_mlp  # noqa: E402
dense = [lvl for lvl in bench.SA_LEVELS
         if not (fused_mlp.COMPACT and bench.BATCH * lvl[1] * lvl[2] >= fused_mlp.COMPACT_MIN_POSITIONS)]
alg = sum(bench.group_algorithmic_bytes(bench.BATCH, *lvl) for lvl in dense) / max(len(dense), 1)
out = {"kernel": " + ".join(sorted(fe)),
       "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes), %s; counters in "
                 "KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B)" % cmd,
       "launches_counted": n,
       "per_kernel_launches": {k: len(v) for k, v in fe.items()},
       "fetch_bytes_per_launch_corrected_x2": fetch, "write_bytes_per_launch": write,
       "traffic_bytes_per_launch": fetch + write, "algorithmic_bytes_per_launch": round(alg)}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))
