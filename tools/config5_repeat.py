"""bench.py's config-5 variant a few times in one process (tuned library as in a full bench run): the spread of its lines."""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from situation3d_amd import gemm_tuning
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
gemm_tuning.enable(tune_missing=True)
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    v = bench.config5_variant(dev)
    for k, d in v.items():
        print(i, k, "own %.2f ms  library %.2f ms  projections %.2f ms (%d launches, %.0f TF)" % (
            d["ms"], d["library_projections_ms"], d["roofline_kv_projection"]["ms"], d["roofline_kv_projection"]["launches"],
            d["roofline_kv_projection"]["achieved"]), flush=True)
