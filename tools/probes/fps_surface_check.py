"""FPS on the bench's surface-shaped scenes, every path against the oracle (indices must be identical)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench
from oracle import pointnet2_ref as oracle
from situation3d_amd.pointnet2 import _ext
dev = torch.device("cuda", 0)
for surface in (True, False):
    for seed in (3, 1234):
        xyz = bench.synthetic_batch(8, 40000, seed, dev, surface=surface)["point_clouds"][..., :3].contiguous()
        ref = oracle.furthest_point_sampling(xyz.cpu(), 2048)
        for name, blocks, waves in (("blocks-16", True, 16), ("blocks-8", True, 8), ("blocks-4", True, 4), ("coop", False, 16)):
            _ext.FPS_BLOCKS, _ext.FPS_WAVES = blocks, waves
            got = _ext.furthest_point_sampling(xyz, 2048).cpu()
            bad = (got != ref).nonzero()
            print("surface=%s seed=%d %-9s: %s" % (surface, seed, name, "identical" if len(bad) == 0 else
                  "MISMATCH at %s got %s ref %s (min %d max %d)" % (bad[0].tolist(), int(got[tuple(bad[0])]), int(ref[tuple(bad[0])]), int(got.min()), int(got.max()))), flush=True)
