"""How full are the ball-query lists of the four SA levels on the bench's synthetic scenes?  ball_query pads a
short list by repeating its first hit, and every repeated neighbour is an identical column of the grouped
tensor, of every SharedMLP layer above it and of the max-pool."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from situation3d_amd.pointnet2 import _ext

dev = torch.device("cuda", 0)
pc = bench.synthetic_batch(bench.BATCH, bench.N_POINTS, 7, dev)["point_clouds"]
xyz = pc[..., :3].contiguous()
for (n, m, ns, c), radius in zip(bench.SA_LEVELS, [0.2, 0.4, 0.8, 1.2]):
    inds = _ext.furthest_point_sampling(xyz, m)
    new_xyz = torch.gather(xyz, 1, inds.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    idx = _ext.ball_query(new_xyz, xyz, radius, ns)
    srt = idx.sort(-1).values
    uniq = 1 + (srt[..., 1:] != srt[..., :-1]).sum(-1)
    print("N=%5d M=%4d r=%.1f ns=%2d: %.1f distinct neighbours per list on average (%.0f %% of the list), %d %% of the "
          "lists full" % (n, m, radius, ns, uniq.float().mean().item(), 100 * uniq.float().mean().item() / ns,
                          100 * (uniq == ns).float().mean().item()))
    xyz = new_xyz
