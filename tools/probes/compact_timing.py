"""Duration of sig3d_compact_neighbour_lists at the SA1 / SA2 shapes (real ball-query lists of a synthetic batch)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
from situation3d_amd.geometry import GeometryPlan
from situation3d_amd.model import PointNet2Encoder
dev = torch.device("cuda", 0)
batch = bench.synthetic_batch(bench.BATCH, bench.N_POINTS, 7, dev)
plan = GeometryPlan(bench.BATCH, bench.N_POINTS, PointNet2Encoder.LEVELS, dev).compute(batch["point_clouds"][..., :3].contiguous())
for lvl in (0, 1):
    cl, idx = plan.compact[lvl], plan.ball_idx[lvl]
    if cl is None:
        continue
    for _ in range(3):
        cl.compute(idx)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        cl.compute(idx)
    e.record(); torch.cuda.synchronize()
    print("SA%d lists %s: %.1f us per call, distinct fraction %.3f" % (lvl + 1, tuple(idx.shape), s.elapsed_time(e) * 50,
          cl.n_act.float().mean().item() / (idx.shape[1] * idx.shape[2])))
