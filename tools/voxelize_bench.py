"""Scene preparation (rotations + min shift + voxelise + de-duplicate) for a B=8 batch of raw scenes:
GPU path (sig3d_voxelize, events on the launch stream) beside the numpy path of the reference's
DataLoader workers (oracle restatement, one core per scene as a worker would be).

    python tools/voxelize_bench.py [points_per_scene]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from situation3d_amd import voxelizer  # noqa: E402

dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 150000
b, voxel = 8, 0.02
rng = np.random.default_rng(0)
# surface-like scene: points on walls/floor planes so that 2 cm cells collide as in ScanNet scans
pts = (rng.random((b * n, 3)) * np.array([8.0, 8.0, 3.0])).astype(np.float32)
pts[::2, 2] = np.round(pts[::2, 2] * 2) / 2
pts = np.round(pts / 0.013) * 0.013
pts = pts.astype(np.float32)
feats = rng.random((b * n, 3)).astype(np.float32)
labels = rng.integers(0, 20, b * n).astype(np.int32)
ang = rng.random((b, 3)) * np.pi / 18 - np.pi / 36
rots = np.zeros((b, 3, 3, 3))
for s in range(b):
    cx, sx = np.cos(ang[s, 0]), np.sin(ang[s, 0])
    cy, sy = np.cos(ang[s, 1]), np.sin(ang[s, 1])
    cz, sz = np.cos(ang[s, 2]), np.sin(ang[s, 2])
    rots[s, 0] = [[1, 0, 0], [0, cx, -sx], [0, sx, cx]]
    rots[s, 1] = [[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]
    rots[s, 2] = [[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]]
off = [i * n for i in range(b + 1)]
tp, tf, tl = torch.from_numpy(pts).to(dev), torch.from_numpy(feats).to(dev), torch.from_numpy(labels).to(dev)
trot = torch.from_numpy(rots).to(dev)


def run():
    return voxelizer.voxelize_batch(tp, off, tf, tl, rotations=trot, voxel_size=voxel)


for _ in range(3):
    vb = run()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
iters = 20
s.record()
for _ in range(iters):
    vb = run()
e.record()
torch.cuda.synchronize()
ms = s.elapsed_time(e) / iters
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    vb = run()
g.replay()
torch.cuda.synchronize()
s.record()
for _ in range(iters):
    g.replay()
e.record()
torch.cuda.synchronize()
gms = s.elapsed_time(e) / iters
u = vb.num_unique.cpu().numpy()
# algorithmic bytes: read coords 12 + feats 12 + labels 4; 8 sort passes x (read 12 + write 12) + key write 12;
# outputs inverse 4 + per kept point (inds 4 + cells 12 + feats 12 + labels 4)
alg = b * n * (28 + 12 + 8 * 24 + 4) + int(u.sum()) * 32
print("B=%d x %d pts, voxel %.2f: kept %s" % (b, n, voxel, u.tolist()))
print("GPU eager %.3f ms, hipGraph %.3f ms  (%.1f M points/s, %.0f GB/s of %.1f MB algorithmic incl. 8 sort passes)"
      % (ms, gms, b * n / gms / 1e3, alg / gms / 1e6, alg / 1e6))

from oracle import voxelize_ref as ref  # noqa: E402  (CPU baseline leg only)
t0 = time.perf_counter()
cells, inds, inverse, mins = ref.prepare_scene(pts[:n], rots[0], voxel)
cpu = time.perf_counter() - t0
c, _, _, inv, ind = vb.scene(0)
assert np.array_equal(ind.cpu().numpy(), inds) and np.array_equal(inv.cpu().numpy(), inverse)
assert np.array_equal(c.cpu().numpy(), cells)
print("numpy path, one scene on one core: %.1f ms  -> a B=8 batch = %.1f ms of worker time; GPU/CPU-core = %.0fx"
      % (cpu * 1e3, cpu * 8e3, cpu * 8e3 / gms))
