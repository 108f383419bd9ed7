"""Furthest point sampling time per round over scene sizes (B=8 x 40000 -> 2048 is the SA1 shape): the
single-workgroup register-resident kernel (n <= 8192) and the cooperative 8 x 512 kernel above it."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from situation3d_amd.pointnet2 import _ext
import bench
dev = torch.device("cuda", 0)
def timeit(fn, iters=5, warm=2):
    for _ in range(warm): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for b, n, m in [(8, 40000, 2048), (4, 40000, 2048), (8, 60000, 2048), (8, 90000, 2048), (8, 10000, 2048), (8, 16384, 2048), (8, 24000, 2048), (8, 8192, 2048), (8, 2048, 1024)]:
    xyz = bench.synthetic_batch(b, n, 5, dev)["point_clouds"][..., :3].contiguous()
    forms = [("blocks-16", True, 16), ("blocks-8", True, 8), ("blocks-4", True, 4), ("coop", False, 16)] if n > 8192 else [("one-wg", True, 16)]
    for name, blocks, waves in forms:
        _ext.FPS_BLOCKS, _ext.FPS_WAVES = blocks, waves    # one workgroup per scene over a Morton-ordered copy in L2 / the cooperative kernel
        ref = _ext.furthest_point_sampling(xyz, m)
        t = timeit(lambda: _ext.furthest_point_sampling(xyz, m))
        print("B=%d N=%5d M=%4d %-9s: %8.1f us  (%.2f us/round)  checksum %d" % (b, n, m, name, t, t / (m - 1), int(ref.long().sum())))
