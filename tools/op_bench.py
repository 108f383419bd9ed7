"""Per-op timings of the HIP path at BASELINE shapes (developer tool, not the judged bench).

python tools/op_bench.py [--b 8] [--n 40000]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import pointnet2._ext as ext  # noqa: E402
from util import feats, scene  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--b", type=int, default=8)
    ap.add_argument("--n", type=int, default=40000)
    a = ap.parse_args()
    dev = "cuda:0"
    B = a.b
    levels = [(a.n, 2048, 0.2, 64, 3), (2048, 1024, 0.4, 32, 128), (1024, 512, 0.8, 16, 256),
              (512, 256, 1.2, 16, 256)]
    xyz = scene(B, a.n, seed=0).to(dev)
    for li, (n, m, r, ns, c) in enumerate(levels):
        f = feats(B, c, n).to(dev)
        t_fps = timeit(lambda: ext.furthest_point_sampling(xyz, m), iters=5, warm=1)
        inds = ext.furthest_point_sampling(xyz, m)
        new_xyz = ext.gather_points(xyz.transpose(1, 2).contiguous(), inds).transpose(1, 2).contiguous()
        t_bq = timeit(lambda: ext.ball_query(new_xyz, xyz, r, ns))
        idx = ext.ball_query(new_xyz, xyz, r, ns)
        t_gp = timeit(lambda: ext.group_points(f, idx))
        go = torch.rand(B, c, m, ns, device=dev)
        t_gg = timeit(lambda: ext.group_points_grad(go, idx, n))
        gb = B * (4 * c * n + 4 * m * ns + 4 * c * m * ns)
        bq_b = B * (12 * n + 12 * m + 4 * m * ns)
        print("SA%d n=%d m=%d ns=%d c=%d | fps %.0f us | ball_query %.1f us (%.2f TB/s alg) | "
              "group %.1f us (%.2f TB/s) | group_grad %.1f us (%.2f TB/s)" %
              (li + 1, n, m, ns, c, t_fps, t_bq, bq_b / t_bq / 1e6, t_gp, gb / t_gp / 1e6, t_gg,
               gb / t_gg / 1e6), flush=True)
        xyz = new_xyz


if __name__ == "__main__":
    main()
