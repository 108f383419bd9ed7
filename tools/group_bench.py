"""query_group_fused at the four SA shapes of the bench (per-level time and achieved HBM rate)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from situation3d_amd import _lib as L
import bench
dev = torch.device("cuda", 0)
def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
b = 8
from situation3d_amd.pointnet2 import _ext
pc = bench.synthetic_batch(b, bench.N_POINTS, 7, dev)["point_clouds"]
cur_xyz = pc[..., :3].contiguous()
RADII = [0.2, 0.4, 0.8, 1.2]
tot_t = tot_b = 0
print("real geometry (FPS centres, ball-query lists) of the bench scenes; random feature values")
for (n, m, ns, c), radius in zip(bench.SA_LEVELS, RADII):
    xyz = cur_xyz
    inds = _ext.furthest_point_sampling(xyz, m)
    new_xyz = torch.gather(xyz, 1, inds.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    idx = _ext.ball_query(new_xyz, xyz, radius, ns)
    cur_xyz = new_xyz
    feat = torch.randn(b, c, n, device=dev)
    out = torch.empty(b, 3 + c, m, ns, device=dev)
    t = timeit(lambda: L.call("sig3d_query_group_fused", b, n, m, c, ns, 1, 1, ctypes.c_float(radius), L.ptr(xyz),
                              L.ptr(new_xyz), L.ptr(feat), L.ptr(idx), L.ptr(out), L.stream_ptr()))
    byt = bench.group_algorithmic_bytes(b, n, m, ns, c)
    extra = ""
    if c % 4 == 0:
        pm = torch.empty(b, n, c, device=dev)
        tt = timeit(lambda: L.call("sig3d_transpose_cn", b, c, n, L.ptr(feat), L.ptr(pm), L.stream_ptr()))
        out2 = torch.empty_like(out)
        tp = timeit(lambda: L.call("sig3d_query_group_fused_pm", b, n, m, c, c, ns, 1, 1, ctypes.c_float(radius), L.ptr(xyz),
                                   L.ptr(new_xyz), L.ptr(pm), L.ptr(idx), L.ptr(out2), L.stream_ptr()))
        assert torch.equal(out, out2)
        extra = "   channel-major kernel %5.1f us; + transpose %4.1f us" % (t, tt)
        t = tp
    tf = timeit(lambda: out.fill_(1.0))
    tot_t += t; tot_b += byt
    print("N=%5d M=%4d ns=%2d C=%3d: %6.1f us  %5.2f TB/s (%.1f MB; fill_ of the output alone: %4.1f us)%s"
          % (n, m, ns, c, t, byt / t / 1e6, byt / 1e6, tf, extra))
print("total %.1f us, %.2f TB/s = %.3f of 8 TB/s" % (tot_t, tot_b / tot_t / 1e6, tot_b / tot_t / 8e6))

# backward: channel-major LDS scatter-add vs the point-major kernel, with ball-query-like padded lists
print("backward (scatter-add), lists padded with their first index like ball query:")
for (n, m, ns, c) in bench.SA_LEVELS[1:]:
    g = torch.Generator().manual_seed(n)
    idx = torch.randint(0, n, (b, m, ns), generator=g, dtype=torch.int32)
    keep = torch.randint(1, max(2, ns // 4), (b, m, 1), generator=g)
    idx = torch.where(torch.arange(ns).view(1, 1, ns) < keep, idx, idx[:, :, :1]).to(dev)
    go = torch.randn(b, 3 + c, m, ns, device=dev)
    ref = torch.empty(b, c, n, device=dev); pm = torch.empty(b, n, c, device=dev); back = torch.empty(b, c, n, device=dev)
    t0 = timeit(lambda: L.call("sig3d_query_group_fused_grad", b, n, m, c, ns, 3 + c, 3, L.ptr(go), L.ptr(idx), L.ptr(ref), L.stream_ptr()))
    t1 = timeit(lambda: L.call("sig3d_query_group_fused_grad_pm", b, n, m, c, c, ns, 3 + c, 3, L.ptr(go), L.ptr(idx), L.ptr(pm), L.stream_ptr()))
    t2 = timeit(lambda: L.call("sig3d_transpose_cn", b, n, c, L.ptr(pm), L.ptr(back), L.stream_ptr()))
    print("  N=%5d M=%4d ns=%2d C=%3d: channel-major %6.1f us, point-major %6.1f us + transpose %4.1f us  (max diff %.2e)"
          % (n, m, ns, c, t0, t1, t2, (ref - back).abs().max().item()))

# practical ceilings at the same output sizes: a pure streaming write (fill) and a copy
print("streaming ceilings at the grouped-tensor sizes (torch fill_ / copy_):")
for (n, m, ns, c) in bench.SA_LEVELS:
    out = torch.empty(b, 3 + c, m, ns, device=dev)
    src = torch.empty_like(out)
    tf = timeit(lambda: out.fill_(1.0))
    tc = timeit(lambda: out.copy_(src))
    byt = out.numel() * 4
    print("  %6.1f MB: fill %5.1f us = %.2f TB/s written; copy %5.1f us = %.2f TB/s moved (R+W)"
          % (byt / 1e6, tf, byt / tf / 1e6, tc, 2 * byt / tc / 1e6))
