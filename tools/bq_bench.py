"""Ball query of the four SA levels of BASELINE config 3: the cell-binned multi-level launch pair against the
ordered brute-force scan, per level and for the whole stack, on volume-uniform and surface-shaped scenes."""
import ctypes
import sys

import torch

sys.path.insert(0, '.')
import bench  # noqa: E402
from situation3d_amd import _lib as L  # noqa: E402
if len(sys.argv) > 1:      # a variant build of the library (tools only: the product loads its own)
    L.LIB_PATH = sys.argv[1]
from situation3d_amd.pointnet2 import _ext  # noqa: E402

dev = torch.device('cuda', 0)
b, n = 8, 40000
LEVELS = [(2048, 0.2, 64), (1024, 0.4, 32), (512, 0.8, 16), (256, 1.2, 16)]


def t(fn, it=20):
    for _ in range(3):
        fn()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    # the side stream starts behind everything queued so far: the operands were produced on the default stream, and
    # scratch that an earlier asynchronous call has already returned to the allocator (the FPS workspace: freed by
    # Python while its kernel still runs, handed out again as `idx`) must not be written here before that call is done
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(it):
                fn()
        g.replay()
        torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        g.replay()
        e.record()
        torch.cuda.synchronize()
    return a.elapsed_time(e) / it * 1e3


for surface in (False, True):
    batch = bench.synthetic_batch(b, n, 3, dev, surface=surface)
    cur = batch['point_clouds'][..., :3].contiguous()
    probs = []
    for m, r, ns in LEVELS:
        inds = _ext.furthest_point_sampling(cur, m)
        nxt = torch.gather(cur, 1, inds.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
        probs.append((nxt, cur, r, ns))
        cur = nxt
    print("surface-shaped scenes" if surface else "volume-uniform scenes (SURVEY 8d)")
    for li, (a, c, r, ns) in enumerate(probs):
        idx = torch.empty(b, a.shape[1], ns, dtype=torch.int32, device=dev)
        work = torch.empty(_ext.ball_query_workspace_bytes(b, a.shape[1]), dtype=torch.uint8, device=dev)
        tc = t(lambda: L.call("sig3d_ball_query_grid", b, c.shape[1], a.shape[1], ctypes.c_float(r), ns, L.ptr(a), L.ptr(c),
                              L.ptr(idx), L.ptr(work), work.numel(), L.stream_ptr()))
        got = idx.clone()
        tb = t(lambda: L.call("sig3d_ball_query", b, c.shape[1], a.shape[1], ctypes.c_float(r), ns, L.ptr(a), L.ptr(c),
                              L.ptr(idx), L.stream_ptr()), it=5)
        hits = (idx != idx[..., :1]).float().sum(-1).mean().item() + 1
        print("  SA%d: cells %.1f us, ordered scan %.1f us, equal=%s, distinct/centre=%.1f of %d"
              % (li + 1, tc, tb, torch.equal(got, idx), hits, ns))
    recs = [(c, a, r, ns, torch.empty(b, a.shape[1], ns, dtype=torch.int32, device=dev)) for a, c, r, ns in probs]
    arr = L.bq_levels(recs)
    work = torch.empty(L.bq_levels_workspace_bytes(b, arr), dtype=torch.uint8, device=dev)
    tl = t(lambda: L.call("sig3d_ball_query_levels", b, len(arr), arr, L.ptr(work), work.numel(), L.stream_ptr()))
    got = [r[4].clone() for r in recs]
    nb = sum(bench.ball_query_algorithmic_bytes(b, c.shape[1], a.shape[1], ns) for a, c, r, ns in probs)
    print("  SA1-4 in one launch pair: %.1f us for %.1f MB algorithmic = %.2f TB/s" % (tl, nb / 1e6, nb / tl / 1e6))
