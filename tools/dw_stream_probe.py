"""Would the SharedMLP weight gradients run faster as a k-streaming split GEMM?  dW[b] = dY[b] X[b]^T on sig3d_gemm16
(f32 core, slabs) and torch.bmm, at the shapes of the dense levels (the batch sum and the BatchNorm prologue left out)."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from situation3d_amd import _lib as L

dev = torch.device("cuda", 0)


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


SHAPES = [("SA4 L3 256x128", 256, 128, 4096, 38.8), ("SA4 L2 128x128", 128, 128, 4096, 36.6), ("SA4 L1 128x259", 128, 259, 4096, 55.8),
          ("SA3 L3 256x128", 256, 128, 8192, 57.6), ("SA3 L2 128x128", 128, 128, 8192, 38.0), ("SA3 L1 128x131", 128, 131, 8192, 65.9),
          ("SA1 L3 128x64 (14k live)", 128, 64, 14400, 35.1), ("SA1 L2 64x64 (14k live)", 64, 64, 14400, 27.6)]
b = 8
for name, m, n, k, now in SHAPES:
    dy = torch.randn(b, m, k, device=dev)
    x = torch.randn(b, n, k, device=dev)
    c = torch.empty(b, m, n, device=dev)
    row = ["%-26s now %5.1f" % (name, now), "bmm %5.1f" % timeit(lambda: torch.bmm(dy, x.transpose(1, 2), out=c))]
    for config in (1, 2):
        best = None
        for s in (1, 2, 4, 8, 16):
            if k // 32 // s < 4:
                continue
            slabs = torch.empty(max(s - 1, 1), b, m, n, device=dev)
            t = timeit(lambda: L.gemm16(dev, A=dy, lda=k, stride_a=m * k, B=x, ldb=k, stride_b=n * k, C=c, ldc=n, stride_c=m * n,
                                        C_slabs=slabs if s > 1 else None, slab_stride=b * m * n, bmode=0, batch=b, m=m, n=n, k=k,
                                        splits=s, config=config))
            if best is None or t < best[0]:
                best = (t, s)
        row.append("cfg%d %5.1f [%d]" % (config, best[0], best[1]))
    print("  ".join(row))

print("sig3d_mlp_layer_dw_stream (prologue, slabs folded) against sig3d_mlp_layer_dw, stand-alone:")
for name, m, n, k, now in SHAPES[:6]:
    dy = torch.randn(b, m, k, device=dev)
    x = torch.randn(b, n, k, device=dev)
    ps, pb = torch.rand(n, device=dev) + 0.5, torch.randn(n, device=dev)
    dw = torch.empty(m, n, device=dev)
    work = torch.empty(max(int(L.load().sig3d_mlp_layer_dw_stream_work_floats(b, n, m, k)), 4), device=dev)
    st = L.stream_ptr(dev)
    t_new = timeit(lambda: L.call("sig3d_mlp_layer_dw_stream", b, n, m, k, L.ptr(dy), L.ptr(x), L.ptr(ps), L.ptr(pb), L.ptr(None),
                                  L.ptr(dw), L.ptr(work), st))
    t_old = timeit(lambda: L.call("sig3d_mlp_layer_dw", b, n, m, k, L.ptr(dy), L.ptr(x), L.ptr(ps), L.ptr(pb), L.ptr(dw), 0, st))
    print("  %-18s stream %5.1f us   row-per-lane kernel %5.1f us" % (name, t_new, t_old))
