"""sig3d_gemm16 per configuration on the step's dense-layer shapes: the f32 core (gemm16_core.h, config 0 = its own
choice), the bf16 x 6 core (gemmx6_core.h, configs 11 / 12) and torch (rocBLAS / hipBLASLt as tuned).

python tools/gemmx6_bench.py
"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from situation3d_amd import _lib as L, gemm_tuning

dev = torch.device("cuda", 0)
gemm_tuning.enable(tune_missing=True)
SHAPES = [("qkv fwd", 0, 1, 416, 2304, 768, 0), ("out-proj fwd", 0, 1, 416, 768, 768, 0),
          ("ffn up fwd (+gelu)", 0, 2, 256, 3072, 768, 1), ("ffn down fwd", 0, 2, 256, 768, 3072, 0),
          ("qkv dgrad", 1, 1, 416, 768, 2304, 0), ("out-proj dgrad", 1, 1, 416, 768, 768, 0),
          ("ffn down dgrad (*gelu')", 1, 2, 256, 3072, 768, 2), ("ffn up dgrad", 1, 2, 256, 768, 3072, 0),
          ("cross k/v fwd", 0, 1, 2048, 9216, 256, 0), ("cross k/v dgrad", 1, 1, 2048, 256, 9216, 0)]


def timeit(fn, n=100):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


print("%-26s %9s %9s %9s %9s   (us per product; splits in brackets)" % ("shape", "torch", "f32 core", "x6 8w", "x6 4w"))
for name, bmode, batch, m, n, k, act in SHAPES:
    a = torch.randn(batch, m, k, device=dev)
    w = torch.randn(batch, n, k, device=dev) / k ** 0.5
    wmat = w if bmode == 0 else w.transpose(1, 2).contiguous()
    bias = torch.randn(batch, n, device=dev)
    aux = torch.randn(batch, m, n, device=dev)
    c = torch.empty(batch, m, n, device=dev)
    slabs = torch.empty(8, batch, m, n, device=dev)
    wt = w.transpose(1, 2)

    def torch_fn():
        if act == 1:
            torch.nn.functional.gelu(torch.baddbmm(bias[:, None, :], a, wt))
        else:
            torch.bmm(a, wt, out=c)
    row = [timeit(torch_fn)]
    note = []
    for config in (0, 11, 12):
        s = L.gemm16_splits(bmode, batch, m, n, k, act, config)

        def fn():
            L.gemm16(dev, A=a, lda=k, stride_a=m * k, B=wmat, ldb=(k if bmode == 0 else n), stride_b=n * k, C=c, ldc=n,
                     stride_c=m * n, C_slabs=slabs if s > 1 else None, slab_stride=batch * m * n,
                     bias=bias if act == 1 else None, stride_bias=n, aux=aux if act else None, bmode=bmode, batch=batch,
                     m=m, n=n, k=k, act=act, splits=s, config=config)
        row.append(timeit(fn))
        note.append(s)
    print("%-26s %9.1f %6.1f[%d] %6.1f[%d] %6.1f[%d]   %.2f GFLOP" % (name, row[0], row[1], note[0], row[2], note[1], row[3],
                                                                     note[2], 2e-9 * batch * m * n * k))
