"""Developer check of the fused SharedMLP forward kernels against torch (values + timing)."""
import ctypes
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from situation3d_amd import _lib as L  # noqa: E402

dev = "cuda:0"


def run(b, chans, P, S, iters=10):
    torch.manual_seed(0)
    E = P * S
    x = torch.randn(b, chans[0], P, S, device=dev)
    Ws = [torch.randn(chans[i + 1], chans[i], device=dev) * (2.0 / chans[i]) ** 0.5 for i in range(len(chans) - 1)]
    gam = [torch.rand(c, device=dev) + 0.5 for c in chans[1:]]
    bet = [torch.randn(c, device=dev) * 0.1 for c in chans[1:]]

    def torch_ref():
        h = x
        for W, g, bt in zip(Ws, gam, bet):
            h = torch.bmm(W[None].expand(b, -1, -1), h.reshape(b, h.shape[1], E)).view(b, -1, P, S)
            h = F.relu(F.batch_norm(h, None, None, g, bt, True, 0.1, 1e-5))
        return h.max(3)

    def fused():
        s = L.stream_ptr()
        cur, ps, pb = x, None, None
        for W, g, bt in zip(Ws, gam, bet):
            cout, cin = W.shape
            y = torch.empty(b, cout, P, S, device=dev)
            st = torch.empty(2, cout, dtype=torch.float64, device=dev)
            L.call("sig3d_mlp_layer_fwd", b, cin, cout, E, L.ptr(cur), L.ptr(W), L.ptr(ps), L.ptr(pb), L.ptr(y),
                   L.ptr(st[0]), L.ptr(st[1]), s)
            aff = torch.empty(4, cout, device=dev)
            L.call("sig3d_bn_finalize", cout, ctypes.c_double(b * E), ctypes.c_float(1e-5), ctypes.c_float(0.1),
                   L.ptr(st[0]), L.ptr(st[1]), L.ptr(g), L.ptr(bt), L.ptr(aff[0]), L.ptr(aff[1]), L.ptr(aff[2]),
                   L.ptr(aff[3]), L.ptr(None), L.ptr(None), L.ptr(None), s)
            cur, ps, pb = y, aff[0], aff[1]
        out = torch.empty(b, chans[-1], P, device=dev)
        arg = torch.empty(b, chans[-1], P, dtype=torch.int32, device=dev)
        L.call("sig3d_bn_relu_maxpool", b, chans[-1], P, S, L.ptr(cur), L.ptr(ps), L.ptr(pb), L.ptr(out), L.ptr(arg), s)
        return out, arg

    rv, ri = torch_ref()
    fv, fi = fused()
    torch.cuda.synchronize()
    err = (rv - fv).abs().max().item()
    same = (ri == fi.long()).float().mean().item()

    def t(fn):
        fn(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) / iters
    print("b=%d chans=%s P=%d S=%d | max|diff| %.2e argmax agree %.4f | torch %.3f ms fused %.3f ms" %
          (b, chans, P, S, err, same, t(torch_ref), t(fused)), flush=True)


if __name__ == "__main__":
    run(2, [6, 16, 32], 37, 12)
    run(8, [6, 64, 64, 128], 2048, 64)
    run(8, [131, 128, 128, 256], 1024, 32)
    run(8, [259, 128, 128, 256], 512, 16)
    run(8, [259, 128, 128, 256], 256, 16)
