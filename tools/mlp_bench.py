"""Per-layer timing of the fused SharedMLP kernels at the SA1 / SA2 shapes of the bench
(sig3d_mlp_layer_fwd, sig3d_mlp_layer_dw, bn_relu_bwd, bn_relu_maxpool) with achieved HBM rates.
(Round 6 removed SIG3D_MLP_NT: the launcher picks the channel-tile width.)

python tools/mlp_bench.py
"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from situation3d_amd import _lib as L

dev = "cuda:0"
def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3  # us

for name, b, chans, p, ns in [("SA1", 8, [6, 64, 64, 128], 2048, 64), ("SA2", 8, [131, 128, 128, 256], 1024, 32),
                             ("SA3", 8, [259, 128, 128, 256], 512, 16), ("SA4", 8, [259, 128, 128, 256], 256, 16)]:
    e = p * ns
    x = torch.randn(b, chans[0], e, device=dev)
    for li in range(3):
        cin, cout = chans[li], chans[li + 1]
        w = torch.randn(cout, cin, device=dev) * 0.1
        y = torch.empty(b, cout, e, device=dev)
        st = torch.empty(2, cout, dtype=torch.float64, device=dev)
        ps = torch.rand(cin, device=dev) + 0.5 if li else None
        pb = torch.randn(cin, device=dev) * 0.1 if li else None
        t = timeit(lambda: L.call("sig3d_mlp_layer_fwd", b, cin, cout, e, L.ptr(x), L.ptr(w), L.ptr(ps), L.ptr(pb),
                                  L.ptr(y), L.ptr(st[0]), L.ptr(st[1]), 0, L.stream_ptr()))
        byt = 4.0 * b * e * (cin + cout)
        print("%s L%d fwd  %3d->%3d : %7.1f us  %5.2f TB/s  (%.0f MB, %.1f GFLOP -> %.0f us at 157 TF)" % (
            name, li + 1, cin, cout, t, byt / t / 1e6, byt / 1e6, 2e-9 * b * e * cin * cout, 2e-3 * b * e * cin * cout / 157e3 * 1e3))
        dw = torch.empty(cout, cin, device=dev)
        t = timeit(lambda: L.call("sig3d_mlp_layer_dw", b, cin, cout, e, L.ptr(y), L.ptr(x), L.ptr(ps), L.ptr(pb),
                                  L.ptr(dw), 0, L.stream_ptr()))
        print("%s L%d dW               : %7.1f us  %5.2f TB/s" % (name, li + 1, t, byt / t / 1e6))
        x = y
