import ctypes, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from situation3d_amd import _lib as L
dev = torch.device("cuda", 0)
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
st_ = L.stream_ptr(dev)
for name, b, cin, cout, e, live in [("SA1 L2 compact", 8, 64, 64, 131072, 14400), ("SA1 L3 compact", 8, 64, 128, 131072, 14400),
                                    ("SA2 L2 compact", 8, 128, 128, 32768, 1300), ("SA2 L3 compact", 8, 128, 256, 32768, 1300),
                                    ("SA1 L2 surface", 8, 64, 64, 131072, 47000), ("SA1 L3 surface", 8, 64, 128, 131072, 47000),
                                    ("SA2 L2 surface", 8, 128, 128, 32768, 5200), ("SA2 L3 surface", 8, 128, 256, 32768, 5200)]:
    x = torch.randn(b, cin, e, device=dev); w = torch.randn(cout, cin, device=dev)
    y = torch.empty(b, cout, e, device=dev); st = torch.zeros(2, cout, dtype=torch.float64, device=dev)
    ps, pb = torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev)
    n_act = torch.full((b,), live, dtype=torch.int32, device=dev); mult = torch.ones(b, e, device=dev)
    def run(stats):
        s0, s1 = (L.ptr(st[0]), L.ptr(st[1])) if stats else (L.ptr(None), L.ptr(None))
        if live:
            L.call("sig3d_mlp_layer_fwd_compact", b, cin, cout, e, L.ptr(x), L.ptr(w), L.ptr(ps), L.ptr(pb), L.ptr(y), s0, s1, 1,
                   L.ptr(n_act), L.ptr(mult if stats else None), st_)
        else:
            L.call("sig3d_mlp_layer_fwd", b, cin, cout, e, L.ptr(x), L.ptr(w), L.ptr(ps), L.ptr(pb), L.ptr(y), s0, s1, 1, st_)
    print("%-16s with statistics %6.1f us   without %6.1f us" % (name, timeit(lambda: run(True)), timeit(lambda: run(False))))
