import sys, torch, torch.nn as nn
sys.path.insert(0, '.')
from situation3d_amd import small_mlp
dev = "cuda:0"
seq = nn.Sequential(nn.Linear(3, 128), nn.GELU(), nn.Linear(128, 256)).to(dev)
x = torch.randn(8, 256, 3, device=dev); res = torch.randn(8, 256, 256, device=dev, requires_grad=True)
G = torch.randn(8, 256, 256, device=dev)
def t(fn, it=20):
    for _ in range(5): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        fn(); torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=st):
            for _ in range(it): fn()
        g.replay(); torch.cuda.synchronize()
        s.record(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3
for en in (True, False):
    small_mlp.ENABLED = en
    def fwd():
        return small_mlp.pos_embed_add(seq, x, res)
    def fb():
        for p in seq.parameters(): p.grad = None
        res.grad = None
        (small_mlp.pos_embed_add(seq, x, res) * G).sum().backward()
    with torch.no_grad():
        tf = t(fwd)
    tfb = t(fb)
    print("fused" if en else "torch", "forward %.1f us, forward+backward (incl. the product with G) %.1f us" % (tf, tfb))
