"""Do two independent small-M GEMM chains (the Q-Former's dX and dW products, M = 416 rows) overlap when they
run on two streams?  Upper bound for moving the weight-gradient GEMMs of backward onto a side branch."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from situation3d_amd import gemm_tuning
dev = torch.device("cuda", 0)
gemm_tuning.enable(tune_missing=True)
M = 416
dy1, w1 = torch.randn(M, 3072, device=dev), torch.randn(3072, 768, device=dev)
dy2, w2 = torch.randn(M, 768, device=dev), torch.randn(768, 3072, device=dev)
x1, x2 = torch.randn(M, 768, device=dev), torch.randn(M, 3072, device=dev)
def dx_chain():
    for _ in range(12):
        a = dy1.mm(w1)      # (416,3072)x(3072,768)
        b = dy2.mm(w2)
def dw_chain():
    for _ in range(12):
        a = dy1.t().mm(x1)  # (3072,416)x(416,768)
        b = dy2.t().mm(x2)
main, side = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(main)
    for _ in range(n): fn()
    e.record(main); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
def seq():
    with torch.cuda.stream(main):
        dx_chain(); dw_chain()
def par():
    with torch.cuda.stream(main):
        side.wait_stream(main)
        with torch.cuda.stream(side):
            dw_chain()
        dx_chain()
        main.wait_stream(side)
def graphed(fn):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(main):
        fn(); torch.cuda.synchronize()
        with gemm_tuning.no_tuning(), torch.cuda.graph(g, stream=main):
            fn()
    return lambda: g.replay()
with torch.cuda.stream(main):
    t_dx = timed(dx_chain); t_dw = timed(dw_chain)
print("eager: dx chain %.3f ms, dw chain %.3f ms, sequential %.3f ms, two streams %.3f ms" % (t_dx, t_dw, timed(seq), timed(par)))
gs, gp = graphed(seq), graphed(par)
print("hipGraph: sequential %.3f ms, forked %.3f ms" % (timed(gs), timed(gp)))
