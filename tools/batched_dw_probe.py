"""Would deferring the Q-Former's weight-gradient GEMMs to the end of backward and running them batched over
the 12 layers pay?  Library GEMMs, hipGraph-timed: 12 separate launches (as today) vs one strided-batched launch."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from situation3d_amd import gemm_tuning, _lib as L
dev = torch.device("cuda:0")
gemm_tuning.enable(tune_missing=True)


def graph_time(fn, inner=1):
    s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        fn(); fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            fn()
        g.replay(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            g.replay()
        b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / 5 * 1e3


NL, P, H, I, R = 12, 256, 768, 3072, 416
cases = [("gW2 = dyo^T act   (2 x 768 x 3072, k 256)", (NL, 2, P, H), (NL, 2, P, I)),
         ("gW1 = gpre^T x    (2 x 3072 x 768, k 256)", (NL, 2, P, I), (NL, 2, P, H)),
         ("gWqkv = dproj^T x (2304 x 768, k 416)", (NL, 1, R, 3 * H), (NL, 1, R, H)),
         ("gWo = dyo^T att   (768 x 768, k 416)", (NL, 1, R, H), (NL, 1, R, H))]
tot_s = tot_b = 0
for name, sa, sb in cases:
    A = torch.randn(*sa, device=dev) * 0.1
    B = torch.randn(*sb, device=dev)
    nb, m, n = sa[0] * sa[1], sa[3], sb[3]
    out = torch.empty(nb, m, n, device=dev)
    cs = torch.empty(nb, m, device=dev)

    def separate():
        for l in range(NL):
            if sa[1] == 2:
                torch.bmm(A[l].transpose(1, 2), B[l], out=out[2 * l:2 * l + 2])
                L.call("sig3d_column_sum", 2, sa[2], m, L.ptr(A[l]), L.ptr(cs[2 * l]), L.stream_ptr(dev))
            else:
                torch.mm(A[l, 0].t(), B[l, 0], out=out[l])
                L.call("sig3d_column_sum", 1, sa[2], m, L.ptr(A[l]), L.ptr(cs[l]), L.stream_ptr(dev))

    def batched():
        torch.bmm(A.view(nb, sa[2], m).transpose(1, 2), B.view(nb, sb[2], n), out=out)
        L.call("sig3d_column_sum", nb, sa[2], m, L.ptr(A), L.ptr(cs), L.stream_ptr(dev))

    ts, tb = graph_time(separate), graph_time(batched)
    fl = 2.0 * nb * m * n * sa[2]
    tot_s += ts; tot_b += tb
    print("%-46s %5.1f GF | 12 launches (+12 column sums) %7.1f us %5.1f TF | one batched launch (+1) %7.1f us %5.1f TF" % (
        name, fl / 1e9, ts, fl / ts / 1e6, tb, fl / tb / 1e6))
print("total: separate %.1f us, batched %.1f us, saving %.1f us per step" % (tot_s, tot_b, tot_s - tot_b))
