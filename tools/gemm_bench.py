"""Q-Former GEMM shapes of the bench step (B = 8: 256 query rows + 160 text rows, 2048 scene tokens):
sig3d_gemm / sig3d_gemm_group (exact-f32 MFMA, csrc/gemm.hip) against the tuned library GEMM torch
dispatches to (which needs the text branch padded to 256 rows), each timed as a hipGraph of launches that
rotate over enough distinct weight buffers to stay HBM-cold (weights are touched once per step).

python tools/gemm_bench.py [--sweep]
"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from situation3d_amd import _lib as L, gemm_tuning

dev = torch.device("cuda:0")
sweep = "--sweep" in sys.argv


def problem(**kw):
    p = L.GemmProblem()
    d = dict(amode=0, bmode=0, batch=1, m=0, n=0, k=0, m_last=None, k_last=None, A=None, lda=0, stride_a=0, B=None,
             ldb=0, stride_b=0, C=None, ldc=0, stride_c=0, bias=None, stride_bias=0, act=0, aux=None, accumulate=0,
             rowsum=None, stride_rowsum=0, tile=0, ksplit=0)
    d.update(kw)
    d["m_last"] = d["m"] if d["m_last"] is None else d["m_last"]
    d["k_last"] = d["k"] if d["k_last"] is None else d["k_last"]
    for k, v in d.items():
        setattr(p, k, v.data_ptr() if isinstance(v, torch.Tensor) else v)
    return p


def group(*ps):
    arr = (L.GemmProblem * len(ps))(*ps)
    L.call("sig3d_gemm_group", len(ps), arr, L.stream_ptr(dev))


def graph_time(fn, reps):
    s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        for i in range(min(reps, 2)):
            fn(i)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for i in range(reps):
                fn(i)
        g.replay()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3):
            g.replay()
        b.record()
        torch.cuda.synchronize()
    return a.elapsed_time(b) / (3 * reps) * 1e3


gemm_tuning.enable(tune_missing=True)
P, TQ, TT, H, I = 256, 256, 160, 768, 3072
R = TQ + TT            # 416 live rows; the library path works on 2P = 512
rows = []


def nbuf(bytes_):
    return max(2, min(48, int(400e6 // max(bytes_, 1)) + 1))


def report(name, flops, t_lib, t_own, extra=""):
    rows.append((name, t_lib, t_own))
    print("%-34s %6.2f GF | library %6.1f us %5.1f TF | sig3d %6.1f us %5.1f TF  (%+.1f us) %s" % (
        name, flops / 1e9, t_lib, flops / t_lib / 1e6, t_own, flops / t_own / 1e6, t_own - t_lib, extra), flush=True)


def best_of(make, flops, tiles=(1, 2, 3, 4), splits=(1, 2, 3, 4, 6, 8, 12), reps=8):
    res = []
    for t in tiles:
        for ks in splits:
            try:
                res.append((graph_time(make(t, ks), reps), t, ks))
            except Exception:
                pass
    res.sort()
    return " | best " + " ".join("t%dk%d:%.1f" % (t, k, us) for us, t, k in res[:4])


# ---- forward products -------------------------------------------------------------------------------------
def fwd(name, m_lib, m_own, n, k, act=0):
    nb = nbuf(4 * n * k)
    X = torch.randn(m_lib, k, device=dev)
    W = [torch.randn(n, k, device=dev) * 0.05 for _ in range(nb)]
    b = torch.randn(n, device=dev)
    C = torch.zeros(m_lib, n, device=dev)
    t_lib = graph_time(lambda i: torch.addmm(b, X, W[i].t(), out=C), nb)
    mk = lambda t, ks: (lambda i: group(problem(m=m_own, n=n, k=k, A=X, lda=k, B=W[i], ldb=k, C=C, ldc=n, bias=b,
                                                 accumulate=1 if ks != 1 else 0, tile=t, ksplit=ks)))
    t_own = graph_time(mk(0, 0), nb)
    report(name, 2.0 * m_own * n * k, t_lib, t_own, best_of(mk, 0, reps=nb) if sweep else "")


fwd("qkv 416(512) x 2304 x 768", 2 * P, R, 3 * H, H)
fwd("out-proj 416(512) x 768 x 768", 2 * P, R, H, H)
fwd("cross q 256 x 768 x 768", TQ, TQ, H, H)
fwd("cross kv x6 2048 x 9216 x 256", 2048, 2048, 12 * H, 256)

# feed-forward: batch of two branches, ragged in sig3d
nb = nbuf(4 * 2 * I * H)
X = torch.randn(2 * P, H, device=dev)
W1 = [torch.randn(2, I, H, device=dev) * 0.05 for _ in range(nb)]
b1 = torch.randn(2, I, device=dev)
pre = torch.zeros(2 * P, I, device=dev)
act = torch.zeros(2 * P, I, device=dev)


def lib_up(i):
    torch.bmm(X.view(2, P, H), W1[i].transpose(1, 2), out=pre.view(2, P, I))
    L.call("sig3d_bias_gelu", 2 * P, I, P, L.ptr(pre), L.ptr(b1), None, L.ptr(act), L.stream_ptr(dev))


t_lib = graph_time(lib_up, nb)
mk = lambda t, ks: (lambda i: group(problem(batch=2, m=P, m_last=TT, n=I, k=H, A=X, lda=H, stride_a=P * H, B=W1[i], ldb=H,
                                             stride_b=I * H, C=act, ldc=I, stride_c=P * I, bias=b1, stride_bias=I, act=1,
                                             aux=pre, tile=t, ksplit=1)))
report("ffn-up (256|160) x 3072 x 768 +gelu", 2.0 * R * I * H, t_lib, graph_time(mk(0, 0), nb),
       best_of(mk, 0, splits=(1,), reps=nb) if sweep else "")
W2 = [torch.randn(2, H, I, device=dev) * 0.05 for _ in range(nb)]
Y = torch.zeros(2 * P, H, device=dev)
t_lib = graph_time(lambda i: torch.bmm(act.view(2, P, I), W2[i].transpose(1, 2), out=Y.view(2, P, H)), nb)
mk = lambda t, ks: (lambda i: group(problem(batch=2, m=P, m_last=TT, n=H, k=I, A=act, lda=I, stride_a=P * I, B=W2[i], ldb=I,
                                             stride_b=H * I, C=Y, ldc=H, stride_c=P * H, accumulate=1, tile=t, ksplit=ks)))
report("ffn-down (256|160) x 768 x 3072", 2.0 * R * I * H, t_lib, graph_time(mk(0, 0), nb),
       best_of(mk, 0, reps=nb) if sweep else "")

# ---- backward pairs: dX and dW of one layer in one launch ---------------------------------------------------
dyo = torch.randn(2 * P, H, device=dev) * 0.1
gw2 = torch.zeros(2, H, I, device=dev)
gact = torch.zeros(2 * P, I, device=dev)
gb = torch.zeros(2, I, device=dev)


def lib_ffn_bwd1(i):
    torch.bmm(dyo.view(2, P, H).transpose(1, 2), act.view(2, P, I), out=gw2)
    torch.bmm(dyo.view(2, P, H), W2[i], out=gact.view(2, P, I))
    L.call("sig3d_bias_gelu", 2 * P, I, P, L.ptr(pre), L.ptr(b1), L.ptr(gact), L.ptr(gact), L.stream_ptr(dev))
    L.call("sig3d_column_sum", 2, P, I, L.ptr(gact), L.ptr(gb), L.stream_ptr(dev))


t_lib = graph_time(lib_ffn_bwd1, nb)
mk = lambda t, ks: (lambda i: group(
    problem(amode=0, bmode=1, batch=2, m=P, m_last=TT, n=I, k=H, A=dyo, lda=H, stride_a=P * H, B=W2[i], ldb=I,
            stride_b=H * I, C=gact, ldc=I, stride_c=P * I, act=2, aux=pre, tile=t, ksplit=1),
    problem(amode=1, bmode=1, batch=2, m=H, n=I, k=P, k_last=TT, A=dyo, lda=H, stride_a=P * H, B=act, ldb=I,
            stride_b=P * I, C=gw2, ldc=I, stride_c=H * I, tile=t, ksplit=1)))
report("ffn bwd 1: gact*gelu' + gW2 (grouped)", 4.0 * R * I * H, t_lib, graph_time(mk(0, 0), nb),
       best_of(mk, 0, splits=(1,), reps=nb) if sweep else "")
gw1 = torch.zeros(2, I, H, device=dev)
gx = torch.zeros(2 * P, H, device=dev)


def lib_ffn_bwd2(i):
    torch.bmm(gact.view(2, P, I).transpose(1, 2), X.view(2, P, H), out=gw1)
    gx.view(2, P, H).baddbmm_(gact.view(2, P, I), W1[i])


t_lib = graph_time(lib_ffn_bwd2, nb)
mk = lambda t, ks: (lambda i: group(
    problem(amode=0, bmode=1, batch=2, m=P, m_last=TT, n=H, k=I, A=gact, lda=I, stride_a=P * I, B=W1[i], ldb=H,
            stride_b=I * H, C=gx, ldc=H, stride_c=P * H, accumulate=1, tile=t, ksplit=ks),
    problem(amode=1, bmode=1, batch=2, m=I, n=H, k=P, k_last=TT, A=gact, lda=I, stride_a=P * I, B=X, ldb=H,
            stride_b=P * H, C=gw1, ldc=H, stride_c=I * H, rowsum=gb, stride_rowsum=I, tile=t, ksplit=1)))
report("ffn bwd 2: gx+= + gW1 + gb1 (grouped)", 4.0 * R * I * H, t_lib, graph_time(mk(0, 0), nb),
       best_of(mk, 0, reps=nb) if sweep else "")

nb = nbuf(4 * 3 * H * H)
Wqkv = [torch.randn(3 * H, H, device=dev) * 0.05 for _ in range(nb)]
dproj = torch.randn(2 * P, 3 * H, device=dev) * 0.1
gwq = torch.zeros(3 * H, H, device=dev)
gbq = torch.zeros(3 * H, device=dev)


def lib_attn_bwd(i):
    gx.addmm_(dproj, Wqkv[i])
    torch.mm(dproj.t(), X, out=gwq)
    L.call("sig3d_column_sum", 1, 2 * P, 3 * H, L.ptr(dproj), L.ptr(gbq), L.stream_ptr(dev))


t_lib = graph_time(lib_attn_bwd, nb)
mk = lambda t, ks: (lambda i: group(
    problem(amode=0, bmode=1, m=R, n=H, k=3 * H, A=dproj, lda=3 * H, B=Wqkv[i], ldb=H, C=gx, ldc=H, accumulate=1, tile=t,
            ksplit=ks),
    problem(amode=1, bmode=1, m=3 * H, n=H, k=R, A=dproj, lda=3 * H, B=X, ldb=H, C=gwq, ldc=H, rowsum=gbq, tile=t,
            ksplit=1)))
report("attn bwd: gx+= dproj Wqkv + gWqkv + gb", 4.0 * R * 3 * H * H, t_lib, graph_time(mk(0, 0), nb),
       best_of(mk, 0, reps=nb) if sweep else "")
Wo = [torch.randn(H, H, device=dev) * 0.05 for _ in range(nb)]
att = torch.randn(2 * P, H, device=dev)
gwo = torch.zeros(H, H, device=dev)
datt = torch.zeros(2 * P, H, device=dev)


def lib_out_bwd(i):
    torch.mm(dyo.t(), att, out=gwo)
    torch.mm(dyo, Wo[i], out=datt)


t_lib = graph_time(lib_out_bwd, nb)
mk = lambda t, ks: (lambda i: group(
    problem(amode=0, bmode=1, m=R, n=H, k=H, A=dyo, lda=H, B=Wo[i], ldb=H, C=datt, ldc=H, accumulate=1, tile=t, ksplit=ks),
    problem(amode=1, bmode=1, m=H, n=H, k=R, A=dyo, lda=H, B=att, ldb=H, C=gwo, ldc=H, accumulate=1, tile=t, ksplit=ks)))
report("out-proj bwd: datt + gWo (grouped)", 4.0 * R * H * H, t_lib, graph_time(mk(0, 0), nb),
       best_of(mk, 0, reps=nb) if sweep else "")

tl, to = sum(r[1] for r in rows), sum(r[2] for r in rows)
print("sum over the rows above: library %.1f us, sig3d %.1f us" % (tl, to))
