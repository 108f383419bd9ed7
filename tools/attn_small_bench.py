"""sig3d_attention_fwd / sig3d_attention_bwd at the two small shapes of the SQA3D step (B = 8, 12 heads of 64): the 52 x 52 self-attention in
the two-segment layout over a fused q / k / v projection, and the 32 x 256 cross-attention.  hipGraph-timed.
SIG3D_ATTN_BWD_SMALL=0 python tools/attn_small_bench.py   -> the generic backward;  =1 (default) the small-problem kernel.
(Forward, streaming kernel: 6.3 / 10.4 us.  A small-problem forward of the same build as the backward -- Q and K in LDS, the
blocks of S on eight waves, whole-row softmax in LDS, P V over key slices -- was written and measured in round 5: 6.7 / 10.2 us,
nothing gained, not kept.)"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from situation3d_amd import _lib as L

dev = torch.device("cuda", 0)
b, h, d = 8, 12, 64
hd = h * d


def graph_time(fn, reps=40):
    s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(5):
            g.replay()
        e1.record(s)
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * reps) * 1e3


def case(name, nq, nk, seg, self_attn, p_drop):
    rows_q = 512 if self_attn else b * nq
    if self_attn:
        proj = torch.randn(rows_q, 3 * hd, device=dev)
        qp, kp, vp = proj.data_ptr(), proj.data_ptr() + 4 * hd, proj.data_ptr() + 8 * hd
        dproj = torch.empty_like(proj)
        dqp, dkp, dvp = dproj.data_ptr(), dproj.data_ptr() + 4 * hd, dproj.data_ptr() + 8 * hd
        ldq = ldk = ldv = 3 * hd
        kseg, base2, live = seg, 256, 416
    else:
        proj = torch.randn(b * nq, hd, device=dev)
        kv = torch.randn(b * nk, 2 * hd, device=dev)
        dproj, dkv = torch.zeros_like(proj), torch.empty_like(kv)
        qp, kp, vp = proj.data_ptr(), kv.data_ptr(), kv.data_ptr() + 4 * hd
        dqp, dkp, dvp = dproj.data_ptr(), dkv.data_ptr(), dkv.data_ptr() + 4 * hd
        ldq, ldk, ldv = hd, 2 * hd, 2 * hd
        seg, kseg, base2, live = nq, nk, 0, 0
    out = torch.randn(rows_q, hd, device=dev)
    lse = torch.randn(b, h, nq, device=dev) + 4
    go = torch.randn(rows_q, hd, device=dev)
    ctr = torch.zeros(1, dtype=torch.int32, device=dev)
    P = ctypes.c_void_p

    def bwd():
        L.call("sig3d_attention_bwd_z" if not self_attn else "sig3d_attention_bwd", b, h, nq, nk, d, seg, kseg,
               base2 if self_attn else b * nq, base2 if self_attn else b * nk, live, live, ldq, ldk, ldv,
               ctypes.c_float(0.125), P(qp), P(kp), P(vp), P(0), L.ptr(out), L.ptr(lse), L.ptr(go), P(dqp), P(dkp), P(dvp),
               ctypes.c_float(p_drop), ctypes.c_uint(7), L.ptr(ctr), L.stream_ptr(dev))
    def fwd():
        L.call("sig3d_attention_fwd", b, h, nq, nk, d, seg, kseg, base2 if self_attn else b * nq,
               base2 if self_attn else b * nk, live, live, ldq, ldk, ldv, ctypes.c_float(0.125), P(qp), P(kp), P(vp), P(0),
               L.ptr(out), L.ptr(lse), ctypes.c_float(p_drop), ctypes.c_uint(7), L.ptr(ctr), 1, P(0), L.stream_ptr(dev))
    t_f = graph_time(fwd)
    lse.copy_(torch.randn(b, h, nq, device=dev) + 4)
    print("%-44s forward %6.1f us   backward %6.1f us" % (name + " p_drop %.1f" % p_drop, t_f, graph_time(bwd)))


for p in (0.0, 0.1):
    case("self-attention 52 x 52 (two segments, fused qkv)", 52, 52, 32, True, p)
    case("cross-attention 32 x 256", 32, 256, 32, False, p)
