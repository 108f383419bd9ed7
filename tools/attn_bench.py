"""Attention core at the 3D-LLM shapes (BASELINE config 5: B=4, 32 queries x Nk scene tokens, 12 heads):
time and achieved f32-MFMA rate of sig3d_attention_fwd / _bwd, with and without the forward key split.
python tools/attn_bench.py [nk ...]"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from situation3d_amd import _lib as L
from situation3d_amd.qformer import _fwd_key_splits
dev = torch.device("cuda", 0)
def timeit(fn, iters=30, warm=50):   # the clocks of an idle GPU ramp over the first ~20 ms: 52 vs 69 TFLOP/s for the same kernel
    for _ in range(warm): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
b, h, nq = 4, 12, 32
if "--fold" in sys.argv:   # heads folded into the batch: every (batch, head) streams CONTIGUOUS 256-byte K / V rows
    sys.argv.remove("--fold"); b, h = 48, 1
for nk in [int(a) for a in sys.argv[1:]] or [256, 5000, 20000, 80000]:
    hd = h * 64
    q, go = torch.randn(b, nq, hd, device=dev), torch.randn(b, nq, hd, device=dev)
    k, v = torch.randn(b, nk, hd, device=dev), torch.randn(b, nk, hd, device=dev)
    out, lse = torch.empty_like(q), torch.empty(b, h, nq, device=dev)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    ks, work = _fwd_key_splits(b, h, nq, nk, dev)
    def fwd(splits, w):
        L.call("sig3d_attention_fwd", b, h, nq, nk, 64, nq, nk, 0, 0, 0, 0, hd, hd, hd, ctypes.c_float(0.125), L.ptr(q),
               L.ptr(k), L.ptr(v), L.ptr(None), L.ptr(out), L.ptr(lse), ctypes.c_float(0.1), ctypes.c_uint(3), L.ptr(None),
               splits, L.ptr(w), L.stream_ptr())
    def bwd():
        L.call("sig3d_attention_bwd", b, h, nq, nk, 64, nq, nk, 0, 0, 0, 0, hd, hd, hd, ctypes.c_float(0.125), L.ptr(q),
               L.ptr(k), L.ptr(v), L.ptr(None), L.ptr(out), L.ptr(lse), L.ptr(go), L.ptr(dq), L.ptr(dk), L.ptr(dv),
               ctypes.c_float(0.1), ctypes.c_uint(3), L.ptr(None), L.stream_ptr())
    t1 = timeit(lambda: fwd(1, None)); ts = timeit(lambda: fwd(ks, work)); tb = timeit(bwd)
    ff = 4.0 * b * h * nq * nk * 64          # QK^T + PV
    fb = 10.0 * b * h * nq * nk * 64         # S, dP, dV, dK, dQ
    print("nk=%6d: fwd 1 pass %8.1f us (%5.1f TF/s) | fwd %2d splits %8.1f us (%5.1f TF/s) | bwd %8.1f us (%5.1f TF/s)"
          % (nk, t1, ff / t1 / 1e6, ks, ts, ff / ts / 1e6, tb, fb / tb / 1e6))
