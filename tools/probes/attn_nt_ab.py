"""A/B of non-temporal K / V loads in the rotating attention forward: builds a private library with -DSIG3D_ATTN_NT=1
and times both at Nk = 80 000 (B = 4, 12 heads, 32 queries)."""
import ctypes, os, subprocess, sys, tempfile
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from situation3d_amd.build import FLAGS, CSRC
from situation3d_amd import _lib as L
from situation3d_amd.qformer import _fwd_key_splits
dev = torch.device("cuda", 0)
b, h, nq, nk = 4, 12, 32, 80000
hd = h * 64
q = torch.randn(b, nq, hd, device=dev); k = torch.randn(b, nk, hd, device=dev); v = torch.randn(b, nk, hd, device=dev)
out, lse = torch.empty_like(q), torch.empty(b, h, nq, device=dev)
for nt in (0, 1):
    so = os.path.join(tempfile.mkdtemp(), "libattn_nt%d.so" % nt)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-x", "hip", "-shared", "-DSIG3D_ATTN_NT=%d" % nt,
                           os.path.join(CSRC, "attention.hip"), os.path.join(CSRC, "capi.hip"), "-o", so] + FLAGS)
    lib = ctypes.CDLL(so)
    lib.sig3d_attention_fwd.argtypes = L.SIGNATURES["sig3d_attention_fwd"]
    for splits in (21, 21, 32, 42, 64, 96, 21):
        work = torch.empty(b * h * 32 * splits * 66, device=dev)
        def fwd():
            lib.sig3d_attention_fwd(b, h, nq, nk, 64, nq, nk, 0, 0, 0, 0, hd, hd, hd, 0.125, q.data_ptr(), k.data_ptr(), v.data_ptr(),
                                    None, out.data_ptr(), lse.data_ptr(), 0.1, 3, None, splits, work.data_ptr(),
                                    torch.cuda.current_stream().cuda_stream)
        for _ in range(40): fwd()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): fwd()
        e.record(); torch.cuda.synchronize()
        t = s.elapsed_time(e) / 20 * 1e3
        print("NT=%d splits %d: %.1f us  %.1f TF/s  K+V stream %.2f TB/s" % (nt, splits, t, 4.0 * b * h * nq * nk * 64 / t / 1e6,
                                                                         2.0 * b * nk * hd * 4 / t / 1e6))
