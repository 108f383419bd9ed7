"""Phase timing of attention_bwd_kernel: builds a private copy of the library with
-DSIG3D_ATTN_TIMING (csrc/attention.hip AT_MARK points, 100 MHz real-time counter of workgroup
(0,0,0)) and prints the deltas between marks for the Q-Former's two shapes.

python tools/attn_timing.py          # needs the GPU
"""
import ctypes, os, subprocess, sys, tempfile
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
from situation3d_amd.build import FLAGS, CSRC

tmp = tempfile.mkdtemp()
so = os.path.join(tmp, "libattn_timing.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "-x", "hip", "-shared", "-DSIG3D_ATTN_TIMING",
                       os.path.join(CSRC, "attention.hip"), os.path.join(CSRC, "capi.hip"), "-o", so] + FLAGS)
lib = ctypes.CDLL(so)
P, I, F = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
lib.sig3d_attention_fwd.argtypes = [I] * 14 + [F] + [P] * 6 + [F, ctypes.c_uint, P, I, P, P]
lib.sig3d_attention_bwd.argtypes = [I] * 14 + [F] + [P] * 10 + [F, ctypes.c_uint, P, P]
NAMES = ["start", "D+zero done", "K/V loaded", "q operands issued", "S done", "dP+ds done", "dV/dK done",
         "dQ mfma done", "dq lds-atomics done", "tiles done", "barrier", "dq written"]
dev = "cuda:0"
for (b, h, nq, nk, label) in [(8, 12, 52, 52, "self 52x52"), (8, 12, 32, 256, "cross 32x256")]:
    hd = h * 64
    q, k, v, go = (torch.randn(b, n, hd, device=dev) for n in (nq, nk, nk, nq))
    out, lse = torch.empty_like(q), torch.empty(b, h, nq, device=dev)
    dq, dk, dv = torch.zeros_like(q), torch.empty_like(k), torch.empty_like(v)
    ptr = lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        lib.sig3d_attention_fwd(b, h, nq, nk, 64, nq, nk, 0, 0, 0, 0, hd, hd, hd, 0.125, ptr(q), ptr(k), ptr(v), None,
                                ptr(out), ptr(lse), 0.1, 7, None, 1, None, st)
        lib.sig3d_attention_bwd(b, h, nq, nk, 64, nq, nk, 0, 0, 0, 0, hd, hd, hd, 0.125, ptr(q), ptr(k), ptr(v), None,
                                ptr(out), ptr(lse), ptr(go), ptr(dq), ptr(dk), ptr(dv), 0.1, 7, None, st)
    torch.cuda.synchronize()
    marks = (ctypes.c_ulonglong * (2 * 4 * 16))()
    assert lib.sig3d_debug_attention_marks(marks) == 0
    print("==", label)
    for wave in range(4):
        m = [marks[(1 * 4 + wave) * 16 + i] for i in range(12)]
        base = marks[(1 * 4 + 0) * 16 + 0]
        print(" wave %d: " % wave + "  ".join("%s@%.2f" % (NAMES[i].split()[0], (m[i] - base) / 100.0) for i in range(12) if m[i]))
