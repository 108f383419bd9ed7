"""Where does mlp_layer_fwd_kernel's time go?  Private builds with the operand loads (-DML_EXP_NO_LOAD)
or the output stores (-DML_EXP_NO_STORE) compiled out, timed at the SA1 / SA2 layer shapes."""
import ctypes, os, subprocess, sys, tempfile
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
from situation3d_amd.build import FLAGS, CSRC
dev = "cuda:0"
def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
P, I = ctypes.c_void_p, ctypes.c_int
shapes = [("SA1 L1", 8, 6, 64, 131072), ("SA1 L2", 8, 64, 64, 131072), ("SA1 L3", 8, 64, 128, 131072),
          ("SA2 L2", 8, 128, 128, 32768), ("SA2 L3", 8, 128, 256, 32768)]
for label, defs in [("full", []), ("no loads", ["-DML_EXP_NO_LOAD"]), ("no stores", ["-DML_EXP_NO_STORE"]),
                    ("neither", ["-DML_EXP_NO_LOAD", "-DML_EXP_NO_STORE"]),
                    ("no transpose+store", ["-DML_EXP_NO_TRANSPOSE"]), ("no stats", ["-DML_EXP_NO_STATS"]),
                    ("mfma only", ["-DML_EXP_NO_LOAD", "-DML_EXP_NO_TRANSPOSE", "-DML_EXP_NO_STATS"])]:
    tmp = tempfile.mkdtemp(); so = os.path.join(tmp, "lib.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-x", "hip", "-shared"] + defs + [os.path.join(CSRC, "shared_mlp.hip"),
                          os.path.join(CSRC, "capi.hip"), "-o", so] + FLAGS)
    lib = ctypes.CDLL(so)
    lib.sig3d_mlp_layer_fwd.argtypes = [I, I, I, ctypes.c_long] + [P] * 7 + [I, P]
    res = []
    for name, b, cin, cout, e in shapes:
        x = torch.randn(b, cin, e, device=dev); w = torch.randn(cout, cin, device=dev)
        y = torch.empty(b, cout, e, device=dev); st = torch.empty(2, cout, dtype=torch.float64, device=dev)
        ps, pb = torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev)
        ptr = lambda t: ctypes.c_void_p(t.data_ptr())
        st_ = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        t = timeit(lambda: lib.sig3d_mlp_layer_fwd(b, cin, cout, e, ptr(x), ptr(w), ptr(ps), ptr(pb), ptr(y), ptr(st[0]),
                                                   ptr(st[1]), 0, st_))
        res.append("%s %6.1f" % (name, t))
    print("%-19s " % label + " | ".join(res))
