"""Phase timing of mlp_layer_fwd_kernel (csrc/shared_mlp.hip ML_MARK points): private build with
-DSIG3D_MLP_TIMING, prints the time line of wave 0 / workgroup 0 for the SA1-L3 and SA2-L3 shapes.
marks: 0 start, 1 weights staged, 2 chunk consume begins, 3 chunk MFMAs issued, 4 epilogue done."""
import ctypes, os, subprocess, sys, tempfile
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
from situation3d_amd.build import FLAGS, CSRC
tmp = tempfile.mkdtemp()
so = os.path.join(tmp, "libmlp_timing.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "-x", "hip", "-shared", "-DSIG3D_MLP_TIMING",
                       os.path.join(CSRC, "shared_mlp.hip"), os.path.join(CSRC, "capi.hip"), os.path.join(CSRC, "group_points.hip"), "-o", so] + FLAGS)
lib = ctypes.CDLL(so)
P, I = ctypes.c_void_p, ctypes.c_int
lib.sig3d_mlp_layer_fwd.argtypes = [I, I, I, ctypes.c_long] + [P] * 7 + [I, P]
lib.sig3d_mlp_layer_fwd_compact.argtypes = [I, I, I, ctypes.c_long] + [P] * 7 + [I, P, P, P]
dev = "cuda:0"
CASES = [("SA1 L3", 8, 64, 128, 131072, 0), ("SA2 L3", 8, 128, 256, 32768, 0), ("SA2 L2", 8, 128, 128, 32768, 0),
         ("SA1 L2 compact (14 400 live of 131 072)", 8, 64, 64, 131072, 14400),
         ("SA1 L3 compact", 8, 64, 128, 131072, 14400), ("SA2 L3 compact (1 300 live)", 8, 128, 256, 32768, 1300)]
for name, b, cin, cout, e, live in CASES:
    x = torch.randn(b, cin, e, device=dev); w = torch.randn(cout, cin, device=dev)
    y = torch.empty(b, cout, e, device=dev); st = torch.empty(2, cout, dtype=torch.float64, device=dev)
    ps, pb = torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev)
    ptr = lambda t: ctypes.c_void_p(t.data_ptr())
    marks = (ctypes.c_ulonglong * 64)(); n = ctypes.c_int(0)
    for it in range(3):
        lib.sig3d_debug_mlp_marks(marks, ctypes.byref(n))
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        t_a, t_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t_a.record()
        if live:
            n_act = torch.full((b,), live, dtype=torch.int32, device=dev)
            mult = torch.ones(b, e, device=dev)
            lib.sig3d_mlp_layer_fwd_compact(b, cin, cout, e, ptr(x), ptr(w), ptr(ps), ptr(pb), ptr(y), ptr(st[0]), ptr(st[1]), 0,
                                            ptr(n_act), ptr(mult), stream)
        else:
            lib.sig3d_mlp_layer_fwd(b, cin, cout, e, ptr(x), ptr(w), ptr(ps), ptr(pb), ptr(y), ptr(st[0]), ptr(st[1]), 0, stream)
        t_b.record()
        torch.cuda.synchronize()
        if it == 2:
            print("   launch: %.1f us" % (t_a.elapsed_time(t_b) * 1e3))
    cyc = (ctypes.c_ulonglong * 64)()
    lib.sig3d_debug_mlp_cycles(cyc)
    lib.sig3d_debug_mlp_marks(marks, ctypes.byref(n))
    t0 = marks[0] & ((1 << 56) - 1)
    last = min(n.value, 64) - 1
    dt_us = ((marks[last] & ((1 << 56) - 1)) - t0) / 100.0
    print("   shader clock over the marked span: %.0f MHz (%d cycles in %.1f us)" % ((cyc[last] - cyc[0]) / dt_us, cyc[last] - cyc[0], dt_us))
    print("==", name, "marks:", n.value)
    print("  " + " ".join("%d@%.2f" % (marks[i] >> 56, ((marks[i] & ((1 << 56) - 1)) - t0) / 100.0) for i in range(min(n.value, 40))))
