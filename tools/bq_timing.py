"""Phase timing of bqc_scatter_kernel (cell role): builds a private copy of the library with -DSIG3D_BQ_TIMING
(csrc/ball_query.hip BQ_MARK points, 100 MHz real-time counter of the first cell-role workgroup) and prints the
deltas between marks at the SA1 shape of BASELINE config 3.      python tools/bq_timing.py   # needs the GPU"""
import ctypes, os, subprocess, sys, tempfile
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
from situation3d_amd.build import FLAGS, CSRC
from situation3d_amd import _lib as L
import bench

tmp = tempfile.mkdtemp()
so = os.path.join(tmp, "libbq_timing.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "-x", "hip", "-shared", "-DSIG3D_BQ_TIMING",
                       os.path.join(CSRC, "ball_query.hip"), os.path.join(CSRC, "capi.hip"), "-o", so] + FLAGS)
lib = ctypes.CDLL(so)
lib.sig3d_ball_query_levels.argtypes = L.SIGNATURES["sig3d_ball_query_levels"]
lib.sig3d_ball_query_levels_workspace_bytes.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(L.BqLevel)]
lib.sig3d_ball_query_levels_workspace_bytes.restype = ctypes.c_long
NAMES = ["start", "zeroed", "counted", "scanned", "table", "ranges", "passA", "reserved", "passB"]
dev = torch.device("cuda", 0)
b, n, m = 8, 40000, 2048
for surface in (False, True):
    xyz = bench.synthetic_batch(b, n, 3, dev, surface=surface)["point_clouds"][..., :3].contiguous()
    from situation3d_amd.pointnet2 import _ext
    inds = _ext.furthest_point_sampling(xyz, m)
    new_xyz = torch.gather(xyz, 1, inds.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    idx = torch.empty(b, m, 64, dtype=torch.int32, device=dev)
    arr = L.bq_levels([(xyz, new_xyz, 0.2, 64, idx)])
    work = torch.empty(lib.sig3d_ball_query_levels_workspace_bytes(b, 1, arr), dtype=torch.uint8, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(5):
        assert lib.sig3d_ball_query_levels(b, 1, arr, ctypes.c_void_p(work.data_ptr()), work.numel(), st) == 0
    torch.cuda.synchronize()
    marks = (ctypes.c_ulonglong * 16)()
    assert lib.sig3d_debug_bq_marks(marks) == 0
    print("surface" if surface else "uniform", "  ".join("%s@%.2f" % (NAMES[i], (marks[i] - marks[0]) / 100.0) for i in range(9)), "(us)")
