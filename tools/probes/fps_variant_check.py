"""FPS indices of a VARIANT build of the library against the product build (tools only), and its time alone:
python tools/probes/fps_variant_check.py tools/probes/libvariant.so"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench
dev = torch.device("cuda", 0)
P, I = ctypes.c_void_p, ctypes.c_int
here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
libs = {"product": ctypes.CDLL(os.path.join(here, "situation3d_amd", "libsig3d_hip.so")), "variant": ctypes.CDLL(os.path.abspath(sys.argv[1]))}
for l in libs.values():
    l.sig3d_furthest_point_sampling.argtypes = [I, I, I, P, P, P, P]
def fps(lib, xyz, m):
    b, n, _ = xyz.shape
    temp = torch.empty(b, n, device=dev); idx = torch.empty(b, m, dtype=torch.int32, device=dev)
    rc = lib.sig3d_furthest_point_sampling(b, n, m, xyz.data_ptr(), temp.data_ptr(), idx.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    return idx
for b, n, m, dup in [(8, 40000, 2048, 0), (3, 30000, 700, 3000), (2, 40960, 500, 0), (1, 24577, 300, 500)]:
    xyz = bench.synthetic_batch(b, n, 5 + n, dev)["point_clouds"][..., :3].contiguous()
    if dup:
        xyz[:, n - dup:] = xyz[:, :dup]
        xyz[:, 100:160] = 0
    a, c = fps(libs["product"], xyz, m), fps(libs["variant"], xyz, m)
    torch.cuda.synchronize()
    print("B=%d N=%d M=%d dup=%d: identical indices %s" % (b, n, m, dup, bool(torch.equal(a, c))))
xyz = bench.synthetic_batch(8, 40000, 5, dev)["point_clouds"][..., :3].contiguous()
for name, lib in libs.items():
    for _ in range(2): fps(lib, xyz, 2048)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): fps(lib, xyz, 2048)
    e.record(); torch.cuda.synchronize()
    print("%s: %.1f us alone (B=8, 40000 -> 2048)" % (name, s.elapsed_time(e) / 5 * 1e3))
