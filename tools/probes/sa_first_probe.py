"""Where does sig3d_sa_first_layer_fwd spend its time at the SA1 compact shape: with / without batch statistics,
and the MFMA layer kernel on the stored compact tensor beside it."""
import ctypes, sys
import torch
sys.path.insert(0, '.')
import bench
from situation3d_amd import _lib as L
from situation3d_amd.pointnet2 import _ext, fused_mlp
dev = torch.device("cuda", 0)
b, n, m, ns, radius = 8, 40000, 2048, 64, 0.2
pc = bench.synthetic_batch(b, n, 3, dev)["point_clouds"]
xyz = pc[..., :3].contiguous()
inds = _ext.furthest_point_sampling(xyz, m)
new_xyz = torch.gather(xyz, 1, inds.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
idx = _ext.ball_query(new_xyz, xyz, radius, ns)
cl = fused_mlp.compact_lists(idx)
cidx, cent, mult, seg, nact = cl.tensors()
e = m * ns
w = torch.randn(64, 6, device=dev)
y = torch.empty(b, 64, e, device=dev)
st = torch.zeros(2, 64, dtype=torch.float64, device=dev)
def t(fn, it=20):
    for _ in range(5): fn()
    s, e2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e2.record(); torch.cuda.synchronize()
    return s.elapsed_time(e2) / it * 1e3
def fwd(stats, compact=True):
    L.call("sig3d_sa_first_layer_fwd", b, n, m, ns, 6, 6, 64, 1, ctypes.c_float(radius), L.ptr(pc), L.ptr(new_xyz),
           L.ptr(cidx if compact else idx.view(b, e)), L.ptr(cent if compact else None), L.ptr(nact if compact else None),
           L.ptr(mult if compact else None), L.ptr(w), L.ptr(y), L.ptr(st[0] if stats else None), L.ptr(st[1] if stats else None), 1,
           L.stream_ptr())
print("active positions per scene:", nact.tolist())
print("compact, with statistics   : %.1f us" % t(lambda: fwd(True)))
print("compact, without statistics: %.1f us" % t(lambda: fwd(False)))
print("dense,   with statistics   : %.1f us" % t(lambda: fwd(True, False)))
print("dense,   without statistics: %.1f us" % t(lambda: fwd(False, False)))
x = torch.randn(b, 6, e, device=dev)
print("MFMA layer kernel on the stored compact tensor (6 -> 64, statistics): %.1f us" % t(lambda: L.call(
    "sig3d_mlp_layer_fwd_compact", b, 6, 64, e, L.ptr(x), L.ptr(w), L.ptr(None), L.ptr(None), L.ptr(y), L.ptr(st[0]), L.ptr(st[1]), 1,
    L.ptr(nact), L.ptr(mult), L.stream_ptr())))
dY = torch.randn(b, 64, e, device=dev); dW = torch.zeros(64, 6, device=dev)
print("first-layer dW from the scan (compact): %.1f us" % t(lambda: L.call(
    "sig3d_sa_first_layer_dw", b, n, m, ns, 6, 6, 64, 1, ctypes.c_float(radius), L.ptr(pc), L.ptr(new_xyz), L.ptr(cidx), L.ptr(cent),
    L.ptr(nact), L.ptr(dY), L.ptr(dW), 1, L.stream_ptr())))
print("MFMA dW on the stored compact tensor: %.1f us" % t(lambda: L.call(
    "sig3d_mlp_layer_dw_compact", b, 6, 64, e, L.ptr(dY), L.ptr(x), L.ptr(None), L.ptr(None), L.ptr(dW), 1, L.ptr(nact), L.stream_ptr())))
