"""The nine native PointNet++ entry points, same names / positional order / return types as
the reference's pybind module (lib/pointnet2/_ext_src/src/bindings.cpp:6-19), implemented by
libsig3d_hip.so through its C ABI (include/sig3d_hip.h).

Host-side behaviour mirrors the reference C++ wrappers: contiguity / dtype / device checks
(include/utils.h:5-25) raise RuntimeError, CPU tensors raise "CPU not supported"
(e.g. ball_query.cpp:27-29), outputs are allocated here on the input's device, and kernels are
enqueued on torch's current stream.  There is no CPU implementation in this module by design.
"""
import ctypes
import os

import torch

from .. import _lib


# SIG3D_NESTED_FPS=0: always run the dependent rounds (A/B timing; results are identical)
NESTED_FPS = os.environ.get("SIG3D_NESTED_FPS", "1") != "0"
# SIG3D_FPS_BLOCKS=1: scenes above 8192 points on sig3d_furthest_point_sampling_blocks (one workgroup per scene over a
# Morton-ordered copy in L2) instead of the cooperative register-resident kernel behind the reference's own argument list.
# Same indices.  Measured: in the steady state of a long run the 4-wave form beside the training
# step is worth -0.07 ms per step, but its chain takes 10-14 ms beside the step (7.7 for the cooperative kernel) and the
# 20 timed steps of bench.py end 0.1 ms per step later; alone and in forward-only serving the cooperative kernel is level
# or ahead.  Not the default.
FPS_BLOCKS = os.environ.get("SIG3D_FPS_BLOCKS", "0") != "0"
FPS_WAVES = 16      # waves per scene of that kernel for a stand-alone call (nothing runs beside it: latency counts)


def _check_contiguous(t, name):
    if not t.is_contiguous():
        raise RuntimeError("%s must be a contiguous tensor" % name)


def _check_float(t, name):
    if t.dtype != torch.float32:
        raise RuntimeError("%s must be a float tensor" % name)


def _check_int(t, name):
    if t.dtype != torch.int32:
        raise RuntimeError("%s must be an int tensor" % name)


def _run(name, dev, *args):
    with torch.cuda.device(dev):
        _lib.call(name, *args, _lib.stream_ptr(dev))


def gather_points(points, idx):
    """sampling.cpp:15-38: points (B,C,N) f32, idx (B,M) i32 -> (B,C,M)."""
    _check_contiguous(points, "points"); _check_contiguous(idx, "idx")
    _check_float(points, "points"); _check_int(idx, "idx")
    dev = _lib.require_device(points, idx)
    b, c, n = points.shape
    m = idx.shape[1]
    out = torch.empty((b, c, m), dtype=torch.float32, device=dev)
    _run("sig3d_gather_points", dev, b, c, n, m, _lib.ptr(points), _lib.ptr(idx), _lib.ptr(out))
    return out


def gather_points_grad(grad_out, idx, n):
    """sampling.cpp:40-65: grad_out (B,C,M), idx (B,M) -> (B,C,n) scatter-add."""
    _check_contiguous(grad_out, "grad_out"); _check_contiguous(idx, "idx")
    _check_float(grad_out, "grad_out"); _check_int(idx, "idx")
    dev = _lib.require_device(grad_out, idx)
    b, c, m = grad_out.shape
    out = torch.empty((b, c, int(n)), dtype=torch.float32, device=dev)
    _run("sig3d_gather_points_grad", dev, b, c, int(n), m, _lib.ptr(grad_out), _lib.ptr(idx),
         _lib.ptr(out))
    return out


def furthest_point_sampling(points, nsamples):
    """sampling.cpp:66-87: points (B,N,3) -> (B,nsamples) i32."""
    _check_contiguous(points, "points"); _check_float(points, "points")
    dev = _lib.require_device(points)
    b, n, _ = points.shape
    nsamples = int(nsamples)
    out = torch.zeros((b, nsamples), dtype=torch.int32, device=dev)
    tmp = torch.empty((b, n), dtype=torch.float32, device=dev)
    if NESTED_FPS and 1 < nsamples <= n <= 8192:
        # stacked set-abstraction levels sample the FPS-ordered centres of the level above: the nested entry point
        # proves the answer 0..m-1 instead of running the dependent rounds, and returns exactly what the plain
        # one returns for ANY input (an unordered cloud fails the proof within its first steps)
        flags = torch.empty((b,), dtype=torch.int32, device=dev)
        _run("sig3d_furthest_point_sampling_nested", dev, b, n, nsamples, _lib.ptr(points), _lib.ptr(tmp),
             _lib.ptr(out), _lib.ptr(flags))
        return out
    if FPS_BLOCKS and n > 8192:
        # one workgroup per scene over a Morton-ordered copy in L2 (csrc/sampling.hip: fps_blocks_kernel); the
        # reference's (B, N) `temp` is too small for it, so it takes a workspace of its own
        work = _lib.fps_workspace(b, n, dev)
        _run("sig3d_furthest_point_sampling_blocks", dev, b, n, nsamples, _lib.ptr(points), _lib.ptr(work),
             work.numel(), FPS_WAVES, _lib.ptr(out))
        return out
    _run("sig3d_furthest_point_sampling", dev, b, n, nsamples, _lib.ptr(points), _lib.ptr(tmp),
         _lib.ptr(out))
    return out


def furthest_point_sampling_nested(points, nsamples, return_proven=False):
    """furthest_point_sampling for a cloud that is itself FPS output in pick order (the centres of the
    SA level above): identical indices for ANY input; scenes whose result is provably 0..m-1 skip the
    dependent rounds (csrc/sampling.hip: fps_prefix_check_kernel).  `return_proven` also returns the
    (B,) int32 flags, 1 where the proof held."""
    _check_contiguous(points, "points"); _check_float(points, "points")
    dev = _lib.require_device(points)
    b, n, _ = points.shape
    nsamples = int(nsamples)
    out = torch.zeros((b, nsamples), dtype=torch.int32, device=dev)
    tmp = torch.empty((b, max(n, 1)), dtype=torch.float32, device=dev)
    flags = torch.zeros((max(b, 1),), dtype=torch.int32, device=dev)
    _run("sig3d_furthest_point_sampling_nested", dev, b, n, nsamples, _lib.ptr(points), _lib.ptr(tmp),
         _lib.ptr(out), _lib.ptr(flags))
    return (out, flags[:b]) if return_proven else out


def furthest_point_sampling_nested_chain(points, nsamples):
    """Levels of nested sampling in one proof (sig3d_fps_nested_chain): level l draws nsamples[l] points from the
    output of level l-1 (level 0 from `points` (B, N, 3)), exactly as furthest_point_sampling_nested + a gather of the
    centres level by level.  -> ([idx_l (B, m_l) i32], [xyz_l (B, m_l, 3)], proven (levels, B) i32)."""
    _check_contiguous(points, "points"); _check_float(points, "points")
    dev = _lib.require_device(points)
    b, n, _ = points.shape
    ms = [int(m) for m in nsamples]
    idxs = [torch.zeros((b, m), dtype=torch.int32, device=dev) for m in ms]
    cent = [torch.zeros((b, m, 3), dtype=torch.float32, device=dev) for m in ms]
    flags = torch.zeros((len(ms), max(b, 1)), dtype=torch.int32, device=dev)
    tmp = torch.empty((b, max(n, 1)), dtype=torch.float32, device=dev)
    _run("sig3d_fps_nested_chain", dev, b, n, len(ms), (ctypes.c_int * len(ms))(*ms), _lib.ptr(points), _lib.ptr(tmp),
         (ctypes.c_void_p * len(ms))(*[t.data_ptr() for t in idxs]),
         (ctypes.c_void_p * len(ms))(*[t.data_ptr() for t in cent]), _lib.ptr(flags))
    return idxs, cent, flags[:, :b]


def three_nn(unknowns, knows):
    """interpolate.cpp:14-40: -> [dist2 (B,n,3) f32, idx (B,n,3) i32]."""
    _check_contiguous(unknowns, "unknowns"); _check_contiguous(knows, "knows")
    _check_float(unknowns, "unknowns"); _check_float(knows, "knows")
    dev = _lib.require_device(unknowns, knows)
    b, n, _ = unknowns.shape
    m = knows.shape[1]
    idx = torch.empty((b, n, 3), dtype=torch.int32, device=dev)
    dist2 = torch.empty((b, n, 3), dtype=torch.float32, device=dev)
    _run("sig3d_three_nn", dev, b, n, m, _lib.ptr(unknowns), _lib.ptr(knows), _lib.ptr(dist2),
         _lib.ptr(idx))
    return [dist2, idx]


def three_interpolate(points, idx, weight):
    """interpolate.cpp:42-70: points (B,C,m), idx/weight (B,n,3) -> (B,C,n)."""
    _check_contiguous(points, "points"); _check_contiguous(idx, "idx")
    _check_contiguous(weight, "weight")
    _check_float(points, "points"); _check_int(idx, "idx"); _check_float(weight, "weight")
    dev = _lib.require_device(points, idx, weight)
    b, c, m = points.shape
    n = idx.shape[1]
    out = torch.empty((b, c, n), dtype=torch.float32, device=dev)
    _run("sig3d_three_interpolate", dev, b, c, m, n, _lib.ptr(points), _lib.ptr(idx),
         _lib.ptr(weight), _lib.ptr(out))
    return out


def three_interpolate_grad(grad_out, idx, weight, m):
    """interpolate.cpp:71-99: grad_out (B,C,n) -> (B,C,m)."""
    _check_contiguous(grad_out, "grad_out"); _check_contiguous(idx, "idx")
    _check_contiguous(weight, "weight")
    _check_float(grad_out, "grad_out"); _check_int(idx, "idx"); _check_float(weight, "weight")
    dev = _lib.require_device(grad_out, idx, weight)
    b, c, n = grad_out.shape
    out = torch.empty((b, c, int(m)), dtype=torch.float32, device=dev)
    _run("sig3d_three_interpolate_grad", dev, b, c, n, int(m), _lib.ptr(grad_out), _lib.ptr(idx),
         _lib.ptr(weight), _lib.ptr(out))
    return out


def ball_query(new_xyz, xyz, radius, nsample):
    """ball_query.cpp:8-32 -- note the argument order: new_xyz first."""
    _check_contiguous(new_xyz, "new_xyz"); _check_contiguous(xyz, "xyz")
    _check_float(new_xyz, "new_xyz"); _check_float(xyz, "xyz")
    dev = _lib.require_device(new_xyz, xyz)
    b, m, _ = new_xyz.shape
    n = xyz.shape[1]
    nsample = int(nsample)
    idx = torch.empty((b, m, nsample), dtype=torch.int32, device=dev)
    if n >= GRID_MIN_POINTS and radius > 0:
        # centres binned into cells, points streamed once: same output bit for bit (csrc/ball_query.hip)
        work = torch.empty(ball_query_workspace_bytes(b, m), dtype=torch.uint8, device=dev)
        _run("sig3d_ball_query_grid", dev, b, n, m, ctypes.c_float(radius), nsample, _lib.ptr(new_xyz),
             _lib.ptr(xyz), _lib.ptr(idx), _lib.ptr(work), work.numel())
        return idx
    _run("sig3d_ball_query", dev, b, n, m, ctypes.c_float(radius), nsample, _lib.ptr(new_xyz),
         _lib.ptr(xyz), _lib.ptr(idx))
    return idx


GRID_MIN_POINTS = 256   # below: the ordered brute-force scan (a chain of n / 64 steps) is the right tool


def ball_query_workspace_bytes(b, m):
    """Scratch of sig3d_ball_query_grid (include/sig3d_hip.h): a counter and 256 list slots per centre."""
    return max(b * m * 4 * 257, 16)


def ball_query_levels(problems, workspace=None):
    """Several ball queries in ONE launch pair (sig3d_ball_query_levels): problems = [(new_xyz, xyz, radius,
    nsample)] in ball_query's argument order, all with the same batch size and radius > 0 -> [idx]."""
    outs, recs = [], []
    dev = None
    for new_xyz, xyz, radius, nsample in problems:
        _check_contiguous(new_xyz, "new_xyz"); _check_contiguous(xyz, "xyz")
        _check_float(new_xyz, "new_xyz"); _check_float(xyz, "xyz")
        dev = _lib.require_device(new_xyz, xyz)
        idx = torch.empty((new_xyz.shape[0], new_xyz.shape[1], int(nsample)), dtype=torch.int32, device=dev)
        outs.append(idx)
        recs.append((xyz, new_xyz, radius, nsample, idx))
    if not recs:
        return outs
    b = recs[0][0].shape[0]
    arr = _lib.bq_levels(recs)
    need = _lib.bq_levels_workspace_bytes(b, arr)
    if need < 0:
        raise RuntimeError("too many centres for one multi-level ball query")
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(max(need, 16), dtype=torch.uint8, device=dev)
    _run("sig3d_ball_query_levels", dev, b, len(arr), arr, _lib.ptr(workspace), workspace.numel())
    return outs


def group_points(points, idx):
    """group_points.cpp:12-36: points (B,C,N), idx (B,P,S) -> (B,C,P,S)."""
    _check_contiguous(points, "points"); _check_contiguous(idx, "idx")
    _check_float(points, "points"); _check_int(idx, "idx")
    dev = _lib.require_device(points, idx)
    b, c, n = points.shape
    _, npoints, nsample = idx.shape
    out = torch.empty((b, c, npoints, nsample), dtype=torch.float32, device=dev)
    _run("sig3d_group_points", dev, b, c, n, npoints, nsample, _lib.ptr(points), _lib.ptr(idx),
         _lib.ptr(out))
    return out


def group_points_grad(grad_out, idx, n):
    """group_points.cpp:38-62: grad_out (B,C,P,S), idx (B,P,S) -> (B,C,n)."""
    _check_contiguous(grad_out, "grad_out"); _check_contiguous(idx, "idx")
    _check_float(grad_out, "grad_out"); _check_int(idx, "idx")
    dev = _lib.require_device(grad_out, idx)
    b, c, npoints, nsample = grad_out.shape
    out = torch.empty((b, c, int(n)), dtype=torch.float32, device=dev)
    _run("sig3d_group_points_grad", dev, b, c, int(n), npoints, nsample, _lib.ptr(grad_out),
         _lib.ptr(idx), _lib.ptr(out))
    return out
