"""ctypes binding of libsig3d_hip.so -- the only way product code reaches the HIP kernels.

The library is looked up in-tree (next to this file).  There is NO CPU fallback: if the
shared object is missing or a symbol cannot be resolved this raises, and every op that needs
it fails loudly.  Mirrors the C ABI declared in include/sig3d_hip.h.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsig3d_hip.so")

_P = ctypes.c_void_p
_I = ctypes.c_int
_F = ctypes.c_float

# name -> argtypes (restype is int status everywhere); order == include/sig3d_hip.h
SIGNATURES = {
    "sig3d_furthest_point_sampling": [_I, _I, _I, _P, _P, _P, _P],
    "sig3d_furthest_point_sampling_blocks": [_I, _I, _I, _P, _P, ctypes.c_long, _I, _P, _P],
    "sig3d_furthest_point_sampling_nested": [_I, _I, _I, _P, _P, _P, _P, _P],
    "sig3d_fps_nested_chain": [_I, _I, _I, _P, _P, _P, _P, _P, _P, _P],
    "sig3d_fps_timeout_count": [_P, _I],
    "sig3d_timestamp": [_P, _P],
    "sig3d_timestamp_rate": [_I, _P],
    "sig3d_gather_points": [_I, _I, _I, _I, _P, _P, _P, _P],
    "sig3d_gather_points_grad": [_I, _I, _I, _I, _P, _P, _P, _P],
    "sig3d_gather_xyz": [_I, _I, _I, _P, _P, _P, _P],
    "sig3d_ball_query": [_I, _I, _I, _F, _I, _P, _P, _P, _P],
    "sig3d_ball_query_grid": [_I, _I, _I, _F, _I, _P, _P, _P, _P, ctypes.c_long, _P],
    "sig3d_compact_neighbour_lists": [_I, _I, _I, _P, _P, _P, _P, _P, _P, _P],
    "sig3d_query_group_compact": [_I, _I, _I, _I, _I, _I, _I, _I, _F, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "sig3d_query_group_compact_grad": [_I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _I, _P, _P],
    "sig3d_mlp_layer0_gather_fwd": [_I, _I, _I, _I, _I, _I, _I, ctypes.c_float, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P],
    "sig3d_mlp_layer0_gather_dw": [_I, _I, _I, _I, _I, _I, _I, ctypes.c_float, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P],
    "sig3d_mlp_layer0_scatter_dx": [_I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P],
    "sig3d_mlp_layer0_scatter_dx_w": [_I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P],
    "sig3d_mlp_layer_dx": [_I, _I, _I, ctypes.c_long, _P, _P, _P, _P, _P],
    "sig3d_mlp_layer_dw_stream": [_I, _I, _I, ctypes.c_long, _P, _P, _P, _P, _P, _P, _P, _P],
    "sig3d_mlp_layer_dw_stream_nofold": [_I, _I, _I, ctypes.c_long, _P, _P, _P, _P, _P, _P, _P, _P],
    "sig3d_mlp_layer_dw_dx": [_I, _I, _I, ctypes.c_long, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "sig3d_mlp_layer_fwd_compact": [_I, _I, _I, ctypes.c_long, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P],
    "sig3d_mlp_layer_dw_compact": [_I, _I, _I, ctypes.c_long, _P, _P, _P, _P, _P, _I, _P, _P],
    "sig3d_bn_relu_maxpool_compact": [_I, _I, _I, ctypes.c_long, _P, _P, _P, _P, _P, _P, _P],
    "sig3d_bn_relu_bwd_compact": [_I, _I, ctypes.c_long, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I,
                                  _P, _P, _P, _P, _P],
    "sig3d_voxelize": [_I, _I, _P, _P, _I, _I, _P, _P, _I, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                       ctypes.c_long, ctypes.c_long, _P],
    "sig3d_fnv_hash_vec": [ctypes.c_long, _I, _P, _P, _P],
    "sig3d_group_points": [_I, _I, _I, _I, _I, _P, _P, _P, _P],
    "sig3d_group_points_grad": [_I, _I, _I, _I, _I, _P, _P, _P, _P],
    "sig3d_three_nn": [_I, _I, _I, _P, _P, _P, _P, _P],
    "sig3d_three_interpolate": [_I, _I, _I, _I, _P, _P, _P, _P, _P],
    "sig3d_three_interpolate_grad": [_I, _I, _I, _I, _P, _P, _P, _P, _P],
    "sig3d_query_group_fused": [_I, _I, _I, _I, _I, _I, _I, _F, _P, _P, _P, _P, _P, _P],
    "sig3d_query_group_fused_pm": [_I, _I, _I, _I, _I, _I, _I, _I, _F, _P, _P, _P, _P, _P, _P],
    "sig3d_query_group_fused_grad_pm": [_I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P],
    "sig3d_query_group_fused_grad_pm_z": [_I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P],
    "sig3d_transpose_cn": [_I, _I, _I, _P, _P, _P],
    "sig3d_query_group_fused_grad": [_I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P],
    "sig3d_mlp_layer_fwd": [_I, _I, _I, ctypes.c_long, _P, _P, _P, _P, _P, _P, _P, _I, _P],
    "sig3d_bn_finalize": [_I, ctypes.c_double, _F, _F, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "sig3d_bn_relu_maxpool": [_I, _I, _I, _I, _P, _P, _P, _P, _P, _P],
    "sig3d_bn_relu_bwd_top_from_pm": [_I, _I, _I, _I, ctypes.c_long, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P],
    "sig3d_sa_first_layer_fwd": [_I, _I, _I, _I, _I, _I, _I, _I, _F, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P],
    "sig3d_sa_first_layer_dw": [_I, _I, _I, _I, _I, _I, _I, _I, _F, _P, _P, _P, _P, _P, _P, _P, _I, _P],
    "sig3d_pos_mlp_fwd": [_I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "sig3d_pos_mlp_fwd_posed": [_I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "sig3d_pos_mlp_bwd": [_I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P],
    "sig3d_pooled_heads_fwd": [_I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _F, ctypes.c_uint, _P, _P, _P, _P, _P,
                               _P, _P],
    "sig3d_pooled_heads_bwd": [_I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _F, ctypes.c_uint, _P, _P, _P, _P, _P],
    "sig3d_pos_mlp_bwd_z": [_I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P],
    "sig3d_bn_relu_maxpool_pm": [_I, _I, _I, _I, ctypes.c_long, _P, _P, _P, _P, _P, _P, _P, _P],
    "sig3d_channel_stats": [_I, _I, ctypes.c_long, _P, _P, _P, _I, _P],
    "sig3d_bn_relu_apply": [_I, _I, ctypes.c_long, _P, _P, _P, _P, _P],
    "sig3d_bn_relu_bwd": [_I, _I, ctypes.c_long, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P],
    "sig3d_mlp_layer_dw": [_I, _I, _I, ctypes.c_long, _P, _P, _P, _P, _P, _I, _P],
    "sig3d_situational_transform": [_I, _I, _P, _P, _P, _I, _P],
    "sig3d_situational_transform_grad": [_I, _I, _P, _P, _P, _P, _P, _I, _P],
    "sig3d_pos_embed_add": [_I, _I, _I, _I, _I, _F, _P, _P, _P, _P, _P],
    "sig3d_column_sum": [_I, _I, _I, _P, _P, _P],
    "sig3d_bias_gelu": [_I, _I, _I, _P, _P, _P, _P, _P],
    "sig3d_dropout_add_ln_fwd": [_I, _I, _I, _I, _F, ctypes.c_uint, _P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P,
                                 _P, _P],
    "sig3d_dropout_add_ln_bwd": [_I, _I, _I, _I, _F, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "sig3d_dropout_add_ln_fwd_slabs": [_I, _I, _I, _I, _F, ctypes.c_uint, _P, _P, _P, _I, ctypes.c_long, _P, _P, _P, _P, _F,
                                       _P, _P, _P, _P, _P, _P],
    "sig3d_dropout_add_ln_bwd_slabs": [_I, _I, _I, _I, _F, _P, _P, _I, ctypes.c_long, _I, _P, _P, _P, _P, _P, _P, _P, _P,
                                       _P, _P],
    "sig3d_dropout_add_mcan_norm_fwd": [_I, _I, _I, _I, _F, ctypes.c_uint, _P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P,
                                 _P, _P],
    "sig3d_dropout_add_mcan_norm_bwd": [_I, _I, _I, _I, _F, _F, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "sig3d_sqa_loss": [_I, _I, _I, _I, _P, _P, _P, _P, _F, _F, _F, _F, _F, _P, _P, _P, _P],
    "sig3d_sqa_loss_scale": [_I, _I, _P, _P, _P, _P, _P, _P],
    "sig3d_gaussian_target": [_I, _I, _I, _F, _P, _P, _I, _P, _P],
    "sig3d_counter_increment": [_P, _P],
    "sig3d_step_increment": [_P, _P],
    "sig3d_adamw_flat": [ctypes.c_long, _P, _P, _P, _P, _P, _F, _P, _F, _F, _F, _F, _F, _I, _P],
    "sig3d_stream_create_with_cu_mask": [_I, _P, _P],
    "sig3d_stream_destroy": [_P],
    "sig3d_whereami": [_P, _I, _I, _I, _P],
    "sig3d_queue_hold": [_I, _P],
    "sig3d_qformer_embed_fwd": [_I, _I, _I, _I, _I, _P, ctypes.c_long, _P, _P, _I, _P, _I, _I, _P, _P, _F, _F,
                                ctypes.c_uint, _P, _P, _P, _P, _P, _P, _P],
    "sig3d_qformer_embed_bwd": [_I, _I, _I, _I, _I, _I, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P,
                                _P, _P, _P],
    "sig3d_additive_mask": [ctypes.c_long, _P, _I, _P, _P],
    "sig3d_hold": [_P, _I, _I, _I, _I, _I, _P],
    "sig3d_ticket_signal": [_P, _P],
    "sig3d_ticket_wait": [_P, _P, ctypes.c_longlong, _P, _P],
    "sig3d_adamw_table": [_I, _P, _P, _F, _P, _F, _F, _F, _F, _P],
    "sig3d_adamw_table_bounded": [_I, _P, _P, _F, _P, _F, _F, _F, _F, _I, _P],
    "sig3d_gather_table": [_I, _P, _P],
    "sig3d_attention_fwd": [_I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _P, _P, _P, _P, _P, _P, _F,
                            ctypes.c_uint, _P, _I, _P, _P],
    "sig3d_attention_bwd": [_I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                            _F, ctypes.c_uint, _P, _P],
    "sig3d_attention_bwd_z": [_I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                              _F, ctypes.c_uint, _P, _P],
}


class ColumnSumJob(ctypes.Structure):
    """struct sig3d_column_sum_job of include/sig3d_hip.h."""
    _fields_ = [("x", _P), ("out", _P), ("parts", _I), ("rows", _I), ("cols", _I), ("pad", _I)]


COLUMN_SUM_MAX_JOBS = 8
SIGNATURES["sig3d_column_sum_multi"] = [_I, ctypes.POINTER(ColumnSumJob), _P]


def column_sum_multi(device, jobs):
    """jobs: (x2, parts, out) triples -- column sums of contiguous (parts * rows, cols) matrices into (parts, cols) / (cols,)
    outputs, COLUMN_SUM_MAX_JOBS per launch."""
    for i in range(0, len(jobs), COLUMN_SUM_MAX_JOBS):
        chunk = jobs[i:i + COLUMN_SUM_MAX_JOBS]
        arr = (ColumnSumJob * len(chunk))()
        for a, (x2, parts, out) in zip(arr, chunk):
            a.x, a.out, a.parts, a.rows, a.cols = x2.data_ptr(), out.data_ptr(), parts, x2.shape[0] // parts, x2.shape[1]
        with torch.cuda.device(device):
            call("sig3d_column_sum_multi", len(chunk), arr, stream_ptr(device))


class SumSlabsJob(ctypes.Structure):
    """struct sig3d_sum_slabs_job of include/sig3d_hip.h."""
    _fields_ = [("dst", _P), ("slabs", _P), ("n", ctypes.c_long), ("slab_stride", ctypes.c_long), ("nslabs", _I), ("pad", _I)]


SUM_SLABS_MAX_JOBS = 8
SIGNATURES["sig3d_sum_slabs_multi"] = [_I, ctypes.POINTER(SumSlabsJob), _P, _P, _I, _P]


def sum_slabs_multi(device, jobs, convert=None):
    """jobs: (dst, work, n, slab_stride, nslabs): dst[:n] += the nslabs slabs of `work`; SUM_SLABS_MAX_JOBS per launch.
    convert: (float64 source, float32 destination) of equal size, contiguous: converted by the (first) launch as well."""
    jobs = [j for j in jobs if j[4] > 0]
    for i in range(0, max(len(jobs), 1), SUM_SLABS_MAX_JOBS):
        chunk = jobs[i:i + SUM_SLABS_MAX_JOBS]
        arr = (SumSlabsJob * max(len(chunk), 1))()
        for a, (dst, work, n, stride, nslabs) in zip(arr, chunk):
            a.dst, a.slabs, a.n, a.slab_stride, a.nslabs = dst.data_ptr(), work.data_ptr(), n, stride, nslabs
        cv = convert if i == 0 else None
        with torch.cuda.device(device):
            call("sig3d_sum_slabs_multi", len(chunk), arr, ptr(cv[0]) if cv else None, ptr(cv[1]) if cv else None,
                 cv[0].numel() if cv else 0, stream_ptr(device))


class Gemm16Problem(ctypes.Structure):
    """struct sig3d_gemm16_problem of include/sig3d_hip.h (field order and types must match)."""
    _fields_ = [("A", _P), ("lda", _I), ("stride_a", ctypes.c_long),
                ("B", _P), ("ldb", _I), ("stride_b", ctypes.c_long),
                ("C", _P), ("ldc", _I), ("stride_c", ctypes.c_long),
                ("C_slabs", _P), ("slab_stride", ctypes.c_long),
                ("bias", _P), ("stride_bias", ctypes.c_long),
                ("addend", _P), ("aux", _P),
                ("bmode", _I), ("batch", _I), ("m", _I), ("n", _I), ("k", _I), ("act", _I), ("splits", _I),
                ("config", _I)]


SIGNATURES["sig3d_gemm16"] = [ctypes.POINTER(Gemm16Problem), _P]
SIGNATURES["sig3d_gemm16_splits"] = [_I, _I, _I, _I, _I, _I, _I]


def gemm16_splits(bmode, batch, m, n, k, act=0, config=0):
    """The split count sig3d_gemm16 is fastest with on MI355X (a RETURN VALUE, not a status)."""
    return int(load().sig3d_gemm16_splits(bmode, batch, m, n, k, act, config))


def gemm16(device, **kw):
    """C = A B (+ bias) (epilogue) (+ addend) through sig3d_gemm16; tensors or raw pointers for the pointer fields."""
    p = Gemm16Problem()
    vals = dict(A=None, lda=0, stride_a=0, B=None, ldb=0, stride_b=0, C=None, ldc=0, stride_c=0, C_slabs=None,
                slab_stride=0, bias=None, stride_bias=0, addend=None, aux=None, bmode=0, batch=1, m=0, n=0, k=0, act=0,
                splits=1, config=0)
    vals.update(kw)
    for name, v in vals.items():
        setattr(p, name, v.data_ptr() if hasattr(v, "data_ptr") else v)
    with torch.cuda.device(device):
        call("sig3d_gemm16", ctypes.byref(p), stream_ptr(device))


class GemmpProblem(ctypes.Structure):
    """struct sig3d_gemmp_problem of include/sig3d_hip.h (field order and types must match)."""
    _fields_ = [("A", _P), ("chunk_a", ctypes.c_long), ("stride_a", ctypes.c_long), ("bytes_a", ctypes.c_long),
                ("B", _P), ("chunk_b", ctypes.c_long), ("stride_b", ctypes.c_long), ("bytes_b", ctypes.c_long),
                ("C", _P), ("ldc", _I), ("stride_c", ctypes.c_long),
                ("C_planes", _P), ("chunk_c", ctypes.c_long), ("stride_cp", ctypes.c_long),
                ("bias", _P), ("stride_bias", ctypes.c_long),
                ("addend", _P), ("aux", _P), ("work", _P), ("counters", _P),
                ("modes", _I), ("batch", _I), ("m", _I), ("n", _I), ("k", _I), ("act", _I), ("splits", _I),
                ("config", _I)]


SIGNATURES["sig3d_gemmp"] = [ctypes.POINTER(GemmpProblem), _P]
SIGNATURES["sig3d_planes_split"] = [_I, _I, _I, _P, _I, ctypes.c_long, _P, ctypes.c_long, ctypes.c_long, _P]


def gemmp(device, **kw):
    """C = A B (+ bias) (epilogue) (+ addend) on chunked bf16 planes through sig3d_gemmp; tensors or raw pointers."""
    p = GemmpProblem()
    vals = dict(A=None, chunk_a=0, stride_a=0, bytes_a=0, B=None, chunk_b=0, stride_b=0, bytes_b=0, C=None, ldc=0,
                stride_c=0, C_planes=None, chunk_c=0, stride_cp=0, bias=None, stride_bias=0, addend=None, aux=None,
                work=None, counters=None, modes=0, batch=1, m=0, n=0, k=0, act=0, splits=1, config=0)
    vals.update(kw)
    for name, v in vals.items():
        setattr(p, name, v.data_ptr() if hasattr(v, "data_ptr") else v)
    with torch.cuda.device(device):
        call("sig3d_gemmp", ctypes.byref(p), stream_ptr(device))


def gemmp_work_floats(batch, m, n, splits, config=0):
    f = load().sig3d_gemmp_work_floats
    f.restype = ctypes.c_long
    return int(f(batch, m, n, splits, config))


def planes_split(src, planes, rows=None, chunk_rows=None):
    """src (batch, R, C) or (R, C) f32 (rows contiguous) -> chunked bf16 planes (batch, C / 32, chunk_rows, 96) int16;
    only the first `rows` rows are split."""
    if src.dim() == 2:
        src = src.unsqueeze(0)
    batch, r_all, cols = src.shape
    rows = r_all if rows is None else rows
    chunk_rows = planes.shape[-2] if chunk_rows is None else chunk_rows
    with torch.cuda.device(src.device):
        call("sig3d_planes_split", batch, rows, cols, ptr(src), src.stride(1), src.stride(0), ptr(planes),
             chunk_rows * 96, (cols // 32) * chunk_rows * 96, stream_ptr(src.device))
    return planes


class BqLevel(ctypes.Structure):
    """sig3d_bq_level of include/sig3d_hip.h: one ball-query problem of a multi-level launch."""
    _fields_ = [("n", _I), ("m", _I), ("nsample", _I), ("radius", _F), ("xyz", _P), ("new_xyz", _P), ("idx", _P)]


SIGNATURES["sig3d_ball_query_levels"] = [_I, _I, ctypes.POINTER(BqLevel), _P, ctypes.c_long, _P]
SIGNATURES["sig3d_ball_query_levels_ex"] = [_I, _I, ctypes.POINTER(BqLevel), _P, ctypes.c_long, _I, _P]
SIGNATURES["sig3d_ball_query_levels_stats"] = [_I, _I, ctypes.POINTER(BqLevel), _P, ctypes.c_long, _I, _P, _P]
BQ_CLEAN = 1


def bq_levels(problems):
    """problems: [(xyz (b,n,3), new_xyz (b,m,3), radius, nsample, idx (b,m,nsample))] -> ctypes array."""
    arr = (BqLevel * len(problems))()
    for q, (xyz, new_xyz, radius, nsample, idx) in zip(arr, problems):
        q.n, q.m, q.nsample, q.radius = xyz.shape[1], new_xyz.shape[1], int(nsample), float(radius)
        q.xyz, q.new_xyz, q.idx = xyz.data_ptr(), new_xyz.data_ptr(), idx.data_ptr()
    return arr


class GroupLevel(ctypes.Structure):
    """sig3d_group_level of include/sig3d_hip.h: the grouping of one level inside sig3d_query_group_levels."""
    _fields_ = [("n", _I), ("m", _I), ("c", _I), ("ld", _I), ("nsample", _I), ("point_major", _I), ("use_xyz", _I),
                ("normalize_xyz", _I), ("radius", _F), ("xyz", _P), ("new_xyz", _P), ("features", _P), ("idx", _P),
                ("out", _P)]


SIGNATURES["sig3d_query_group_levels"] = [_I, _I, ctypes.POINTER(GroupLevel), _P]


def group_levels(problems, use_xyz=True, normalize_xyz=True):
    """problems: [(xyz (b,n,3), new_xyz (b,m,3), radius, idx (b,m,ns), features (b,c,n) or point-major (b,n,c), point_major,
    out (b,3+c,m,ns))] -> ctypes array for sig3d_query_group_levels."""
    arr = (GroupLevel * len(problems))()
    for q, (xyz, new_xyz, radius, idx, feat, pm, out) in zip(arr, problems):
        q.n, q.m, q.nsample, q.radius = xyz.shape[1], new_xyz.shape[1], idx.shape[2], float(radius)
        q.c = feat.shape[2] if pm else feat.shape[1]
        q.ld, q.point_major, q.use_xyz, q.normalize_xyz = q.c, int(bool(pm)), int(use_xyz), int(normalize_xyz)
        q.xyz, q.new_xyz, q.features, q.idx, q.out = xyz.data_ptr(), new_xyz.data_ptr(), feat.data_ptr(), idx.data_ptr(), out.data_ptr()
    return arr


def bq_levels_workspace_bytes(batch, arr):
    return int(load().sig3d_ball_query_levels_workspace_bytes(batch, len(arr), arr))


INFO_SYMBOLS = ("sig3d_version", "sig3d_last_error", "sig3d_voxelize_workspace_bytes", "sig3d_fps_blocks_workspace_bytes",
                "sig3d_ball_query_levels_workspace_bytes", "sig3d_pooled_heads_work_floats",
                "sig3d_mlp_layer_dw_stream_work_floats", "sig3d_gemmp_work_floats")

_lib = None


class Sig3dError(RuntimeError):
    """A libsig3d_hip call returned a non-zero status (the reference raises RuntimeError via
    AT_ASSERT for argument errors; kernel failures there exit(-1), cuda_utils.h:30-39)."""


def load():
    """Load the shared library (no GPU needed to load or to resolve symbols)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libsig3d_hip.so not found at %s -- build it with `python -m situation3d_amd.build` "
            "(there is no CPU fallback for the HIP hot path)" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: fail loudly
        fn.argtypes = argtypes
        fn.restype = ctypes.c_int
    lib.sig3d_version.restype = ctypes.c_char_p
    lib.sig3d_last_error.restype = ctypes.c_char_p
    lib.sig3d_fps_blocks_workspace_bytes.argtypes = [_I, _I]
    lib.sig3d_fps_blocks_workspace_bytes.restype = ctypes.c_long
    lib.sig3d_voxelize_workspace_bytes.argtypes = [_I, ctypes.c_long, _I]
    lib.sig3d_voxelize_workspace_bytes.restype = ctypes.c_long
    lib.sig3d_ball_query_levels_workspace_bytes.argtypes = [_I, _I, ctypes.POINTER(BqLevel)]
    lib.sig3d_ball_query_levels_workspace_bytes.restype = ctypes.c_long
    lib.sig3d_pooled_heads_work_floats.argtypes = [_I, _I]
    lib.sig3d_pooled_heads_work_floats.restype = ctypes.c_long
    lib.sig3d_mlp_layer_dw_stream_work_floats.argtypes = [_I, _I, _I, ctypes.c_long]
    lib.sig3d_mlp_layer_dw_stream_work_floats.restype = ctypes.c_long
    _lib = lib
    return lib


def version():
    return load().sig3d_version().decode()


def stream_ptr(device=None):
    """Raw hipStream_t of torch's current stream on `device`."""
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


_timed = None  # None, or {entry-point name: [(start_event, end_event, int-args), ...]}


def enable_timing(names):
    """Bracket every call of the named entry points with HIP events on the launch stream
    (torch's current stream).  bench.py uses this to get per-launch kernel durations live."""
    global _timed
    _timed = {n: [] for n in names} if names else None


def timing_records():
    return _timed


def call(name, *args):
    lib = load()
    rec = _timed.get(name) if _timed is not None else None
    if rec is not None:
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
    status = getattr(lib, name)(*args)
    if rec is not None:
        end.record()
        rec.append((start, end, tuple(a for a in args if isinstance(a, int))))
    if status != 0:
        raise Sig3dError("%s failed: %s" % (name, lib.sig3d_last_error().decode()))


def fps_workspace(b, n, device):
    """Scratch of sig3d_furthest_point_sampling_blocks for (b, n): an uninitialised byte tensor (16-byte aligned by
    the allocator)."""
    need = int(load().sig3d_fps_blocks_workspace_bytes(int(b), int(n)))
    return torch.empty(max(need, 16), dtype=torch.uint8, device=device)


def fps_timeouts(reset=False):
    """Scenes whose cooperative furthest-point sampling gave up waiting for a peer workgroup since the
    library was loaded / the last reset (blocking device read: call between steps).  Non-zero means
    batches were sampled with poisoned indices -- treat as an error."""
    n = ctypes.c_uint(0)
    call("sig3d_fps_timeout_count", ctypes.byref(n), int(bool(reset)))
    return int(n.value)


def require_device(*tensors):
    """The reference asserts "CPU not supported" for host tensors (e.g. ball_query.cpp:27-29)."""
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("CPU not supported")
    dev = None
    for t in tensors:
        if t is None:
            continue
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError("all tensors must be on the same device")
    return dev
