"""BLIP-2 Q-Former (BERT encoder with learned queries + cross-attention) over the gfx950
attention kernels -- host-side mirror of the reference's
3DLLM_BLIP2-base/lavis/models/blip2_models/Qformer.py for the path Blip2T5.forward uses
(blip2_t5.py:121-128):

    Qformer.bert(query_embeds=..., encoder_hidden_states=..., encoder_attention_mask=...,
                 return_dict=True).last_hidden_state

Module tree and parameter names equal the reference's, so a reference checkpoint's
`Qformer.bert.*` keys load with strict=True:
    bert.embeddings.{word_embeddings,position_embeddings,LayerNorm}
    bert.encoder.layer.{i}.attention.{self.{query,key,value},output.{dense,LayerNorm}}
    bert.encoder.layer.{i}.crossattention.(same)              (layers with i % freq == 0)
    bert.encoder.layer.{i}.{intermediate,output,intermediate_query,output_query}
`strip_text_branch()` reproduces what Blip2T5.__init__ removes (blip2_t5.py:63-69).

softmax(QK^T/sqrt(d) + mask)V runs in sig3d_attention_fwd/bwd (exact-f32 MFMA, scores never
materialised); the dense layers are plain library GEMMs (torch / hipBLASLt); LayerNorm, GELU and
the bias + hidden-state dropout + residual + LayerNorm tails are one fused row kernel each way
(csrc/rowops.hip).  Attention-probability dropout (Qformer.py:219) is applied INSIDE the attention
kernels: the keep bit of element (b, head, query, key) is a hash of a device counter (advanced
once per training forward), a per-module call id and the element index, so forward and backward
regenerate the same mask and no (B,H,Nq,Nk) tensor is ever stored.  The random stream differs from
torch's Philox (same distribution, different bits); eval mode is deterministic.
"""
import contextlib
import ctypes
import math
import os
import threading
import weakref
from types import SimpleNamespace

import torch
import torch.nn as nn

from . import _lib, scratch


class QFormerConfig:
    """Field names follow transformers' BertConfig (+ the four Q-Former extras set in
    Blip2Base.init_Qformer, blip2.py:50-60).  Defaults == bert-base-uncased."""

    def __init__(self, **kw):
        self.vocab_size = 30522
        self.hidden_size = 768
        self.num_hidden_layers = 12
        self.num_attention_heads = 12
        self.intermediate_size = 3072
        self.hidden_act = "gelu"
        self.hidden_dropout_prob = 0.1
        self.attention_probs_dropout_prob = 0.1
        self.max_position_embeddings = 512
        self.initializer_range = 0.02
        self.layer_norm_eps = 1e-12
        self.pad_token_id = 0
        self.position_embedding_type = "absolute"
        self.encoder_width = 1408
        self.add_cross_attention = True
        self.cross_attention_freq = 2
        self.query_length = 32
        for k, v in kw.items():
            setattr(self, k, v)


class _LinearFn(torch.autograd.Function):
    """F.linear with the same library GEMMs, but the bias gradient comes from the deterministic
    column-sum kernel (sig3d_column_sum) instead of torch's generic reduce kernel (~12 us per
    416 x 768 input on MI355X, 122 launches per Q-Former forward+backward)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return torch.nn.functional.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gy2 = gy.reshape(-1, gy.shape[-1])
        if not gy2.is_contiguous():
            gy2 = gy2.contiguous()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = gy2.mm(weight).view(x.shape)
        if ctx.needs_input_grad[1]:
            gw = gy2.t().mm(x.reshape(-1, x.shape[-1]))
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = torch.empty(gy2.shape[1], dtype=gy2.dtype, device=gy2.device)
            with torch.cuda.device(gy2.device):
                _lib.call("sig3d_column_sum", 1, gy2.shape[0], gy2.shape[1], _lib.ptr(gy2), _lib.ptr(gb),
                          _lib.stream_ptr(gy2.device))
        return gx, gw, gb


def linear(layer, x):
    """`layer(x)` for an nn.Linear on the GPU hot path."""
    if x.is_cuda and x.dtype == torch.float32:
        return _LinearFn.apply(x, layer.weight, layer.bias)
    return layer(x)


_rng_counters = {}   # device -> uint32 scalar advanced once per training forward (dropout seed)
_call_ids = iter(range(1, 1 << 30))


def _rng_counter(device):
    key = (device.type, device.index)
    if key not in _rng_counters:
        _rng_counters[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return _rng_counters[key]


def advance_dropout_seed(device):
    """One captured-safe kernel per forward pass: fresh dropout masks on every (graph) replay."""
    c = _rng_counter(device)
    with torch.cuda.device(device):
        _lib.call("sig3d_counter_increment", _lib.ptr(c), _lib.stream_ptr(device))


# SIG3D_QF_GEMM=1: the Q-Former's dense products on sig3d_gemm16 (csrc/gemm16_core.h: exact-f32 MFMA, bias / GELU / gelu' /
# residual-gradient epilogues, split reductions whose slabs the LayerNorm tails add while loading): query/key/value,
# attention output, feed-forward up (+ GELU) and down, and the three input-gradient products.
# Correct at every golden and full-size parity test and level with rocBLAS's default picks kernel by kernel
# (tools/micro/gemm16_bench.hip) -- but INSIDE the training step the tuned library is not beaten: all products +0.30 ms
# (round 5, four step objects per arm), and no single product pays either (product by product: -0.007 ... +0.119 ms,
# DESIGN.md section 4i).  Default: the vendor library (rocBLAS / hipBLASLt through torch); this switch is the A/B.
OWN_GEMM = os.environ.get("SIG3D_QF_GEMM", "0") != "0"
OWN_MASK = 127      # every product (tests pick single ones; the per-product environment switch went with its measurements)
OWN_CONFIG = 0      # sig3d_gemm16's own choice among its f32 tilings


# the flush's column sums (bias gradients, LayerNorm-tail folds) as one launch (sig3d_column_sum_multi); False: one per kind
COLSUM_MULTI = True


# Cross-attention over MANY encoder tokens (the 3D-LLM shape: B x 5000 ... 80 000 point tokens of width 1408,
# blip2_t5.py:102-129): the key / value projection of every cross layer, its input gradient and its weight gradient are
# products with 10^4 ... 10^5 rows -- 94 % of the Q-Former's FLOPs.  From SIG3D_QF_BIG_ROWS source rows on
# they run on sig3d_gemmp: the encoder tokens are split into chunked bf16 planes ONCE per forward (all cross layers
# read the same tokens), the stacked [Wk; Wv] per layer; 1.2-2.3 x the library at these shapes (profiles/r05_gemmp.md).
BIG_ROWS = int(os.environ.get("SIG3D_QF_BIG_ROWS", "8192"))     # 0: never


def _planes(t2):
    """Chunked bf16 planes of a contiguous (rows, cols) f32 matrix: (1, cols / 32, rows, 96) int16."""
    rows, cols = t2.shape
    return _lib.planes_split(t2, torch.empty((1, cols // 32, rows, 96), dtype=torch.int16, device=t2.device))


class _EncoderPlanes:
    """The planes of the encoder tokens, shared by the cross layers of ONE BertEncoder.forward: the encoder opens a scope,
    the layers inside it are handed the same tensor OBJECT and split it once; the scope ends with the forward (the
    backward pass reads the planes its nodes saved).  Nothing is remembered across forwards -- a reused encoder buffer
    (a static hipGraph input, a tensor refilled through a raw pointer or a `.data` swap; none of those advance
    `_version`) is split again -- and a cross-attention module called on its own, outside a scope, splits per call."""
    _tls = threading.local()

    @classmethod
    @contextlib.contextmanager
    def scope(cls):
        prev, cls._tls.held = getattr(cls._tls, "held", None), []
        try:
            yield
        finally:
            cls._tls.held = prev

    @classmethod
    def of(cls, enc):
        held = getattr(cls._tls, "held", None)
        if held is not None:
            for t, pl in held:
                if t is enc:
                    return pl
        with torch.no_grad():
            pl = _planes(enc.detach().reshape(-1, enc.shape[-1]).contiguous())
        if held is not None:
            held.append((enc, pl))
        return pl


def _big_source(enc, width_out):
    return (BIG_ROWS > 0 and enc is not None and enc.is_cuda and enc.dtype == torch.float32 and enc.dim() == 3
            and enc.shape[0] * enc.shape[1] >= BIG_ROWS and (enc.shape[0] * enc.shape[1]) % 8 == 0
            and enc.shape[2] % 32 == 0 and width_out % 32 == 0
            and enc.shape[0] * enc.shape[1] * max(enc.shape[2], width_out) * 6 < (1 << 32) - 4096)   # 32-bit buffer offsets


def _gp(planes, rows_cap=None):
    """Keyword pieces of a sig3d_gemmp operand: planes (1, cols / 32, rows, 96)."""
    return planes, planes.shape[2] * 96, planes.numel(), planes.numel() * 2


def _g16(dev, **kw):
    _lib.gemm16(dev, **kw)


def _dense_fwd(x2, w, bias, out=None, act=0, aux=None, split=False):
    """x2 (M, K) or (batch, M, K) contiguous, w (N, K) or (batch, N, K) nn.Linear weights, bias (N) / (batch, N) / None
    -> (out, slabs): out = x w^T (+ bias) (gelu; aux keeps the pre-activation), shaped like x2 with N columns.
    split: the reduction may be split; `slabs` (S - 1, rows, N) are then to be ADDED to out by the consumer."""
    batch = x2.shape[0] if x2.dim() == 3 else 1
    m, k = x2.shape[-2], x2.shape[-1]
    n = w.shape[-2]
    dev = x2.device
    if out is None:
        out = torch.empty(x2.shape[:-1] + (n,), dtype=torch.float32, device=dev)
    splits = _lib.gemm16_splits(0, batch, m, n, k, act, OWN_CONFIG) if split else 1
    slabs = torch.empty((splits - 1, batch * m, n), dtype=torch.float32, device=dev) if splits > 1 else None
    _g16(dev, A=x2, lda=k, stride_a=m * k, B=w, ldb=k, stride_b=n * k, C=out, ldc=n, stride_c=m * n, C_slabs=slabs,
         slab_stride=batch * m * n, bias=bias, stride_bias=n, aux=aux, bmode=0, batch=batch, m=m, n=n, k=k, act=act,
         splits=splits, config=OWN_CONFIG)
    return out, slabs


def _dense_dgrad(dy2, w, out=None, addend=None, act=0, aux=None, split=False):
    """dy2 (M, K) or (batch, M, K) contiguous, w (K, N) or (batch, K, N): the nn.Linear weight whose OUTPUT index is
    reduced over -> (out, slabs): out = dy w (* gelu'(aux)) (+ addend; addend may be out itself)."""
    batch = dy2.shape[0] if dy2.dim() == 3 else 1
    m, k = dy2.shape[-2], dy2.shape[-1]
    n = w.shape[-1]
    dev = dy2.device
    if out is None:
        out = torch.empty(dy2.shape[:-1] + (n,), dtype=torch.float32, device=dev)
    splits = _lib.gemm16_splits(1, batch, m, n, k, act, OWN_CONFIG) if split else 1
    slabs = torch.empty((splits - 1, batch * m, n), dtype=torch.float32, device=dev) if splits > 1 else None
    _g16(dev, A=dy2, lda=k, stride_a=m * k, B=w, ldb=n, stride_b=n * k, C=out, ldc=n, stride_c=m * n, C_slabs=slabs,
         slab_stride=batch * m * n, addend=addend, aux=aux, bmode=1, batch=batch, m=m, n=n, k=k, act=act, splits=splits,
         config=OWN_CONFIG)
    return out, slabs


def _ln_tail_fwd(x2, bias, r2, gamma, beta, p_drop, eps, call_id, part_rows=0, mcan=False, out=None,
                 pass_through=False, x_slabs=None):
    """sig3d_dropout_add_ln_fwd on contiguous (rows, cols) operands -> out, v, stats, mask.
    part_rows > 0: bias / gamma / beta are (parts, cols), one set per block of part_rows rows.
    mcan: the MCAN blocks' normalisation (unbiased std, eps on the std) instead of nn.LayerNorm's.
    x2 may hold fewer rows than the residual r2: the rows beyond are padding of the two-segment layout, the
    kernel writes zeros there (out has r2's row count, v has x2's) -- or, with pass_through, the residual's rows
    (rows the block leaves alone).  `out`: a preallocated (rows, cols) destination."""
    dev = x2.device
    live, cols = x2.shape
    rows = r2.shape[0]
    if out is None:
        out = torch.empty((rows, cols), dtype=torch.float32, device=dev)
    v = torch.empty_like(x2)
    stats = torch.empty((2, rows), dtype=torch.float32, device=dev)
    mask = torch.empty((rows, 64), dtype=torch.int16, device=dev) if p_drop > 0 else None
    with torch.cuda.device(dev):
        if x_slabs is not None:    # x2 is slab 0 of a split dense layer: the others are added on the way in
            assert not mcan and x_slabs.shape[1:] == x2.shape
            _lib.call("sig3d_dropout_add_ln_fwd_slabs", rows, cols, part_rows,
                      -live if (pass_through and live < rows) else live, ctypes.c_float(p_drop),
                      ctypes.c_uint(call_id), _lib.ptr(_rng_counter(dev)), _lib.ptr(x2), _lib.ptr(x_slabs),
                      x_slabs.shape[0], x_slabs.stride(0), _lib.ptr(bias), _lib.ptr(r2), _lib.ptr(gamma), _lib.ptr(beta),
                      ctypes.c_float(eps), _lib.ptr(out), _lib.ptr(v), _lib.ptr(stats[0]), _lib.ptr(stats[1]),
                      _lib.ptr(mask), _lib.stream_ptr(dev))
            return out, v, stats, mask
        _lib.call("sig3d_dropout_add_mcan_norm_fwd" if mcan else "sig3d_dropout_add_ln_fwd", rows, cols, part_rows,
                  -live if (pass_through and live < rows) else live, ctypes.c_float(p_drop),
                  ctypes.c_uint(call_id), _lib.ptr(_rng_counter(dev)), _lib.ptr(x2), _lib.ptr(bias),
                  _lib.ptr(r2), _lib.ptr(gamma), _lib.ptr(beta), ctypes.c_float(eps), _lib.ptr(out),
                  _lib.ptr(v), _lib.ptr(stats[0]), _lib.ptr(stats[1]), _lib.ptr(mask),
                  _lib.stream_ptr(dev))
    return out, v, stats, mask


def _ln_bwd_blocks(rows, part_rows=0):
    """Partial rows sig3d_dropout_add_ln_bwd leaves in its workspace (see include/sig3d_hip.h)."""
    rpw = 8 if rows >= 4096 else (2 if rows >= 2048 else 1)      # csrc/rowops.hip: ln_bwd_rows_per_wave
    if 0 < part_rows < rows:
        while part_rows % (4 * rpw) != 0:
            rpw >>= 1
    return (-(-rows // rpw) + 3) // 4


def _ln_tail_bwd(dy2, v, stats, gamma, mask, p_drop, part_rows=0, mcan_eps=None, dx_out=None, pass_through=False,
                 work_out=None, dy_slabs=None, slab_rows=0):
    """sig3d_dropout_add_ln_bwd -> dx (grad of the GEMM output), dres (grad of the residual),
    dparams = [d gamma | d beta | d bias]  ((parts, 3, cols) when part_rows > 0).
    v may hold fewer rows than dy2 (see _ln_tail_fwd): dx has v's rows, dres has dy2's with zeros beyond."""
    live, cols = v.shape
    rows = dy2.shape[0]
    dx = torch.empty_like(v) if dx_out is None else dx_out
    dres = torch.empty((rows, cols), dtype=torch.float32, device=v.device)
    if pass_through and live < rows:
        live = -live
    shape = (rows // part_rows, 3, cols) if part_rows > 0 else (3, cols)
    if work_out is None:
        dparams = torch.empty(shape, dtype=torch.float32, device=v.device)
        work = torch.empty(((rows + 3) // 4, 3 * cols), dtype=torch.float32, device=v.device)
    else:   # the caller folds the per-workgroup partial sums later (many tails in one launch)
        dparams, work = None, work_out
    tail = (_lib.ptr(dy2), _lib.ptr(v), _lib.ptr(stats[0]), _lib.ptr(stats[1]), _lib.ptr(gamma), _lib.ptr(mask),
            _lib.ptr(dx), _lib.ptr(dres), _lib.ptr(dparams), _lib.ptr(work), _lib.stream_ptr(v.device))
    with torch.cuda.device(v.device):
        if dy_slabs is not None:   # dy2 is slab 0 of a split input-gradient product (rows < slab_rows have more)
            assert mcan_eps is None and dy_slabs.shape[2] == cols
            _lib.call("sig3d_dropout_add_ln_bwd_slabs", rows, cols, part_rows, live, ctypes.c_float(p_drop),
                      _lib.ptr(dy2), _lib.ptr(dy_slabs), dy_slabs.shape[0], dy_slabs.stride(0), int(slab_rows), *tail[1:])
        elif mcan_eps is None:
            _lib.call("sig3d_dropout_add_ln_bwd", rows, cols, part_rows, live, ctypes.c_float(p_drop), *tail)
        else:
            _lib.call("sig3d_dropout_add_mcan_norm_bwd", rows, cols, part_rows, live, ctypes.c_float(p_drop),
                      ctypes.c_float(mcan_eps), *tail)
    return dx, dres, dparams


def _colsum(t2, parts=1, out=None):
    """Column sums of a contiguous (parts*rows, cols) matrix -> (cols,) or (parts, cols)."""
    rows, cols = t2.shape[0] // parts, t2.shape[1]
    o = out if out is not None else \
        torch.empty((parts, cols) if parts > 1 else (cols,), dtype=torch.float32, device=t2.device)
    with torch.cuda.device(t2.device):
        _lib.call("sig3d_column_sum", parts, rows, cols, _lib.ptr(t2), _lib.ptr(o), _lib.stream_ptr(t2.device))
    return o


def _bias_gelu(x2, bias, part_rows, gy=None, out=None):
    """gelu(x + bias) (gy None) or gy * gelu'(x + bias); bias (parts, cols) per block of part_rows rows."""
    if out is None:
        out = torch.empty_like(x2)
    with torch.cuda.device(x2.device):
        _lib.call("sig3d_bias_gelu", x2.shape[0], x2.shape[1], part_rows, _lib.ptr(x2), _lib.ptr(bias),
                  _lib.ptr(gy), _lib.ptr(out), _lib.stream_ptr(x2.device))
    return out


class _DropoutAddLayerNormFn(torch.autograd.Function):
    """out = LayerNorm(dropout(x + bias) + residual): the tail of BertSelfOutput / BertOutput
    (Qformer.py:241-246, 323-328) as one kernel each way (csrc/rowops.hip)."""

    @staticmethod
    def forward(ctx, x, bias, residual, gamma, beta, p_drop, eps, call_id, mcan=False):
        cols = x.shape[-1]
        x2 = x.reshape(-1, cols).contiguous()
        r2 = residual.reshape(-1, cols).contiguous()
        out, v, stats, mask = _ln_tail_fwd(x2, bias, r2, gamma, beta, p_drop, eps, call_id, mcan=mcan)
        ctx.save_for_backward(v, stats, gamma, mask)
        ctx.p_drop = p_drop
        ctx.mcan_eps = eps if mcan else None
        ctx.has_bias = bias is not None
        ctx.shape = x.shape
        return out.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        v, stats, gamma, mask = ctx.saved_tensors
        rows, cols = v.shape
        dx, dres, dparams = _ln_tail_bwd(dy.reshape(rows, cols).contiguous(), v, stats, gamma, mask,
                                         ctx.p_drop, mcan_eps=ctx.mcan_eps)
        return (dx.view(ctx.shape), dparams[2] if ctx.has_bias else None, dres.view(ctx.shape), dparams[0],
                dparams[1], None, None, None, None)


def dense_dropout_add_layer_norm(dense, dropout, layer_norm, hidden_states, input_tensor, call_id):
    """LayerNorm(dropout(dense(hidden_states)) + input_tensor) -- GEMM (library) + one fused kernel."""
    if not (hidden_states.is_cuda and hidden_states.dtype == torch.float32
            and hidden_states.shape[-1] <= 4096 and layer_norm.weight.shape[0] <= 1024):
        return layer_norm(dropout(dense(hidden_states)) + input_tensor)
    x = _LinearFn.apply(hidden_states, dense.weight, None)  # bias is added inside the fused tail
    p = dropout.p if dropout.training else 0.0
    return _DropoutAddLayerNormFn.apply(x, dense.bias, input_tensor, layer_norm.weight, layer_norm.bias,
                                        float(p), float(layer_norm.eps), call_id)


class _AttentionFn(torch.autograd.Function):
    """softmax(q k^T * scale + mask) v on token-major (B, N, H*64) operands."""

    @staticmethod
    def forward(ctx, q, k, v, mask, num_heads, scale, p_drop=0.0, call_id=0):
        dev = _lib.require_device(q, k, v, mask)
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        b, nq, hd = q.shape
        nk = k.shape[1]
        d = hd // num_heads
        out = torch.empty((b, nq, hd), dtype=torch.float32, device=dev)
        lse = torch.empty((b, num_heads, nq), dtype=torch.float32, device=dev)
        if mask is not None:
            mask = mask.contiguous()
        _ks, _kw = _fwd_key_splits(b, num_heads, nq, nk, dev, d)
        with torch.cuda.device(dev):
            _lib.call("sig3d_attention_fwd", b, num_heads, nq, nk, d, nq, nk, 0, 0, 0, 0, hd, hd, hd, ctypes.c_float(scale),
                      _lib.ptr(q), _lib.ptr(k), _lib.ptr(v), _lib.ptr(mask), _lib.ptr(out),
                      _lib.ptr(lse), ctypes.c_float(p_drop), ctypes.c_uint(call_id),
                      _lib.ptr(_rng_counter(dev)), _ks, _lib.ptr(_kw), _lib.stream_ptr(dev))
        ctx.save_for_backward(q, k, v, mask, out, lse)
        ctx.cfg = (num_heads, scale, p_drop, call_id)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        q, k, v, mask, out, lse = ctx.saved_tensors
        num_heads, scale, p_drop, call_id = ctx.cfg
        b, nq, hd = q.shape
        nk = k.shape[1]
        grad_out = grad_out.contiguous()
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        with torch.cuda.device(q.device):
            _lib.call("sig3d_attention_bwd", b, num_heads, nq, nk, hd // num_heads, nq, nk, 0, 0, 0, 0, hd, hd, hd,
                      ctypes.c_float(scale), _lib.ptr(q), _lib.ptr(k), _lib.ptr(v), _lib.ptr(mask),
                      _lib.ptr(out), _lib.ptr(lse), _lib.ptr(grad_out), _lib.ptr(dq), _lib.ptr(dk),
                      _lib.ptr(dv), ctypes.c_float(p_drop), ctypes.c_uint(call_id),
                      _lib.ptr(_rng_counter(q.device)), _lib.stream_ptr(q.device))
        return dq, dk, dv, None, None, None, None, None


def _fwd_key_splits(b, heads, nq, nk, device, d=64):
    """(key_splits, workspace) for sig3d_attention_fwd: split the key range when (batch x heads x query
    tiles) alone cannot fill 256 CUs and there are enough 32-key tiles to share out (3D-LLM shapes)."""
    qtiles = (nq + 31) // 32
    ntiles = (nk + 31) // 32
    wgs = b * heads * qtiles
    splits = min(max(1, 1024 // max(wgs, 1)), ntiles // 16, 64)
    if splits <= 1:
        return 1, None
    work = torch.empty(b * heads * qtiles * 32 * splits * (d + 2), dtype=torch.float32, device=device)
    return splits, work


def _off(t, floats):
    return ctypes.c_void_p(t.data_ptr() + 4 * floats)


def _stacked(ts):
    """[W0; W1; ...] as ONE tensor.  Zero-copy when the pieces sit back to back in memory -- which is
    how optim.FlatAdamW lays the query/key/value weights (and their biases) out -- else torch.cat."""
    first = ts[0]
    ptr = first.data_ptr()
    adjacent = True
    for t in ts:
        if not t.is_contiguous() or t.data_ptr() != ptr or t.shape[1:] != first.shape[1:]:
            adjacent = False
            break
        ptr += 4 * t.numel()
    if adjacent:
        rows = sum(t.shape[0] for t in ts)
        need = first.storage_offset() + rows * (first.numel() // first.shape[0])
        if need <= first.untyped_storage().nbytes() // 4:
            return first.as_strided((rows,) + tuple(first.shape[1:]), first.stride(),
                                    first.storage_offset())
    return torch.cat(list(ts), 0)


class _ProjAttentionFn(torch.autograd.Function):
    """query/key/value projections + attention core of BertSelfAttention.forward
    (Qformer.py:150-232) with the projections FUSED into one library GEMM per source tensor:
      self-attention : [Q|K|V] = hidden @ [Wq;Wk;Wv]^T           (one GEMM instead of three)
      cross-attention: Q = hidden @ Wq^T,  [K|V] = enc @ [Wk;Wv]^T (two instead of three)
    The attention kernels read Q, K, V as column slices of those outputs (row stride 3*H*64 /
    2*H*64) and write dQ, dK, dV into the matching slices of ONE gradient buffer, so the backward
    is again one dX GEMM, one dW GEMM and one column-sum per source -- no cat / split / accumulate
    kernels.  Parameters stay separate tensors (state_dict keys unchanged); the stacked weight is a
    zero-copy view when they are adjacent in memory (_stacked).

    layout=None: hidden is (B, N, C) in plain token order.  layout=(B, N, seg): hidden is the 2-D
    (B*N, C) row matrix in TWO-SEGMENT order ([first `seg` tokens of every batch element | the
    rest], see sig3d_attention_fwd) and so is the result (self-attention only)."""

    @staticmethod
    def forward(ctx, hidden, kv_src, wq, bq, wk, bk, wv, bv, mask, num_heads, p_drop, call_id,
                layout=None, enc_planes=None):
        dev = hidden.device
        if layout is None:
            b, nq, c = hidden.shape
            seg = nq
        else:
            b, nq, seg = layout
            c = hidden.shape[-1]
        hd = wq.shape[0]
        d = hd // num_heads
        scale = 1.0 / math.sqrt(d)
        x2 = hidden.reshape(b * nq, c)
        if kv_src is None:  # self-attention
            w_all = _stacked((wq, wk, wv))
            b_all = _stacked((bq, bk, bv))
            proj = torch.addmm(b_all, x2, w_all.t())            # (B*N, 3*hd)
            qp, kp, vp, ldq, ldk, ldv, nk = _off(proj, 0), _off(proj, hd), _off(proj, 2 * hd), 3 * hd, 3 * hd, 3 * hd, nq
            kvproj, kseg = None, seg
        else:
            assert layout is None, "two-segment layout is for self-attention"
            nk = kv_src.shape[1]
            kseg = nk
            e2 = kv_src.reshape(b * nk, kv_src.shape[2])
            w_all = _stacked((wk, wv))
            b_all = _stacked((bk, bv))
            proj = torch.addmm(bq, x2, wq.t())                  # (B*N, hd)
            if enc_planes is None:
                kvproj = torch.addmm(b_all, e2, w_all.t())      # (B*Nk, 2*hd)
            else:   # many encoder tokens: six bf16 products per f32 product over the planes split once per forward
                w_planes = _planes(w_all)
                kvproj = torch.empty((b * nk, 2 * hd), dtype=torch.float32, device=dev)
                pa, ca, sa_, ba = _gp(enc_planes)
                pb, cb, sb_, bb = _gp(w_planes)
                _lib.gemmp(dev, A=pa, chunk_a=ca, stride_a=sa_, bytes_a=ba, B=pb, chunk_b=cb, stride_b=sb_, bytes_b=bb,
                           C=kvproj, ldc=2 * hd, bias=b_all, modes=0, m=b * nk, n=2 * hd, k=e2.shape[1])
            qp, kp, vp, ldq, ldk, ldv = _off(proj, 0), _off(kvproj, 0), _off(kvproj, hd), hd, 2 * hd, 2 * hd
        out = torch.empty((b * nq, hd), dtype=torch.float32, device=dev)
        lse = torch.empty((b, num_heads, nq), dtype=torch.float32, device=dev)
        if mask is not None:
            mask = mask.contiguous()
        _ks, _kw = _fwd_key_splits(b, num_heads, nq, nk, dev)
        with torch.cuda.device(dev):
            _lib.call("sig3d_attention_fwd", b, num_heads, nq, nk, d, seg, kseg, b * seg, b * kseg, 0, 0,
                      ldq, ldk, ldv, ctypes.c_float(scale), qp, kp, vp, _lib.ptr(mask), _lib.ptr(out), _lib.ptr(lse),
                      ctypes.c_float(p_drop), ctypes.c_uint(call_id), _lib.ptr(_rng_counter(dev)),
                      _ks, _lib.ptr(_kw), _lib.stream_ptr(dev))
        ctx.save_for_backward(hidden, kv_src, w_all, wq, proj, kvproj, mask, out, lse)
        ctx.cfg = (num_heads, scale, p_drop, call_id, hd, nk, b, nq, seg, kseg)
        ctx.big = (enc_planes, w_planes) if (kv_src is not None and enc_planes is not None) else None
        # key / value for the reference's "present_key_value" are views of the projections
        if kv_src is None:
            kview, vview = proj[:, hd:2 * hd], proj[:, 2 * hd:]
        else:
            kview, vview = kvproj[:, :hd], kvproj[:, hd:]
        if layout is None:
            out = out.view(b, nq, hd)
            kview, vview = kview.view(b, nk, hd), vview.view(b, nk, hd)
        kview, vview = kview.detach(), vview.detach()
        ctx.mark_non_differentiable(kview, vview)
        ctx.set_materialize_grads(False)  # no zero-filled gradients for the key / value views
        return out, kview, vview

    @staticmethod
    def backward(ctx, grad_out, _gk, _gv):
        hidden, kv_src, w_all, wq, proj, kvproj, mask, out, lse = ctx.saved_tensors
        num_heads, scale, p_drop, call_id, hd, nk, b, nq, seg, kseg = ctx.cfg
        dev = hidden.device
        c = hidden.shape[-1]
        d = hd // num_heads
        grad_out = grad_out.contiguous()
        x2 = hidden.reshape(b * nq, c)
        self_attn = kv_src is None
        dproj = torch.empty_like(proj)
        if self_attn:
            qp, kp, vp = _off(proj, 0), _off(proj, hd), _off(proj, 2 * hd)
            dqp, dkp, dvp = _off(dproj, 0), _off(dproj, hd), _off(dproj, 2 * hd)
            ldq = ldk = ldv = 3 * hd
        else:
            dkv = torch.empty_like(kvproj)
            qp, kp, vp = _off(proj, 0), _off(kvproj, 0), _off(kvproj, hd)
            dqp, dkp, dvp = _off(dproj, 0), _off(dkv, 0), _off(dkv, hd)
            ldq, ldk, ldv = hd, 2 * hd, 2 * hd
        with torch.cuda.device(dev):
            _lib.call("sig3d_attention_bwd", b, num_heads, nq, nk, d, seg, kseg, b * seg, b * kseg, 0, 0,
                      ldq, ldk, ldv, ctypes.c_float(scale), qp, kp, vp, _lib.ptr(mask), _lib.ptr(out), _lib.ptr(lse),
                      _lib.ptr(grad_out), dqp, dkp, dvp, ctypes.c_float(p_drop), ctypes.c_uint(call_id),
                      _lib.ptr(_rng_counter(dev)), _lib.stream_ptr(dev))

        colsum = _colsum
        if self_attn:
            g_hidden = dproj.mm(w_all).view(hidden.shape)
            gw = dproj.t().mm(x2)          # (3*hd, c)
            gb = colsum(dproj)
            return (g_hidden, None, gw[:hd], gb[:hd], gw[hd:2 * hd], gb[hd:2 * hd], gw[2 * hd:], gb[2 * hd:],
                    None, None, None, None, None, None)
        e2 = kv_src.reshape(b * nk, kv_src.shape[2])
        g_hidden = dproj.mm(wq).view(hidden.shape)
        gwq, gbq = dproj.t().mm(x2), colsum(dproj)
        if ctx.big is None:
            g_enc = dkv.mm(w_all).view(kv_src.shape) if ctx.needs_input_grad[1] else None
            gwkv = dkv.t().mm(e2)
        else:   # both gradients of the key / value projection on sig3d_gemmp; dkv is split once for the two
            enc_planes, w_planes = ctx.big
            c_enc = e2.shape[1]
            pd, cd, sd_, bd = _gp(_planes(dkv))
            g_enc = None
            if ctx.needs_input_grad[1]:     # dX = dkv [Wk; Wv]: the weight's planes read with their rows as reduction index
                g_enc = torch.empty((b * nk, c_enc), dtype=torch.float32, device=dev)
                pb, cb, sb_, bb = _gp(w_planes)
                _lib.gemmp(dev, A=pd, chunk_a=cd, stride_a=sd_, bytes_a=bd, B=pb, chunk_b=cb, stride_b=sb_, bytes_b=bb,
                           C=g_enc, ldc=c_enc, modes=1, m=b * nk, n=c_enc, k=2 * hd)
                g_enc = g_enc.view(kv_src.shape)
            # dW = dkv^T enc: 12 x 11 tiles and a reduction over every token -> the reduction is split
            gwkv = torch.empty((2 * hd, c_enc), dtype=torch.float32, device=dev)
            tiles = -(-2 * hd // 128) * -(-c_enc // 128)
            splits = max(1, min(16, 512 // tiles, (b * nk) // 128))
            work = torch.empty(_lib.gemmp_work_floats(1, 2 * hd, c_enc, splits, 3), dtype=torch.float32, device=dev)
            counters = torch.zeros(tiles, dtype=torch.int32, device=dev)
            pe, ce, se_, be = _gp(enc_planes)
            _lib.gemmp(dev, A=pd, chunk_a=cd, stride_a=sd_, bytes_a=bd, B=pe, chunk_b=ce, stride_b=se_, bytes_b=be,
                       C=gwkv, ldc=c_enc, work=work, counters=counters, modes=2, m=2 * hd, n=c_enc, k=b * nk,
                       splits=splits, config=3)
        gbkv = colsum(dkv)
        return (g_hidden, g_enc, gwq, gbq, gwkv[:hd], gbkv[:hd], gwkv[hd:], gbkv[hd:], None, None, None,
                None, None, None)


class _Runs:
    """A buffer with a leading (layer / cross-layer / row) index that is stored as runs [lo, hi) of that index lying
    apart in memory: the parameters of the layers above and below a storage cut (parameter_adjacency_groups(cut=k)) are
    two kind-major arenas, and a layer-batched gradient buffer that IS a slice of the flat gradients has one run in
    each.  Indexing by a layer, a (layer, ...) tuple or a range inside one run returns plain tensors; a range that
    straddles the cut is an error (flush() never issues one: it cuts its ranges at the boundary)."""

    def __init__(self, parts):
        self.parts = list(parts)      # [(lo, hi, tensor whose dim 0 is hi - lo)]

    def _run(self, a, b):
        for lo, hi, t in self.parts:
            if lo <= a and b <= hi:
                return lo, t
        raise IndexError("range [%d, %d) straddles the storage cut" % (a, b))

    def __getitem__(self, key):
        rest = ()
        if isinstance(key, tuple):
            key, rest = key[0], key[1:]
        if isinstance(key, slice):
            a = key.start or 0
            b = key.stop if key.stop is not None else self.parts[-1][1]
            lo, t = self._run(a, b)
            return t[(slice(a - lo, b - lo),) + rest]
        lo, t = self._run(key, key + 1)
        return t[(key - lo,) + rest]

    def tensors(self):
        return [t for _, _, t in self.parts]

    def empty_like(self):
        return _Runs([(lo, hi, torch.empty_like(t)) for lo, hi, t in self.parts])


class _WeightGradArena:
    """Per-forward storage that lets the weight-gradient products of ALL Q-Former layers run as a handful of
    strided-batched GEMMs at the end of the backward pass instead of ~90 small ones inside it.

    A weight gradient (dW = dY^T X, db = column sums of dY) is off the dependency chain of the backward pass:
    nothing needs it before the optimizer.  At the Q-Former's row counts (416 live rows) every GEMM launch pays
    ~6-8 us of ramp and drain around ~10-20 us of work, so the twelve layers' products of one kind as ONE
    batched launch run at 110-116 TFLOP/s instead of 37-84 (tools/batched_dw_probe.py: -0.37 ms per step),
    and the 58 bias-gradient column sums shrink to six launches.  Mechanics:
      * the block functions write the operands of those products -- layer inputs, attention outputs, GELU
        activations in the forward pass; dY of every dense layer in the backward pass -- into slices of the
        buffers below (same bytes as before, only their addresses are planned);
      * their backward returns, as the gradients of the weights, VIEWS of the result buffers, which autograd
        adopts as `.grad` (parameters must have no gradient yet: checked when the arena is built);
      * `flush()` fills the result buffers: automatically when the last block of the stack has run its
        backward, or explicitly between the pieces of a split backward pass (graph_step.GraphedTrainStep).
    The key / value projections of all cross-attention layers read the same scene tokens (Qformer.py:116-118,
    164-170), so they are ONE GEMM in the forward pass (tokens x [Wk0;Wv0;Wk2;Wv2;...]^T) and their weight /
    input gradients two GEMMs in flush() / the lowest cross-attention block instead of twelve."""

    def __init__(self, layers, batch, tq, tt, part_rows, enc2, num_heads, return_enc_at, grad_store=None,
                 storage_cut=None):
        first = layers[0].attention
        wq = first.self.query.weight
        dev, H, I = wq.device, wq.shape[0], layers[0].intermediate_query.dense.weight.shape[0]
        NL = len(layers)
        P, rows, L, rq = part_rows, 2 * part_rows, part_rows + batch * tt, batch * tq
        self.nl, self.P, self.rows, self.L, self.rq, self.H, self.I = NL, P, rows, L, rq, H, I
        self.cross = [i for i, l in enumerate(layers) if l.has_cross_attention]
        self.cross_ord = {l: j for j, l in enumerate(self.cross)}
        e = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)   # noqa: E731
        # operands written by the forward pass
        self.x_attn, self.att, self.x_ffn, self.act = e(NL, rows, H), e(NL, L, H), e(NL, rows, H), e(NL, rows, I)
        # operands written by the backward pass
        self.dyo_attn, self.dproj, self.dyo_ffn, self.gpre = e(NL, L, H), e(NL, L, 3 * H), e(NL, rows, H), e(NL, rows, I)
        # results.  grad_store (optim.FlatAdamW.flat_grad_run, data-parallel step): the big ones ARE slices of the
        # optimizer's flat gradient buffers when the parameters they belong to lie there in this very order
        # (parameter_adjacency_groups) -- the batched products then write the gradients where the all-reduce
        # and AdamW read them, and nothing is gathered or zeroed for them
        # storage_cut = k: the optimizer keeps the layers below and from k on as two kind-major arenas (trainer.build_optimizer
        # (qf_cut=k)), so a buffer that IS gradient storage has a run per arena, and the stacked key / value weights too
        k = storage_cut if storage_cut and 0 < storage_cut < NL else None
        self.runs = [(0, NL)] if k is None else [(0, k), (k, NL)]
        jc = sum(1 for i in self.cross if k is not None and i < k)
        self.cross_runs = [(0, len(self.cross))] if k is None or jc in (0, len(self.cross)) else [(0, jc), (jc, len(self.cross))]

        def res(per_index, runs, unit, *tail):
            """per_index(i) -> the parameters behind index i; runs over that index; `unit` rows of dim 0 per index"""
            parts = []
            for lo, hi in runs:
                shape = ((hi - lo) * unit,) + tail if unit else (hi - lo,) + tail
                params = [p for i in range(lo, hi) for p in per_index(i)]
                v = grad_store(params) if grad_store is not None else None
                t = v.view(*shape) if v is not None and v.numel() == math.prod(shape) else e(*shape)
                parts.append((lo * (unit or 1), hi * (unit or 1), t))
            return parts[0][2] if len(parts) == 1 else _Runs(parts)

        sa = [l.attention.self for l in layers]
        self.gwqkv = res(lambda i: (sa[i].query.weight, sa[i].key.weight, sa[i].value.weight), self.runs, 0, 3 * H, H)
        self.gbqkv = e(NL, 3 * H)
        self.gwo = res(lambda i: (layers[i].attention.output.dense.weight,), self.runs, 0, H, H)
        self.gw1 = res(lambda i: (layers[i].intermediate_query.dense.weight, layers[i].intermediate.dense.weight),
                       self.runs, 0, 2, I, H)
        self.gb1 = e(NL, 2, I)
        self.gw2 = res(lambda i: (layers[i].output_query.dense.weight, layers[i].output.dense.weight), self.runs, 0, 2, H, I)
        # LayerNorm-tail parameter gradients [d gamma | d beta | d bias]: per-workgroup partial rows of every
        # tail's backward kernel, folded over all layers at once
        self.ln_blocks_ffn, self.ln_blocks_attn = _ln_bwd_blocks(rows, P), _ln_bwd_blocks(rows)
        self.ln_work_ffn, self.ln_ffn = e(NL, self.ln_blocks_ffn, 3 * H), e(NL, 2, 3, H)
        self.ln_work_attn, self.ln_attn = e(NL, self.ln_blocks_attn, 3 * H), e(NL, 3, H)
        nc = len(self.cross)
        self.enc2 = enc2
        if nc:
            nrow_e, cenc = enc2.shape
            self.sa_out, self.att_x = e(nc, rows, H), e(nc, rq, H)
            self.dyo_x = e(nc, rq, H)
            # dQ of the cross-attention blocks: few queries against many keys -> the backward splits the keys over
            # workgroups and accumulates dQ with atomics; zeroed once for all blocks (sig3d_attention_bwd_z)
            self.dq_x = scratch.zeros((nc, rq, H), torch.float32, dev)
            self.kv, self.dkv = e(nrow_e, nc * 2 * H), e(nrow_e, nc * 2 * H)
            xs = [layers[i].crossattention for i in self.cross]
            self.gwq_x = res(lambda j: (xs[j].self.query.weight,), self.cross_runs, 0, H, H)
            self.gbq_x = e(nc, H)
            self.gwo_x = res(lambda j: (xs[j].output.dense.weight,), self.cross_runs, 0, H, H)
            self.gwkv = res(lambda j: (xs[j].self.key.weight, xs[j].self.value.weight), self.cross_runs, 2 * H, cenc)
            self.gbkv = e(nc * 2 * H)
            self.ln_work_x, self.ln_x = e(nc, self.ln_blocks_attn, 3 * H), e(nc, 3, H)
            with torch.no_grad():   # [Wk;Wv] of every cross layer stacked: one projection GEMM for all of them (one per arena)
                self.wkv_runs = []
                for j0, j1 in self.cross_runs:
                    ws = [w for j in range(j0, j1) for w in (xs[j].self.key.weight, xs[j].self.value.weight)]
                    bs = [b_ for j in range(j0, j1) for b_ in (xs[j].self.key.bias, xs[j].self.value.bias)]
                    w_all = _stacked(ws)
                    self.wkv_runs.append((j0, j1, w_all))
                    torch.addmm(_stacked(bs), enc2, w_all.t(), out=self.kv[:, j0 * 2 * H:j1 * 2 * H])
        # cross layers whose backward returns the gradient of the scene tokens, accumulated over the cross layers
        # above them (the lowest one; with a split backward pass also the lowest one of the upper piece)
        self.return_enc_at = sorted(set(return_enc_at) & set(self.cross), reverse=True) if nc else []
        self._enc_done_hi = nc          # cross ordinals >= this have had their dkv folded into a returned g_enc
        self._layers = list(layers)
        # shared: another grad-enabled forward ran while this arena was waiting for its backward pass (two forwards
        # before one backward).  The views handed to autograd may then be accumulated into, cloned or joined by
        # other contributions, so the result buffers were zero-filled (BertEncoder.forward) and flush() ADDS the
        # products to whatever `.grad` has become instead of assuming it owns it.
        self.shared = False
        self._marks = set()
        self._expected = 2 * NL + nc
        self._flushed_hi = NL           # layers >= this are flushed
        # Input-gradient products whose reduction is split hand their slabs from one block's backward to the next
        # block's LayerNorm-tail backward, which adds them while loading (sig3d_dropout_add_ln_bwd_slabs).  The
        # tensor autograd carries is slab 0; the others wait here under its address.  Only between blocks of THIS
        # stack (the consumer is known to look here); whatever is left when the last block is done is an error.
        self.slabs_ok = OWN_GEMM
        self._slabs = {}

    def put_slabs(self, grad, slabs, slab_rows):
        self._slabs[grad.data_ptr()] = (slabs, slab_rows)

    def take_slabs(self, grad):
        return self._slabs.pop(grad.data_ptr(), (None, 0))

    # ---- bookkeeping -----------------------------------------------------------------------------------
    def mark(self, kind, layer):
        key = (kind, layer)
        if key in self._marks:
            raise RuntimeError("deferred weight gradients: a Q-Former block ran its backward twice "
                               "(retain_graph is not supported; set encoder.defer_weight_grads = False)")
        self._marks.add(key)
        if len(self._marks) == self._expected and self._slabs:
            raise RuntimeError("Q-Former backward: the slabs of a split input-gradient product were never added "
                               "(their gradient did not reach the block below unchanged); set SIG3D_QF_GEMM=0")
        if len(self._marks) == self._expected:
            if self.shared:
                # the products are ADDED to `.grad`, which autograd only completes after this backward call has
                # returned (the gradients of this very block are still on their way): run when the pass is over
                torch.autograd.Variable._execution_engine.queue_callback(self.flush)
            else:
                self.flush()

    def _layer_done(self, l):
        return ("attn", l) in self._marks and ("ffn", l) in self._marks and \
            (l not in self.cross_ord or ("cross", l) in self._marks)

    def g_enc(self, layer):
        """Gradient of the scene tokens over the cross layers [layer's ordinal, those already returned)."""
        j0, j1 = self.cross_ord[layer], self._enc_done_hi
        self._enc_done_hi = j0
        H2 = 2 * self.H
        out = None
        for a, b, w in self.wkv_runs:     # the stacked weights of the arenas inside [j0, j1)
            a2, b2 = max(a, j0), min(b, j1)
            if a2 >= b2:
                continue
            d, wv = self.dkv[:, a2 * H2:b2 * H2], w[(a2 - a) * H2:(b2 - a) * H2]
            out = d.mm(wv) if out is None else out.addmm_(d, wv)
        return out

    # ---- which parameter gets which view (mirrors what the block functions return) --------------------------
    _RESULTS = ("gwqkv", "gbqkv", "gwo", "gw1", "gb1", "gw2", "ln_ffn", "ln_attn",
                "gwq_x", "gbq_x", "gwo_x", "gwkv", "gbkv", "ln_x")

    def result_buffers(self):
        out = []
        for k in self._RESULTS:
            t = getattr(self, k, None)
            if t is not None:
                out += t.tensors() if isinstance(t, _Runs) else [t]
        return out

    def param_views(self, lo, hi):
        """[(parameter, view of its deferred gradient)] for layers [lo, hi)."""
        H = self.H
        out = []
        for l in range(lo, hi):
            layer = self._layers[l]
            sa, so = layer.attention.self, layer.attention.output
            gw, gb, ln = self.gwqkv[l], self.gbqkv[l], self.ln_attn[l]
            out += [(sa.query.weight, gw[:H]), (sa.query.bias, gb[:H]), (sa.key.weight, gw[H:2 * H]),
                    (sa.key.bias, gb[H:2 * H]), (sa.value.weight, gw[2 * H:]), (sa.value.bias, gb[2 * H:]),
                    (so.dense.weight, self.gwo[l]), (so.dense.bias, ln[2]), (so.LayerNorm.weight, ln[0]),
                    (so.LayerNorm.bias, ln[1])]
            out += [(layer.intermediate_query.dense.weight, self.gw1[l, 0]), (layer.intermediate_query.dense.bias, self.gb1[l, 0]),
                    (layer.intermediate.dense.weight, self.gw1[l, 1]), (layer.intermediate.dense.bias, self.gb1[l, 1]),
                    (layer.output_query.dense.weight, self.gw2[l, 0]), (layer.output.dense.weight, self.gw2[l, 1])]
            for part, mod in ((0, layer.output_query), (1, layer.output)):
                lf = self.ln_ffn[l, part]
                out += [(mod.dense.bias, lf[2]), (mod.LayerNorm.weight, lf[0]), (mod.LayerNorm.bias, lf[1])]
            if l in self.cross_ord:
                j = self.cross_ord[l]
                xa, xo = layer.crossattention.self, layer.crossattention.output
                r0, lx = j * 2 * H, self.ln_x[j]
                out += [(xa.query.weight, self.gwq_x[j]), (xa.query.bias, self.gbq_x[j]),
                        (xa.key.weight, self.gwkv[r0:r0 + H]), (xa.key.bias, self.gbkv[r0:r0 + H]),
                        (xa.value.weight, self.gwkv[r0 + H:r0 + 2 * H]), (xa.value.bias, self.gbkv[r0 + H:r0 + 2 * H]),
                        (xo.dense.weight, self.gwo_x[j]), (xo.dense.bias, lx[2]), (xo.LayerNorm.weight, lx[0]),
                        (xo.LayerNorm.bias, lx[1])]
        return out

    def _check_adopted(self, lo, hi):
        """The deferred scheme is only right when autograd ADOPTED the (then unfilled) views as `.grad`.  A clone
        (create_graph, tensor hooks, a shared reference) or an in-place accumulation into an older gradient has
        copied / added unfilled memory: nothing can repair that afterwards, so fail loudly.  Host-side pointer
        compares only; inside a hipGraph capture this runs once, at capture time."""
        for prm, view in self.param_views(lo, hi):
            g = prm.grad
            if g is not None and g.data_ptr() != view.data_ptr():
                raise RuntimeError(
                    "deferred Q-Former weight gradients: autograd did not adopt the gradient view of a %s parameter "
                    "(another gradient contribution, create_graph or a tensor hook).  Set SIG3D_QF_DEFER=0 / "
                    "encoder.defer_weight_grads = False for this pattern." % (tuple(prm.shape),))

    def _products_shared(self, lo, hi):
        """flush() of a shared arena: products into temporaries, then ADDED to whatever each `.grad` is by now --
        the adopted view (zero-filled + anything accumulated into it in place), an older gradient the zeros were
        added to, or a clone of the zeros."""
        real = {k: getattr(self, k) for k in self._RESULTS if hasattr(self, k)}
        views = self.param_views(lo, hi)
        try:
            for k, t in real.items():
                setattr(self, k, t.empty_like() if isinstance(t, _Runs) else torch.empty_like(t))
            self._products_cut(lo, hi)
            temps = self.param_views(lo, hi)
        finally:
            for k, t in real.items():
                setattr(self, k, t)
        with torch.no_grad():
            for (prm, view), (_, tmp) in zip(views, temps):
                if prm.grad is None:
                    continue               # this backward pass did not accumulate into the parameter
                prm.grad.add_(tmp)         # the view itself when it was adopted

    # ---- the deferred products -------------------------------------------------------------------------
    def flush(self):
        """Weight / bias gradients of every layer whose blocks have all run their backward and that is not
        flushed yet (a contiguous range below the last flush), batched over that range.
        (Issuing them on a side stream beside the point encoder's backward pass was measured slower -- 9.27 against
        8.99 ms per step: the batched GEMMs fill every CU -- and was removed in round 6.)"""
        hi = self._flushed_hi
        lo = hi
        while lo > 0 and self._layer_done(lo - 1):
            lo -= 1
        if hi - lo <= 0:
            return
        self._flushed_hi = lo
        if self.shared:
            self._products_shared(lo, hi)
            return
        self._check_adopted(lo, hi)
        self._products_cut(lo, hi)

    def _products_cut(self, lo, hi):
        """The products of layers [lo, hi), one batch per storage arena (upper arena first, like the backward pass)."""
        for a, b in reversed(self.runs):
            a, b = max(a, lo), min(b, hi)
            if a < b:
                self._products(a, b)

    @torch.no_grad()
    def _products(self, lo, hi):
        n = hi - lo
        P, H, I, L, rq = self.P, self.H, self.I, self.L, self.rq
        js = [j for j, l in enumerate(self.cross) if lo <= l < hi]
        # feed-forward pair: (query branch, text branch) x n layers
        torch.bmm(self.dyo_ffn[lo:hi].view(2 * n, P, H).transpose(1, 2), self.act[lo:hi].view(2 * n, P, I),
                  out=self.gw2[lo:hi].view(2 * n, H, I))
        torch.bmm(self.gpre[lo:hi].view(2 * n, P, I).transpose(1, 2), self.x_ffn[lo:hi].view(2 * n, P, H),
                  out=self.gw1[lo:hi].view(2 * n, I, H))
        # self-attention
        torch.bmm(self.dyo_attn[lo:hi].transpose(1, 2), self.att[lo:hi], out=self.gwo[lo:hi])
        torch.bmm(self.dproj[lo:hi].transpose(1, 2), self.x_attn[lo:hi, :L], out=self.gwqkv[lo:hi])
        # bias gradients, and the LayerNorm tails' (layer, part) x blocks-per-part partial rows -> [d gamma | d beta |
        # d bias]: ONE launch for all kinds (sig3d_column_sum_multi; seven launches of 4-14 us before)
        bf, ba = self.ln_blocks_ffn, self.ln_blocks_attn
        sums = [(self.gpre[lo:hi].view(2 * n * P, I), 2 * n, self.gb1[lo:hi]),
                (self.dproj[lo:hi].view(n * L, 3 * H), n, self.gbqkv[lo:hi]),
                (self.ln_work_ffn[lo:hi].view(n * bf, 3 * H), 2 * n, self.ln_ffn[lo:hi].view(2 * n, 3 * H)),
                (self.ln_work_attn[lo:hi].view(n * ba, 3 * H), n, self.ln_attn[lo:hi].view(n, 3 * H))]
        # cross-attention layers inside the range
        if js:
            j0, j1 = js[0], js[-1] + 1
            m = j1 - j0
            torch.bmm(self.dyo_x[j0:j1].transpose(1, 2), self.att_x[j0:j1], out=self.gwo_x[j0:j1])
            torch.bmm(self.dq_x[j0:j1].transpose(1, 2), self.sa_out[j0:j1, :rq], out=self.gwq_x[j0:j1])
            sums.append((self.dq_x[j0:j1].view(m * rq, H), m, self.gbq_x[j0:j1]))
            sums.append((self.ln_work_x[j0:j1].view(m * ba, 3 * H), m, self.ln_x[j0:j1].view(m, 3 * H)))
            cols = slice(j0 * 2 * H, j1 * 2 * H)
            torch.mm(self.dkv[:, cols].t(), self.enc2, out=self.gwkv[cols])
            if m == len(self.cross):
                sums.append((self.dkv, 1, self.gbkv))
            else:
                torch.sum(self.dkv[:, cols], dim=0, out=self.gbkv[cols])
        if COLSUM_MULTI:
            _lib.column_sum_multi(self.gb1.device, sums)
        else:
            for x2, parts, out in sums:
                _colsum(x2, parts=parts, out=out)


class _AttentionBlockFn(torch.autograd.Function):
    """BertAttention as ONE autograd node on 2-D row matrices (Qformer.py:249-299):
        LayerNorm(dropout(dense(attention(x, kv))) + x)
    = _ProjAttentionFn + the BertSelfOutput tail.  Besides sparing autograd bookkeeping, the block
    form lets the two gradient paths into x (residual and Q/K/V projections) meet inside a GEMM
    epilogue (addmm with beta = 1) instead of an extra accumulate kernel, and hands no key/value
    side outputs to autograd.
    layout = (B, N, seg, base2, rows): x has `rows` storage rows; tokens [0, seg) of every batch
    element sit in rows [0, B*seg), the others from row base2 on (sig3d_attention_fwd); rows that
    hold no token are kept finite (zeros from the attention kernels, row-wise ops elsewhere) and
    carry zero gradients.  seg == N, rows == B*N: plain (B, N) order.
    Everything between the input and the LayerNorm tail runs on the LIVE rows only -- rows [0, L),
    L = base2 + B*(N - seg): the padding behind the shorter segment (96 of 512 rows at B = 8 with 32
    queries + 20 question tokens) costs no GEMM work; the tail kernels write the zeros the padded output
    rows must hold (sig3d_dropout_add_ln_fwd / _bwd, live_rows).
    pass_rows: cross-attention on the leading B*N rows of a LONGER matrix (the text rows behind them
    pass through the block untouched: Qformer.py:375-402 without split / cat).
    arena (a _WeightGradArena) + layer index: operands of the weight-gradient products go to the arena and
    the products themselves are deferred to arena.flush()."""

    @staticmethod
    def forward(ctx, x, kv_src, wq, bq, wk, bk, wv, bv, wo, bo, gamma, beta, mask, num_heads, p_attn,
                p_hidden, eps, id_attn, id_out, layout, arena=None, li=-1):
        x = x.contiguous()
        dev = x.device
        b, nq, seg, base2, rows = layout
        total = x.shape[0]
        assert total >= rows
        live = min(rows, base2 + b * (nq - seg)) if nq > seg else min(rows, b * seg)
        cross = kv_src is not None
        j = arena.cross_ord[li] if (arena is not None and cross) else -1
        if arena is not None:   # flush() reads this block's input from the arena: a producer that did not write in place
            home = arena.sa_out[j] if cross else arena.x_attn[li]
            if x.data_ptr() != home.data_ptr():
                home.copy_(x)
                x = home
        xl = x[:live]
        hd = wq.shape[0]
        d = hd // num_heads
        scale = 1.0 / math.sqrt(d)
        own = OWN_GEMM
        if not cross:  # self-attention
            w_all = _stacked((wq, wk, wv))
            if own and OWN_MASK & 1:
                proj, _ = _dense_fwd(xl, w_all, _stacked((bq, bk, bv)))   # (L, 3*hd)
            else:
                proj = torch.addmm(_stacked((bq, bk, bv)), xl, w_all.t())
            qp, kp, vp, ldq, ldk, ldv, nk = _off(proj, 0), _off(proj, hd), _off(proj, 2 * hd), 3 * hd, 3 * hd, 3 * hd, nq
            klay = (seg, base2, live)
            kvproj, e2 = None, None
        else:
            nk = kv_src.shape[1]
            klay = (nk, 0, 0)
            proj = torch.addmm(bq, xl, wq.t())                            # (L, hd)
            if arena is None:
                e2 = kv_src.reshape(b * nk, kv_src.shape[2])
                w_all = _stacked((wk, wv))
                kvproj = torch.addmm(_stacked((bk, bv)), e2, w_all.t())   # (B*Nk, 2*hd)
                kp, vp, ldk, ldv = _off(kvproj, 0), _off(kvproj, hd), 2 * hd, 2 * hd
            else:   # projected for all cross layers at once when the arena was built
                w_all, kvproj = None, None
                ldk = ldv = arena.kv.shape[1]
                kp, vp = _off(arena.kv, j * 2 * hd), _off(arena.kv, j * 2 * hd + hd)
            qp, ldq = _off(proj, 0), hd
        if arena is None:
            att = torch.empty((live, hd), dtype=torch.float32, device=dev)
        else:
            att = arena.att_x[j] if cross else arena.att[li]
        lse = torch.empty((b, num_heads, nq), dtype=torch.float32, device=dev)
        if mask is not None:
            mask = mask.contiguous()
        _ks, _kw = _fwd_key_splits(b, num_heads, nq, nk, dev)
        with torch.cuda.device(dev):
            _lib.call("sig3d_attention_fwd", b, num_heads, nq, nk, d, seg, klay[0], base2, klay[1], live, klay[2],
                      ldq, ldk, ldv, ctypes.c_float(scale), qp, kp, vp, _lib.ptr(mask), _lib.ptr(att), _lib.ptr(lse),
                      ctypes.c_float(p_attn), ctypes.c_uint(id_attn), _lib.ptr(_rng_counter(dev)),
                      _ks, _lib.ptr(_kw), _lib.stream_ptr(dev))
        if own and OWN_MASK & 2:   # split reduction: the LayerNorm tail adds the slabs while it loads them
            y, y_slabs = _dense_fwd(att, wo, None, split=True)
        else:
            y, y_slabs = att.mm(wo.t()), None
        out_buf = None
        if arena is not None:   # the block's output is the next block's input: written where flush() will read it
            out_buf = arena.x_ffn[li] if (cross or li not in arena.cross_ord) else arena.sa_out[arena.cross_ord[li]]
        out, v, stats, keep = _ln_tail_fwd(y, bo, x, gamma, beta, p_hidden, eps, id_out, out=out_buf,
                                           pass_through=total > rows, x_slabs=y_slabs)
        ctx.save_for_backward(x, kv_src, w_all, wq, wo, proj, kvproj, mask, att, lse, v, stats, gamma, keep)
        ctx.cfg = (num_heads, scale, p_attn, p_hidden, id_attn, hd, nk, b, nq, seg, base2, live, klay, total > rows)
        ctx.arena, ctx.li = arena, li
        return out

    @staticmethod
    def backward(ctx, dy):
        x, kv_src, w_all, wq, wo, proj, kvproj, mask, att, lse, v, stats, gamma, keep = ctx.saved_tensors
        num_heads, scale, p_attn, p_hidden, id_attn, hd, nk, b, nq, seg, base2, live, klay, passing = ctx.cfg
        arena, li = ctx.arena, ctx.li
        dev = x.device
        d = hd // num_heads
        xl = x[:live]
        self_attn = kv_src is None
        j = arena.cross_ord[li] if (arena is not None and not self_attn) else -1
        dyo_buf = None
        if arena is not None:
            dyo_buf = arena.dyo_attn[li] if self_attn else arena.dyo_x[j]
        work = None
        if arena is not None:
            work = arena.ln_work_attn[li] if self_attn else arena.ln_work_x[j]
        dy = dy.contiguous()
        dy_slabs, slab_rows = arena.take_slabs(dy) if arena is not None else (None, 0)
        dyo, dres, dparams = _ln_tail_bwd(dy, v, stats, gamma, keep, p_hidden, dx_out=dyo_buf,
                                          pass_through=passing, work_out=work, dy_slabs=dy_slabs, slab_rows=slab_rows)
        if arena is not None:
            dparams = arena.ln_attn[li] if self_attn else arena.ln_x[j]
        datt = dyo.mm(wo)
        if arena is None:
            dproj = torch.empty_like(proj)
        else:
            dproj = arena.dproj[li] if self_attn else arena.dq_x[j]
        if self_attn:
            qp, kp, vp = _off(proj, 0), _off(proj, hd), _off(proj, 2 * hd)
            dqp, dkp, dvp = _off(dproj, 0), _off(dproj, hd), _off(dproj, 2 * hd)
            ldq = ldk = ldv = 3 * hd
        elif arena is None:
            dkv = torch.empty_like(kvproj)
            qp, kp, vp = _off(proj, 0), _off(kvproj, 0), _off(kvproj, hd)
            dqp, dkp, dvp = _off(dproj, 0), _off(dkv, 0), _off(dkv, hd)
            ldq, ldk, ldv = hd, 2 * hd, 2 * hd
        else:
            ldq, ldk = hd, arena.kv.shape[1]
            ldv = ldk
            qp, kp, vp = _off(proj, 0), _off(arena.kv, j * 2 * hd), _off(arena.kv, j * 2 * hd + hd)
            dqp, dkp, dvp = _off(dproj, 0), _off(arena.dkv, j * 2 * hd), _off(arena.dkv, j * 2 * hd + hd)
        with torch.cuda.device(dev):
            _lib.call("sig3d_attention_bwd" if (self_attn or arena is None) else "sig3d_attention_bwd_z",
                      b, num_heads, nq, nk, d, seg, klay[0], base2, klay[1], live, klay[2],
                      ldq, ldk, ldv, ctypes.c_float(scale), qp, kp, vp, _lib.ptr(mask), _lib.ptr(att), _lib.ptr(lse),
                      _lib.ptr(datt), dqp, dkp, dvp, ctypes.c_float(p_attn), ctypes.c_uint(id_attn),
                      _lib.ptr(_rng_counter(dev)), _lib.stream_ptr(dev))
        nones = (None,) * 10
        # residual + projection paths meet in the product's epilogue.  With an arena the block BELOW is one of this
        # stack's (layer 0's self-attention sits on the embeddings): the reduction is split and the slabs wait in
        # the arena for that block's LayerNorm-tail backward
        split_ok = OWN_GEMM and OWN_MASK & 64 and arena is not None and arena.slabs_ok and (li > 0 or not self_attn)

        def input_grad(w):
            if split_ok:
                _, slabs = _dense_dgrad(dproj, w, out=dres[:live], addend=dres[:live], split=True)
                if slabs is not None:
                    arena.put_slabs(dres, slabs, live)
            else:
                dres[:live].addmm_(dproj, w)
        if self_attn:
            input_grad(w_all)
            if arena is None:
                gwo = dyo.t().mm(att)
                gw = dproj.t().mm(xl)                  # (3*hd, c)
                gb = _colsum(dproj)
            else:
                gwo, gw, gb = arena.gwo[li], arena.gwqkv[li], arena.gbqkv[li]
                arena.mark("attn", li)
            return (dres, None, gw[:hd], gb[:hd], gw[hd:2 * hd], gb[hd:2 * hd], gw[2 * hd:], gb[2 * hd:],
                    gwo, dparams[2], dparams[0], dparams[1]) + nones
        input_grad(wq)
        if arena is None:
            e2 = kv_src.reshape(b * nk, kv_src.shape[2])
            gwo = dyo.t().mm(att)
            gwq, gbq = dproj.t().mm(xl), _colsum(dproj)
            g_enc = dkv.mm(w_all).view(kv_src.shape) if ctx.needs_input_grad[1] else None
            gwkv, gbkv = dkv.t().mm(e2), _colsum(dkv)
            gwk, gbk, gwv, gbv = gwkv[:hd], gbkv[:hd], gwkv[hd:], gbkv[hd:]
        else:
            gwo, gwq, gbq = arena.gwo_x[j], arena.gwq_x[j], arena.gbq_x[j]
            r0 = j * 2 * hd
            gwk, gwv = arena.gwkv[r0:r0 + hd], arena.gwkv[r0 + hd:r0 + 2 * hd]
            gbk, gbv = arena.gbkv[r0:r0 + hd], arena.gbkv[r0 + hd:r0 + 2 * hd]
            g_enc = None
            if ctx.needs_input_grad[1] and li in arena.return_enc_at:
                g_enc = arena.g_enc(li).view(kv_src.shape)
            arena.mark("cross", li)
        return (dres, g_enc, gwq, gbq, gwk, gbk, gwv, gbv, gwo, dparams[2], dparams[0], dparams[1]) + nones


class _FFNBlockFn(torch.autograd.Function):
    """BertIntermediate + BertOutput as ONE autograd node on a 2-D row matrix (Qformer.py:302-328):
        LayerNorm(dropout(dense2(gelu(dense1(x)))) + x)
    Library GEMMs + erf-GELU + the fused LayerNorm tail; in the backward pass the residual gradient
    enters the dense1 input-gradient GEMM as its beta = 1 addend (no accumulate kernel)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, gamma, beta, p_drop, eps, call_id):
        x = x.contiguous()
        pre = torch.addmm(b1, x, w1.t())
        act = torch.nn.functional.gelu(pre)
        y = act.mm(w2.t())
        out, v, stats, keep = _ln_tail_fwd(y, b2, x, gamma, beta, p_drop, eps, call_id)
        ctx.save_for_backward(x, w1, w2, pre, act, v, stats, gamma, keep)
        ctx.p_drop = p_drop
        return out

    @staticmethod
    def backward(ctx, dy):
        x, w1, w2, pre, act, v, stats, gamma, keep = ctx.saved_tensors
        dyo, dres, dparams = _ln_tail_bwd(dy.contiguous(), v, stats, gamma, keep, ctx.p_drop)
        gw2 = dyo.t().mm(act)
        gpre = torch.ops.aten.gelu_backward(dyo.mm(w2), pre)
        gw1 = gpre.t().mm(x)
        gx = dres.addmm_(gpre, w1)   # in place: torch.addmm would first copy dres into a new result
        return gx, gw1, _colsum(gpre), gw2, dparams[2], dparams[0], dparams[1], None, None, None


def _pair(a, b):
    """Parameters of the query branch and of the text branch as ONE (2, ...) tensor: a zero-copy view
    when they are adjacent in memory (trainer.build_optimizer asks optim.FlatAdamW for that), else a copy."""
    return _stacked((a.unsqueeze(0), b.unsqueeze(0)))


class _FFNPairBlockFn(torch.autograd.Function):
    """The query feed-forward block (intermediate_query / output_query) on rows [0, P) and the text
    feed-forward block (intermediate / output) on rows [P, 2P) of one padded (2P, C) row matrix
    (Qformer.py:396-405) as ONE autograd node and HALF the launches: both branches have the same
    shapes, so every GEMM is one strided-batched GEMM over the (2, ...) parameter pairs, bias + GELU,
    the LayerNorm tail and the bias-gradient column sums take a `part_rows` argument.
    Rows without a token (padding of the shorter part) only ever see finite values and zero
    gradients, so they add exact zeros to the weight gradients.
    arena + layer index: see _WeightGradArena (weight-gradient products deferred and batched over layers)."""

    @staticmethod
    def forward(ctx, x, w1q, b1q, w1t, b1t, w2q, b2q, w2t, b2t, gq, bq, gt, bt, p_drop, eps, call_id,
                arena=None, li=-1):
        x = x.contiguous()
        P = x.shape[0] // 2
        if arena is not None and x.data_ptr() != arena.x_ffn[li].data_ptr():
            arena.x_ffn[li].copy_(x)    # a producer that did not write in place
            x = arena.x_ffn[li]
        w1, b1, w2, b2 = _pair(w1q, w1t), _pair(b1q, b1t), _pair(w2q, w2t), _pair(b2q, b2t)
        gamma, beta = _pair(gq, gt), _pair(bq, bt)
        x3 = x.view(2, P, -1)
        own = OWN_GEMM and bool(OWN_MASK & 4)
        if own:   # bias + GELU in the product's epilogue; `pre` keeps x w1^T + b1 for the backward pass
            inter = w1.shape[1]
            pre = torch.empty((2, P, inter), dtype=torch.float32, device=x.device)
            act = arena.act[li] if arena is not None else torch.empty((2 * P, inter), dtype=torch.float32, device=x.device)
            _dense_fwd(x3, w1, b1, out=act.view(2, P, inter), act=1, aux=pre)
        else:
            pre = torch.bmm(x3, w1.transpose(1, 2))                       # (2, P, I), bias added below
            act = _bias_gelu(pre.view(2 * P, -1), b1, P, out=None if arena is None else arena.act[li])   # (2P, I)
        if OWN_GEMM and OWN_MASK & 8:
            y, y_slabs = _dense_fwd(act.view(2, P, -1), w2, None, split=True)            # (2, P, C) + slabs
        else:
            y, y_slabs = torch.bmm(act.view(2, P, -1), w2.transpose(1, 2)), None         # (2, P, C)
        out_buf = arena.x_attn[li + 1] if (arena is not None and li + 1 < arena.nl) else None
        out, v, stats, keep = _ln_tail_fwd(y.view(2 * P, -1), b2, x, gamma, beta, p_drop, eps, call_id, P, out=out_buf,
                                           x_slabs=y_slabs)
        ctx.save_for_backward(x, w1, b1, w2, pre, act, v, stats, gamma, keep)
        ctx.p_drop, ctx.own = p_drop, own
        ctx.arena, ctx.li = arena, li
        return out

    @staticmethod
    def backward(ctx, dy):
        x, w1, b1, w2, pre, act, v, stats, gamma, keep = ctx.saved_tensors
        arena, li = ctx.arena, ctx.li
        P = x.shape[0] // 2
        dy = dy.contiguous()
        dy_slabs, slab_rows = arena.take_slabs(dy) if arena is not None else (None, 0)
        dyo, dres, dparams = _ln_tail_bwd(dy, v, stats, gamma, keep, ctx.p_drop, P,
                                          dx_out=None if arena is None else arena.dyo_ffn[li],
                                          work_out=None if arena is None else arena.ln_work_ffn[li],
                                          dy_slabs=dy_slabs, slab_rows=slab_rows)
        if arena is not None:
            dparams = arena.ln_ffn[li]
        dyo3 = dyo.view(2, P, -1)
        if ctx.own and OWN_MASK & 16:   # `pre` holds the bias already; gelu' rides in the product's epilogue
            gpre = arena.gpre[li] if arena is not None else torch.empty_like(act)
            gpre3 = gpre.view(2, P, -1)
            _dense_dgrad(dyo3, w2, out=gpre3, act=2, aux=pre)
        else:
            gact = torch.bmm(dyo3, w2)                                     # (2, P, I)
            if ctx.own:      # `pre` holds the bias already
                gpre = torch.ops.aten.gelu_backward(gact.view(2 * P, -1), pre.view(2 * P, -1))
                if arena is not None:
                    arena.gpre[li].copy_(gpre)
                    gpre = arena.gpre[li]
            else:
                gpre = _bias_gelu(pre.view(2 * P, -1), b1, P, gy=gact.view(2 * P, -1),
                                  out=None if arena is None else arena.gpre[li])
            gpre3 = gpre.view(2, P, -1)
        if OWN_GEMM and OWN_MASK & 32 and arena is not None and arena.slabs_ok:
            # residual + dense1 input gradients; the block below (this layer's attention block) adds the slabs
            dres3 = dres.view(2, P, -1)
            _, slabs = _dense_dgrad(gpre3, w1, out=dres3, addend=dres3, split=True)
            if slabs is not None:
                arena.put_slabs(dres, slabs, 2 * P)
            gx = dres
        else:
            gx = dres.view(2, P, -1).baddbmm_(gpre3, w1).view(2 * P, -1)   # residual + dense1 input grads
        if arena is None:
            gw2 = torch.bmm(dyo3.transpose(1, 2), act.view(2, P, -1))  # (2, C, I)
            gb1 = _colsum(gpre, parts=2)                               # (2, I)
            gw1 = torch.bmm(gpre3.transpose(1, 2), x.view(2, P, -1))   # (2, I, C)
        else:
            gw2, gb1, gw1 = arena.gw2[li], arena.gb1[li], arena.gw1[li]
            arena.mark("ffn", li)
        # dparams (2, 3, C): [d gamma | d beta | d bias2] per part
        return (gx, gw1[0], gb1[0], gw1[1], gb1[1], gw2[0], dparams[0, 2], gw2[1], dparams[1, 2],
                dparams[0, 0], dparams[0, 1], dparams[1, 0], dparams[1, 1], None, None, None, None, None)


def fused_attention(q, k, v, additive_mask, num_heads, p_drop=0.0, call_id=0):
    """q (B,Nq,H*64), k/v (B,Nk,H*64), additive_mask (B,Nk) or None -> (B,Nq,H*64).
    p_drop > 0: dropout on the attention probabilities (Qformer.py:219), mask = hash of the device
    dropout counter (advance_dropout_seed), `call_id` and the element index."""
    d = q.shape[-1] // num_heads
    return _AttentionFn.apply(q, k, v, additive_mask, num_heads, 1.0 / math.sqrt(d), float(p_drop),
                              int(call_id))


def _key_mask(mask, batch, nk):
    """The reference passes additive masks shaped (B,1,1,Nk) (Qformer.py:700-732 and
    invert_attention_mask); the kernel wants (B,Nk)."""
    if mask is None:
        return None
    if mask.dim() == 4:
        if mask.shape[1] != 1 or mask.shape[2] != 1:
            raise NotImplementedError("per-query (3-D / causal) attention masks are not on the "
                                      "hot path (Qformer.py:677-712 decoder branch)")
        mask = mask[:, 0, 0, :]
    return mask.expand(batch, nk).to(torch.float32)


FUSED_EMBED = True   # False: BertEmbeddings / masks through torch ops (tests compare the two forms)


class _EmbeddingsFn(torch.autograd.Function):
    """dropout(LayerNorm(cat(query_embeds, word[ids] + pos[off + t]))) as ONE launch each way
    (sig3d_qformer_embed_fwd / _bwd, csrc/qformer_embed.hip) instead of ~22 + 13 torch launches.
    `query`: the (1, Q, C) block the batch shares (shared=True: query_tokens itself, its gradient comes back summed
    over the batch) or a contiguous (B, Q, C) tensor.  seg_rows > 0: output in the two-segment row layout (2P, C).
    sink: ddp.SparseRowExchange -- the word rows' gradients go there as rows, the table gets no dense gradient."""

    @staticmethod
    def forward(ctx, query, shared, b, ids, word, pos, pos_off, gamma, beta, eps, p_drop, call_id, seg_rows, pad_id,
                sink):
        dev = word.device
        q, cols = query.shape[-2], query.shape[-1]
        t = ids.shape[1]
        rows = 2 * seg_rows if seg_rows else b * (q + t)
        ids = ids.contiguous()
        out = torch.empty((rows, cols), dtype=torch.float32, device=dev)
        v = torch.empty((rows, cols), dtype=torch.float32, device=dev)
        stats = torch.empty((2, rows), dtype=torch.float32, device=dev)
        mask = torch.empty((rows, 64), dtype=torch.int16, device=dev) if p_drop > 0 else None
        if sink is not None:
            sink.ids.copy_(ids.reshape(-1))
        with torch.cuda.device(dev):
            _lib.call("sig3d_qformer_embed_fwd", b, q, t, cols, seg_rows, _lib.ptr(query), 0 if shared else q * cols,
                      _lib.ptr(ids), _lib.ptr(word), word.shape[0], _lib.ptr(pos), pos.shape[0], pos_off,
                      _lib.ptr(gamma), _lib.ptr(beta), ctypes.c_float(eps), ctypes.c_float(p_drop),
                      ctypes.c_uint(call_id), _lib.ptr(_rng_counter(dev)), _lib.ptr(out), _lib.ptr(v),
                      _lib.ptr(stats[0]), _lib.ptr(stats[1]), _lib.ptr(mask), _lib.stream_ptr(dev))
        ctx.save_for_backward(ids, v, stats, gamma, mask)
        ctx.cfg = (b, q, t, cols, seg_rows, bool(shared), word.shape, pos.shape[0], pos_off, p_drop, pad_id, sink,
                   tuple(query.shape))
        return out if seg_rows else out.view(b, q + t, cols)

    @staticmethod
    def backward(ctx, dy):
        ids, v, stats, gamma, mask = ctx.saved_tensors
        b, q, t, cols, seg_rows, shared, wshape, pos_rows, pos_off, p_drop, pad_id, sink, qshape = ctx.cfg
        dev = v.device
        dy = dy.contiguous()
        dquery = torch.empty(qshape, dtype=torch.float32, device=dev)
        dpos = torch.empty((pos_rows, cols), dtype=torch.float32, device=dev)
        dparams = torch.empty((2, cols), dtype=torch.float32, device=dev)
        work = torch.empty((q + t, 2 * cols), dtype=torch.float32, device=dev)
        dword = rows_out = None
        if sink is not None:
            rows_out = sink.rows
        else:
            dword = torch.zeros(wshape, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.call("sig3d_qformer_embed_bwd", b, q, t, cols, seg_rows, int(shared), _lib.ptr(ids), wshape[0],
                      pos_rows, pos_off, -1 if pad_id is None else int(pad_id), _lib.ptr(dy), _lib.ptr(v),
                      _lib.ptr(stats[0]), _lib.ptr(stats[1]), _lib.ptr(gamma), _lib.ptr(mask), ctypes.c_float(p_drop),
                      _lib.ptr(dquery), _lib.ptr(dpos), _lib.ptr(dword), _lib.ptr(rows_out), _lib.ptr(dparams),
                      _lib.ptr(work), _lib.stream_ptr(dev))
        return (dquery, None, None, None, dword, dpos, None, dparams[0], dparams[1], None, None, None, None, None,
                None)


def _shared_query_block(query_embeds):
    """query_tokens.expand(B, -1, -1) -> the (1, Q, C) block itself (its gradient is then summed over the batch
    inside the embedding kernel); anything else -> None."""
    base = query_embeds._base if query_embeds._is_view() else None
    if (base is not None and query_embeds.dim() == 3 and query_embeds.stride(0) == 0 and base.dim() == 3
            and base.shape[0] == 1 and base.shape[1:] == query_embeds.shape[1:] and base.is_contiguous()
            and base.data_ptr() == query_embeds.data_ptr() and query_embeds.stride()[1:] == base.stride()[1:]):
        return base
    return None


class BertEmbeddings(nn.Module):
    """Qformer.py:51-98"""

    def __init__(self, config):
        super().__init__()
        self.word_embeddings = nn.Embedding(config.vocab_size, config.hidden_size,
                                            padding_idx=config.pad_token_id)
        self.position_embeddings = nn.Embedding(config.max_position_embeddings, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)
        self.register_buffer("position_ids",
                             torch.arange(config.max_position_embeddings).expand((1, -1)))
        self.position_embedding_type = getattr(config, "position_embedding_type", "absolute")
        self.config = config

    def forward(self, input_ids=None, position_ids=None, query_embeds=None,
                past_key_values_length=0, segmented=0):
        seq_length = input_ids.size()[1] if input_ids is not None else 0
        if position_ids is None:   # a view; the torch path below indexes with it (the reference clones, :80-83)
            position_ids = self.position_ids[:, past_key_values_length: seq_length + past_key_values_length]
        if (FUSED_EMBED and input_ids is not None and query_embeds is not None and query_embeds.is_cuda
                and query_embeds.dtype == torch.float32 and self.position_embedding_type == "absolute"
                and input_ids.dim() == 2 and input_ids.shape[1] > 0 and input_ids.dtype == torch.int64
                and query_embeds.shape[-1] <= 1024
                and past_key_values_length + seq_length <= self.position_embeddings.weight.shape[0]
                and position_ids.shape[1] == seq_length
                and position_ids.data_ptr() == self.position_ids.data_ptr() + 8 * past_key_values_length):
            sink = getattr(self, "row_grad_sink", None) if torch.is_grad_enabled() else None
            if sink is not None and sink.ids.numel() != input_ids.numel():
                raise RuntimeError("word-embedding row exchange was set up for %d token positions per step, "
                                   "this batch has %d" % (sink.ids.numel(), input_ids.numel()))
            if not hasattr(self, "_call_id"):
                self._call_id = next(_call_ids)
            block = _shared_query_block(query_embeds)
            query = block if block is not None else query_embeds.contiguous()
            return _EmbeddingsFn.apply(
                query, block is not None, query_embeds.shape[0], input_ids, self.word_embeddings.weight,
                self.position_embeddings.weight, int(past_key_values_length), self.LayerNorm.weight,
                self.LayerNorm.bias, float(self.LayerNorm.eps), self.dropout.p if self.training else 0.0,
                self._call_id, int(segmented), self.word_embeddings.padding_idx, sink)
        if input_ids is not None:
            sink = getattr(self, "row_grad_sink", None)
            if sink is not None and torch.is_grad_enabled():
                # data parallel: the table's gradient travels as rows (ddp.SparseRowExchange).  The sink is sized
                # for one batch shape; the table's dense gradient has no bucket any more, so a differently sized
                # batch must not fall back silently
                if sink.ids.numel() != input_ids.numel():
                    raise RuntimeError("word-embedding row exchange was set up for %d token positions per step, "
                                       "this batch has %d" % (sink.ids.numel(), input_ids.numel()))
                from .ddp import embedding_rows
                embeddings = embedding_rows(self.word_embeddings.weight, input_ids, sink)
            else:
                embeddings = self.word_embeddings(input_ids)
            if self.position_embedding_type == "absolute":
                embeddings = embeddings + self.position_embeddings(position_ids)
            if query_embeds is not None:
                if segmented:  # (2P, C): [query rows, pad | text rows, pad], see BertLayer.forward_segmented
                    c = embeddings.shape[-1]
                    pieces = []
                    for part in (query_embeds.reshape(-1, c), embeddings.reshape(-1, c)):
                        pieces.append(part)
                        if part.shape[0] < segmented:
                            pieces.append(part.new_zeros(segmented - part.shape[0], c))
                    embeddings = torch.cat(pieces, dim=0)
                else:
                    embeddings = torch.cat((query_embeds, embeddings), dim=1)
        else:
            embeddings = query_embeds
        return self.dropout(self.LayerNorm(embeddings))


class BertSelfAttention(nn.Module):
    """Qformer.py:101-232; forward returns (context_layer, (key_layer, value_layer)) with the
    key/value pair in the reference's (B, H, N, 64) view."""

    def __init__(self, config, is_cross_attention):
        super().__init__()
        self.config = config
        if config.hidden_size % config.num_attention_heads != 0:
            raise ValueError("The hidden size (%d) is not a multiple of the number of attention "
                             "heads (%d)" % (config.hidden_size, config.num_attention_heads))
        self.num_attention_heads = config.num_attention_heads
        self.attention_head_size = int(config.hidden_size / config.num_attention_heads)
        self.all_head_size = self.num_attention_heads * self.attention_head_size
        self.query = nn.Linear(config.hidden_size, self.all_head_size)
        kv_in = config.encoder_width if is_cross_attention else config.hidden_size
        self.key = nn.Linear(kv_in, self.all_head_size)
        self.value = nn.Linear(kv_in, self.all_head_size)
        self.dropout = nn.Dropout(config.attention_probs_dropout_prob)
        self._call_id = next(_call_ids)
        if getattr(config, "position_embedding_type", "absolute") != "absolute":
            raise NotImplementedError("relative position embeddings (Qformer.py:189-205) are not "
                                      "used by BLIP-2 and are not on the hot path")

    def transpose_for_scores(self, x):
        return x.view(*x.size()[:-1], self.num_attention_heads,
                      self.attention_head_size).permute(0, 2, 1, 3)

    def forward(self, hidden_states, attention_mask=None, head_mask=None,
                encoder_hidden_states=None, encoder_attention_mask=None, past_key_value=None,
                output_attentions=False):
        if head_mask is not None or past_key_value is not None or output_attentions:
            raise NotImplementedError("head_mask / past_key_value / output_attentions need the "
                                      "materialised probabilities; not on the hot path")
        is_cross_attention = encoder_hidden_states is not None
        kv_src = encoder_hidden_states if is_cross_attention else hidden_states
        if is_cross_attention:
            attention_mask = encoder_attention_mask
        if hidden_states.is_cuda and hidden_states.dtype == torch.float32:
            # fused projections + attention (one GEMM per source tensor, strided Q/K/V slices)
            nk = kv_src.shape[1]
            mask = _key_mask(attention_mask, hidden_states.shape[0], nk)
            p_drop = self.dropout.p if self.training else 0.0
            enc_planes = None
            if is_cross_attention and _big_source(encoder_hidden_states, 2 * self.all_head_size):
                enc_planes = _EncoderPlanes.of(encoder_hidden_states)
            context_layer, key, value = _ProjAttentionFn.apply(
                hidden_states, encoder_hidden_states if is_cross_attention else None,
                self.query.weight, self.query.bias, self.key.weight, self.key.bias, self.value.weight,
                self.value.bias, mask, self.num_attention_heads, float(p_drop), self._call_id, None, enc_planes)
            return (context_layer, (self.transpose_for_scores(key), self.transpose_for_scores(value)))
        key = linear(self.key, kv_src)
        value = linear(self.value, kv_src)
        query = linear(self.query, hidden_states)
        mask = _key_mask(attention_mask, query.shape[0], key.shape[1])
        p_drop = self.dropout.p if self.training else 0.0
        context_layer = fused_attention(query, key, value, mask, self.num_attention_heads, p_drop,
                                        self._call_id)
        return (context_layer, (self.transpose_for_scores(key), self.transpose_for_scores(value)))


def _ffn_rows(intermediate, output, rows):
    """BertIntermediate + BertOutput on a contiguous (rows, C) matrix as one autograd node."""
    p = output.dropout.p if output.training else 0.0
    return _FFNBlockFn.apply(rows, intermediate.dense.weight, intermediate.dense.bias, output.dense.weight,
                             output.dense.bias, output.LayerNorm.weight, output.LayerNorm.bias, float(p),
                             float(output.LayerNorm.eps), output._call_id)


class BertSelfOutput(nn.Module):
    """Qformer.py:235-246"""

    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)
        self._call_id = next(_call_ids)

    def forward(self, hidden_states, input_tensor):
        return dense_dropout_add_layer_norm(self.dense, self.dropout, self.LayerNorm, hidden_states,
                                            input_tensor, self._call_id)


class BertAttention(nn.Module):
    """Qformer.py:249-299"""

    def __init__(self, config, is_cross_attention=False):
        super().__init__()
        self.self = BertSelfAttention(config, is_cross_attention)
        self.output = BertSelfOutput(config)

    def forward(self, hidden_states, attention_mask=None, head_mask=None,
                encoder_hidden_states=None, encoder_attention_mask=None, past_key_value=None,
                output_attentions=False):
        self_outputs = self.self(hidden_states, attention_mask, head_mask, encoder_hidden_states,
                                 encoder_attention_mask, past_key_value, output_attentions)
        return (self.output(self_outputs[0], hidden_states),) + self_outputs[1:]

    def forward_rows(self, rows, attention_mask, layout, encoder_hidden_states=None,
                     encoder_attention_mask=None, arena=None, li=-1):
        """rows (R, C) in the token order `layout` = (B, N, seg, base2, R') of _AttentionBlockFn ->
        same shape and order.  R > R' (cross-attention on the leading rows of the two-segment matrix): the
        rows beyond R' pass through.  arena / li: deferred weight gradients (_WeightGradArena)."""
        att, outp = self.self, self.output
        batch, n_tokens = layout[0], layout[1]
        if encoder_hidden_states is not None:
            mask = _key_mask(encoder_attention_mask, batch, encoder_hidden_states.shape[1])
        else:
            mask = _key_mask(attention_mask, batch, n_tokens)
        p_attn = att.dropout.p if att.training else 0.0
        p_hidden = outp.dropout.p if outp.training else 0.0
        return _AttentionBlockFn.apply(
            rows, encoder_hidden_states, att.query.weight, att.query.bias, att.key.weight, att.key.bias,
            att.value.weight, att.value.bias, outp.dense.weight, outp.dense.bias, outp.LayerNorm.weight,
            outp.LayerNorm.bias, mask, att.num_attention_heads, float(p_attn), float(p_hidden),
            float(outp.LayerNorm.eps), att._call_id, outp._call_id, layout, arena, li)


class BertIntermediate(nn.Module):
    """Qformer.py:302-314 (erf GELU)"""

    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.intermediate_size)
        if config.hidden_act != "gelu":
            raise NotImplementedError("only hidden_act='gelu' (bert-base) is supported")
        self.intermediate_act_fn = nn.GELU()

    def forward(self, hidden_states):
        return self.intermediate_act_fn(linear(self.dense, hidden_states))


class BertOutput(nn.Module):
    """Qformer.py:317-328"""

    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.intermediate_size, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)
        self._call_id = next(_call_ids)

    def forward(self, hidden_states, input_tensor):
        return dense_dropout_add_layer_norm(self.dense, self.dropout, self.LayerNorm, hidden_states,
                                            input_tensor, self._call_id)


class BertLayer(nn.Module):
    """Qformer.py:331-428: self-attention over all tokens, cross-attention (every
    `cross_attention_freq`-th layer) and the query FFN on the first `query_length` tokens, the
    text FFN on the rest."""

    def __init__(self, config, layer_num):
        super().__init__()
        self.config = config
        self.attention = BertAttention(config)
        self.layer_num = layer_num
        if config.add_cross_attention and layer_num % config.cross_attention_freq == 0:
            self.crossattention = BertAttention(config, is_cross_attention=True)
            self.has_cross_attention = True
        else:
            self.has_cross_attention = False
        self.intermediate = BertIntermediate(config)
        self.output = BertOutput(config)
        self.intermediate_query = BertIntermediate(config)
        self.output_query = BertOutput(config)

    def forward(self, hidden_states, attention_mask=None, head_mask=None,
                encoder_hidden_states=None, encoder_attention_mask=None, past_key_value=None,
                output_attentions=False, query_length=0):
        self_attention_outputs = self.attention(hidden_states, attention_mask, head_mask)
        attention_output = self_attention_outputs[0]
        present_key_value = self_attention_outputs[-1]
        if query_length > 0:
            # one split instead of two slices (Qformer.py:375,396): same values, but its backward
            # is a single cat instead of zero-fill + copy + accumulate per slice
            n_text = attention_output.shape[1] - query_length
            if n_text > 0:
                query_attention_output, text_attention_output = torch.split(
                    attention_output, [query_length, n_text], dim=1)
            else:
                query_attention_output = attention_output
            if self.has_cross_attention:
                assert encoder_hidden_states is not None, \
                    "encoder_hidden_states must be given for cross-attention layers"
                query_attention_output = self.crossattention(
                    query_attention_output, attention_mask, head_mask, encoder_hidden_states,
                    encoder_attention_mask)[0]
            layer_output = self.feed_forward_chunk_query(query_attention_output)
            if n_text > 0:
                layer_output_text = self.feed_forward_chunk(text_attention_output)
                layer_output = torch.cat([layer_output, layer_output_text], dim=1)
        else:
            layer_output = self.feed_forward_chunk(attention_output)
        return (layer_output, present_key_value)

    def forward_segmented(self, rows, attention_mask, encoder_hidden_states, encoder_attention_mask,
                          batch, query_length, text_length, part_rows, arena=None, li=-1):
        """Same computation as forward() on the padded two-segment row matrix (2P, C), P = part_rows:
            rows [0, B*Tq)      query tokens        rows [B*Tq, P)       padding
            rows [P, P + B*Tt)  text tokens         rows [P + B*Tt, 2P)  padding
        The query / text split (Qformer.py:375,396) and the final cat (:402) disappear: self-attention
        maps tokens to rows inside the kernel, cross-attention works on the leading B*Tq rows and lets the
        others pass through its LayerNorm tail, and the two feed-forward branches -- same shapes, different
        weights -- run as ONE strided-batched GEMM chain over the two P-row halves (_FFNPairBlockFn).  The
        (B, N, C) layout pays a strided copy per slice, per residual and per gradient, and twice the GEMM
        launches.  arena / li: deferred, layer-batched weight gradients (_WeightGradArena)."""
        rq = batch * query_length
        attention_output = self.attention.forward_rows(
            rows, attention_mask, (batch, query_length + text_length, query_length, part_rows, 2 * part_rows),
            arena=arena, li=li)
        if self.has_cross_attention:
            assert encoder_hidden_states is not None, \
                "encoder_hidden_states must be given for cross-attention layers"
            attention_output = self.crossattention.forward_rows(
                attention_output, None, (batch, query_length, query_length, 0, rq), encoder_hidden_states,
                encoder_attention_mask, arena=arena, li=li)
        iq, oq, it, ot = self.intermediate_query, self.output_query, self.intermediate, self.output
        p = oq.dropout.p if oq.training else 0.0
        return _FFNPairBlockFn.apply(
            attention_output, iq.dense.weight, iq.dense.bias, it.dense.weight, it.dense.bias,
            oq.dense.weight, oq.dense.bias, ot.dense.weight, ot.dense.bias, oq.LayerNorm.weight,
            oq.LayerNorm.bias, ot.LayerNorm.weight, ot.LayerNorm.bias, float(p), float(oq.LayerNorm.eps),
            oq._call_id, arena, li)

    def feed_forward_chunk(self, attention_output):
        return self.output(self.intermediate(attention_output), attention_output)

    def feed_forward_chunk_query(self, attention_output):
        return self.output_query(self.intermediate_query(attention_output), attention_output)


class _SegmentedOutput:
    """Encoder result kept as the padded two-segment row matrix.  `query_hidden_state` (B, Tq, C) -- what
    BLIP-2 consumes (`last_hidden_state[:, :query_length]`, blip2_t5.py / model.py) -- is a free
    view; the full (B, Tq+Tt, C) `last_hidden_state` is assembled on first access."""

    def __init__(self, rows, batch, tq, tt, part_rows):
        self.rows, self._shape, self._part_rows = rows, (batch, tq, tt), part_rows
        self._full = None
        self.past_key_values = self.hidden_states = self.attentions = self.cross_attentions = None
        self.pooler_output = None

    @property
    def query_hidden_state(self):
        b, tq, _ = self._shape
        return self.rows[:b * tq].view(b, tq, -1)

    @property
    def last_hidden_state(self):
        if self._full is None:
            b, tq, tt = self._shape
            p = self._part_rows
            self._full = torch.cat([self.rows[:b * tq].view(b, tq, -1),
                                    self.rows[p:p + b * tt].view(b, tt, -1)], dim=1)
        return self._full


class BertEncoder(nn.Module):
    """Qformer.py:431-526"""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.layer = nn.ModuleList([BertLayer(config, i) for i in range(config.num_hidden_layers)])
        self.cut_after = None   # set for ONE forward: detach after this many layers, (output, leaf) left in .cut
        self.cut = None
        # weight-gradient products of all layers deferred to the end of the backward pass and batched over the
        # layers (_WeightGradArena).  Needs: gradients enabled, every parameter trainable and WITHOUT a gradient
        # at forward time (autograd must adopt, not accumulate into, the views it is handed), no per-parameter
        # gradient hooks that would fire before flush().  trainer.train_step / graph_step switch it per mode.
        self.defer_weight_grads = os.environ.get("SIG3D_QF_DEFER", "1") != "0"
        self._arena_ref = None

    # The arena of the last grad-enabled forward, held WEAKLY: the autograd graph (every block's ctx) owns it, so an
    # abandoned forward's arena dies with its outputs and never counts as pending.
    @property
    def _arena(self):
        return self._arena_ref() if self._arena_ref is not None else None

    @_arena.setter
    def _arena(self, arena):
        self._arena_ref = weakref.ref(arena) if arena is not None else None

    def _make_arena(self, hidden_states, encoder_hidden_states, batch, tq, tt, part_rows, cut):
        if not (self.defer_weight_grads and torch.is_grad_enabled() and hidden_states.is_cuda):
            return None
        if encoder_hidden_states is not None and encoder_hidden_states.dim() != 3:
            return None
        for p in self.parameters():
            if not p.requires_grad or p.grad is not None or getattr(p, "_post_accumulate_grad_hooks", None) \
                    or getattr(p, "_backward_hooks", None):
                return None
        for layer in self.layer:   # stripped text branch (Blip2T5) or no scene tokens for a cross layer
            if layer.intermediate is None or (layer.has_cross_attention and encoder_hidden_states is None):
                return None
        cross = [i for i, l in enumerate(self.layer) if l.has_cross_attention]
        ret = [cross[0]] if cross else []
        if cut is not None and cross:
            upper = [i for i in cross if i >= cut]
            ret += upper[:1]
        enc2 = None
        if encoder_hidden_states is not None:
            enc2 = encoder_hidden_states.reshape(-1, encoder_hidden_states.shape[2])
        arena = _WeightGradArena(self.layer, batch, tq, tt, part_rows, enc2,
                                 self.layer[0].attention.self.num_attention_heads, ret,
                                 grad_store=getattr(self, "grad_store", None),
                                 storage_cut=getattr(self, "storage_cut", None))
        if cut is not None:
            arena.slabs_ok = False   # the gradient crosses the cut through a leaf's .grad, not from block to block
        return arena

    def flush_weight_grads(self):
        """Fill the deferred weight gradients of every layer whose backward has run (see _WeightGradArena;
        automatic at the end of a whole backward pass, explicit between the pieces of a split one) and make
        the current stream wait for them."""
        arena = self._arena
        if arena is not None:
            arena.flush()
            if arena._flushed_hi == 0:
                self._arena = None   # every gradient is in place (the views keep their storage alive)

    def forward(self, hidden_states, attention_mask=None, head_mask=None,
                encoder_hidden_states=None, encoder_attention_mask=None, past_key_values=None,
                use_cache=None, output_attentions=False, output_hidden_states=False,
                return_dict=True, query_length=0, segments=None):
        with _EncoderPlanes.scope():
            return self._forward(hidden_states, attention_mask, encoder_hidden_states, encoder_attention_mask,
                                 output_attentions, output_hidden_states, return_dict, query_length, segments)

    def _forward(self, hidden_states, attention_mask, encoder_hidden_states, encoder_attention_mask,
                 output_attentions, output_hidden_states, return_dict, query_length, segments):
        if segments is not None:
            # hidden_states is the two-segment row matrix (see BertLayer.forward_segmented)
            batch, tq, tt, part_rows = segments
            cut, self.cut_after, self.cut = self.cut_after, None, None
            pending = self._arena
            if pending is not None and torch.is_grad_enabled() and \
                    (len(pending._marks) < pending._expected or pending._flushed_hi > 0):
                # A grad-enabled forward while the previous one still waits for its backward pass (BLIP-2 stage-1
                # style: several forwards, one backward -- or simply an abandoned forward).  Its views are not
                # handed out yet: zero-fill its result buffers and let its flush ADD (shared mode); this forward
                # runs with immediate weight gradients.
                if not pending.shared:
                    pending.shared = True
                    for t in pending.result_buffers():
                        t.zero_()
                arena = None
            else:
                arena = self._make_arena(hidden_states, encoder_hidden_states, batch, tq, tt, part_rows, cut)
            if arena is not None or pending is None or not pending.shared:
                self._arena = arena   # else: the shared arena stays reachable for an explicit flush_weight_grads()
            for i, layer_module in enumerate(self.layer):
                hidden_states = layer_module.forward_segmented(
                    hidden_states, attention_mask, encoder_hidden_states, encoder_attention_mask,
                    batch, tq, tt, part_rows, arena=arena, li=i)
                if cut is not None and i + 1 == cut:
                    # data-parallel step (graph_step.py): the backward pass is cut here so that the gradient
                    # all-reduce of the layers above overlaps the backward of the layers below
                    leaf = hidden_states.detach().requires_grad_(True)
                    self.cut = (hidden_states, leaf)
                    hidden_states = leaf
            return _SegmentedOutput(hidden_states, batch, tq, tt, part_rows)
        all_hidden_states = () if output_hidden_states else None
        for layer_module in self.layer:
            if output_hidden_states:
                all_hidden_states = all_hidden_states + (hidden_states,)
            hidden_states = layer_module(hidden_states, attention_mask, None,
                                         encoder_hidden_states, encoder_attention_mask, None,
                                         output_attentions, query_length)[0]
        if output_hidden_states:
            all_hidden_states = all_hidden_states + (hidden_states,)
        if not return_dict:
            return tuple(v for v in (hidden_states, all_hidden_states) if v is not None)
        return SimpleNamespace(last_hidden_state=hidden_states, past_key_values=None,
                               hidden_states=all_hidden_states, attentions=None,
                               cross_attentions=None)


class BertModel(nn.Module):
    """Qformer.py:613-869 (encoder mode: is_decoder=False; no pooler, as BLIP-2 builds it)."""

    def __init__(self, config, add_pooling_layer=False):
        super().__init__()
        if add_pooling_layer:
            raise NotImplementedError("BLIP-2 builds the Q-Former without a pooler (Qformer.py:903)")
        self.config = config
        self.embeddings = BertEmbeddings(config)
        self.encoder = BertEncoder(config)
        self.segmented_layout = True  # False: keep (B, N, C) tensors between layers (same values)
        self.apply(self._init_weights)

    def _init_weights(self, module):
        """Qformer.py:600-610"""
        if isinstance(module, (nn.Linear, nn.Embedding)):
            module.weight.data.normal_(mean=0.0, std=self.config.initializer_range)
        elif isinstance(module, nn.LayerNorm):
            module.bias.data.zero_()
            module.weight.data.fill_(1.0)
        if isinstance(module, nn.Linear) and module.bias is not None:
            module.bias.data.zero_()

    @staticmethod
    def _additive(mask_2d, dtype):
        """(1 - mask) * -10000 on a (B,N) 0/1 mask (Qformer.py:729-731, invert_attention_mask)."""
        kind = {torch.float32: 0, torch.int64: 1, torch.int32: 2, torch.uint8: 3, torch.bool: 3}.get(mask_2d.dtype)
        if (FUSED_EMBED and mask_2d.is_cuda and dtype == torch.float32 and kind is not None
                and mask_2d.is_contiguous() and not mask_2d.requires_grad):
            out = torch.empty(mask_2d.shape, dtype=torch.float32, device=mask_2d.device)
            with torch.cuda.device(mask_2d.device):     # one launch instead of three
                _lib.call("sig3d_additive_mask", mask_2d.numel(), _lib.ptr(mask_2d), kind, _lib.ptr(out),
                          _lib.stream_ptr(mask_2d.device))
            return out
        return (1.0 - mask_2d.to(dtype)) * -10000.0

    def forward(self, input_ids=None, attention_mask=None, position_ids=None, head_mask=None,
                query_embeds=None, encoder_hidden_states=None, encoder_attention_mask=None,
                past_key_values=None, use_cache=None, output_attentions=None,
                output_hidden_states=None, return_dict=None, is_decoder=False):
        if is_decoder or past_key_values is not None or head_mask is not None:
            raise NotImplementedError("decoder / cache / head-mask paths are outside the hot path")
        if input_ids is None:
            assert query_embeds is not None, \
                "You have to specify query_embeds when input_ids is None"
        query_length = query_embeds.shape[1] if query_embeds is not None else 0
        ref = query_embeds if query_embeds is not None else encoder_hidden_states
        if self.training and ref is not None and ref.is_cuda:
            advance_dropout_seed(ref.device)  # fused dropout kernels hash (this counter, call id, index)
        # hot path: queries + text on the GPU -> two-segment row layout through the whole stack
        segments = None
        if (input_ids is not None and query_embeds is not None and query_embeds.is_cuda
                and query_embeds.dtype == torch.float32 and input_ids.shape[1] > 0
                and not output_hidden_states and return_dict is not False and self.segmented_layout):
            b_, tt_ = query_embeds.shape[0], input_ids.shape[1]
            part_rows = (max(b_ * query_length, b_ * tt_) + 7) // 8 * 8
            segments = (b_, query_length, tt_, part_rows)
        embedding_output = self.embeddings(input_ids=input_ids, position_ids=position_ids,
                                           query_embeds=query_embeds,
                                           segmented=segments[3] if segments is not None else 0)
        if segments is not None:
            batch_size, seq_length = segments[0], segments[1] + segments[2]
        else:
            batch_size, seq_length = embedding_output.shape[:2]
        device = embedding_output.device
        if attention_mask is None:
            attention_mask = torch.ones((batch_size, seq_length), device=device)
        if attention_mask.dim() != 2:
            raise NotImplementedError("only (B, N) padding masks are on the hot path")
        extended_attention_mask = self._additive(attention_mask, embedding_output.dtype)
        encoder_extended_attention_mask = None
        if encoder_hidden_states is not None and not (encoder_attention_mask is None and segments is not None):
            # (no encoder mask on the hot path: the reference builds ones -> an additive mask of zeros, :843-850;
            # the attention kernels take "no mask" instead -- x + 0.0 == x)
            if encoder_attention_mask is None:
                encoder_attention_mask = torch.ones(encoder_hidden_states.shape[:2], device=device)
            encoder_extended_attention_mask = self._additive(encoder_attention_mask,
                                                             embedding_output.dtype)
        enc = self.encoder(embedding_output, attention_mask=extended_attention_mask,
                           encoder_hidden_states=encoder_hidden_states,
                           encoder_attention_mask=encoder_extended_attention_mask,
                           output_hidden_states=bool(output_hidden_states), return_dict=True,
                           query_length=query_length, segments=segments)
        if segments is not None:
            return enc
        if return_dict is False:
            return (enc.last_hidden_state, None)
        return SimpleNamespace(last_hidden_state=enc.last_hidden_state, pooler_output=None,
                               past_key_values=None, hidden_states=enc.hidden_states,
                               attentions=None, cross_attentions=None)


def parameter_adjacency_groups(module, cut=None):
    """Tuples of parameters the hot path wants back to back in memory (in this order).  Two users:
      * [Wq;Wk;Wv] of a layer, the (query branch, text branch) pairs of the feed-forward blocks and
        [Wk0;Wv0;Wk2;Wv2;...] of all cross-attention layers are then zero-copy views (_stacked / _pair);
      * the groups run over ALL layers of a kind, in layer order, so that the layer-batched result buffers of
        the deferred weight gradients (_WeightGradArena: (layers, 3H, H), (layers, 2, I, H), ...) can BE slices of
        a flat gradient buffer with the same layout (optim.FlatAdamW.flat_grad_run): the data-parallel step then
        neither gathers nor zeroes 600 MB of gradients per step.
    cut = k (data parallel, the backward pass cut after layer k so that the upper layers' gradients travel while the
    lower layers compute): every kind is TWO groups, layers [0, k) and [k, NL) -- two kind-major arenas instead of
    one, so that each piece of the backward pass owns one contiguous stretch of the flat gradients (one kind-major
    arena interleaves the pieces kind by kind: dozens of small collectives and AdamW launches, +0.95 ms per step).
    trainer.build_optimizer passes the order on to optim.FlatAdamW and notes the cut on the encoder
    (`storage_cut`); with any other storage the views silently become copies and the arena keeps buffers of its own."""
    layers = [m for m in module.modules() if isinstance(m, BertLayer)]
    if not layers:
        return []
    k = cut if cut and 0 < cut < len(layers) else None
    flat = lambda rows: tuple(p for row in rows for p in row if p is not None)   # noqa: E731
    groups = []
    for part in ([layers] if k is None else [layers[:k], layers[k:]]):
        cross = [l for l in part if l.has_cross_attention]
        groups += [
            flat((l.attention.self.query.weight, l.attention.self.key.weight, l.attention.self.value.weight) for l in part),
            flat((l.attention.self.query.bias, l.attention.self.key.bias, l.attention.self.value.bias) for l in part),
            flat((l.attention.output.dense.weight,) for l in part),
        ]
        if cross:
            groups += [
                flat((l.crossattention.self.key.weight, l.crossattention.self.value.weight) for l in cross),
                flat((l.crossattention.self.key.bias, l.crossattention.self.value.bias) for l in cross),
                flat((l.crossattention.self.query.weight,) for l in cross),
                flat((l.crossattention.self.query.bias,) for l in cross),
                flat((l.crossattention.output.dense.weight,) for l in cross),
            ]
        if all(l.intermediate is not None and l.output is not None for l in layers):
            groups += [
                flat((l.intermediate_query.dense.weight, l.intermediate.dense.weight) for l in part),
                flat((l.intermediate_query.dense.bias, l.intermediate.dense.bias) for l in part),
                flat((l.output_query.dense.weight, l.output.dense.weight) for l in part),
                flat((l.output_query.dense.bias, l.output.dense.bias) for l in part),
                flat((l.output_query.LayerNorm.weight, l.output.LayerNorm.weight) for l in part),
                flat((l.output_query.LayerNorm.bias, l.output.LayerNorm.bias) for l in part),
            ]
    return [g for g in groups if len(g) > 1]


class QFormer(nn.Module):
    """The object Blip2Base.init_Qformer returns as `Qformer` (blip2.py:50-60), minus the LM
    head (`cls`), which Blip2T5 deletes (blip2_t5.py:64)."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.bert = BertModel(config)
        self.cls = None

    def strip_text_branch(self):
        """blip2_t5.py:63-69: no word/position embeddings, no text FFN."""
        self.bert.embeddings.word_embeddings = None
        self.bert.embeddings.position_embeddings = None
        for layer in self.bert.encoder.layer:
            layer.output = None
            layer.intermediate = None
        return self


def init_Qformer(num_query_token, vision_width, cross_attention_freq=2, **overrides):
    """Blip2Base.init_Qformer (blip2.py:50-60) without the hub download: the config is
    bert-base-uncased's, built locally; weights are random-initialised (_init_weights)."""
    config = QFormerConfig(encoder_width=vision_width, add_cross_attention=True,
                           cross_attention_freq=cross_attention_freq,
                           query_length=num_query_token, **overrides)
    qformer = QFormer(config)
    query_tokens = nn.Parameter(torch.zeros(1, num_query_token, config.hidden_size))
    query_tokens.data.normal_(mean=0.0, std=config.initializer_range)
    return qformer, query_tokens
