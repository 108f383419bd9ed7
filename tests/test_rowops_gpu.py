"""Row kernels of the Q-Former dense blocks (csrc/rowops.hip): column_sum (bias gradient) and the
fused bias + dropout + residual + LayerNorm tail of BertSelfOutput / BertOutput
(Qformer.py:241-246, 323-328), forward and backward, against torch."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_column_sum():
    from situation3d_amd import _lib as L
    for rows, cols in [(416, 768), (1, 5), (1000, 3072), (37, 130)]:
        x = torch.randn(rows, cols, device=DEV)
        out = torch.empty(cols, device=DEV)
        L.call("sig3d_column_sum", 1, rows, cols, L.ptr(x), L.ptr(out), L.stream_ptr())
        torch.testing.assert_close(out, x.sum(0), rtol=1e-5, atol=1e-4)
    x = torch.randn(3, 40, 200, device=DEV)      # parts: (3, 40, 200) -> (3, 200)
    out = torch.empty(3, 200, device=DEV)
    L.call("sig3d_column_sum", 3, 40, 200, L.ptr(x), L.ptr(out), L.stream_ptr())
    torch.testing.assert_close(out, x.sum(1), rtol=1e-5, atol=1e-4)


def _fused(x, bias, res, gamma, beta, p, call_id=7):
    from situation3d_amd.qformer import _DropoutAddLayerNormFn
    return _DropoutAddLayerNormFn.apply(x, bias, res, gamma, beta, p, 1e-12, call_id)


@pytest.mark.parametrize("shape", [(8, 52, 768), (3, 5, 128), (2, 7, 1000)])
def test_dropout_add_layer_norm_p0_matches_torch(shape):
    g = torch.Generator().manual_seed(shape[-1])
    c = shape[-1]
    mk = lambda *s: torch.randn(*s, generator=g).to(DEV)
    x, res, bias, gamma, beta, dy = mk(*shape), mk(*shape), mk(c), mk(c) + 1.0, mk(c), mk(*shape)
    args = [t.clone().requires_grad_(True) for t in (x, bias, res, gamma, beta)]
    out = _fused(*args, 0.0)
    out.backward(dy)
    refs = [t.clone().requires_grad_(True) for t in (x, bias, res, gamma, beta)]
    exp = F.layer_norm(refs[0] + refs[1] + refs[2], (c,), refs[3], refs[4], 1e-12)
    exp.backward(dy)
    torch.testing.assert_close(out, exp, rtol=1e-4, atol=1e-4)
    for a, r, name in zip(args, refs, ("x", "bias", "residual", "gamma", "beta")):
        scale = max(1.0, r.grad.abs().max().item())
        torch.testing.assert_close(a.grad, r.grad, rtol=1e-3, atol=1e-4 * scale, msg=lambda m: name + ": " + m)


def test_dropout_add_layer_norm_with_dropout():
    """p > 0: the mask is a pure function of (device counter, call id, element index); recover it
    and check values + gradients against torch with that same mask; check the drop rate."""
    p, c = 0.1, 768
    g = torch.Generator().manual_seed(0)
    mk = lambda *s: torch.randn(*s, generator=g).to(DEV)
    shape = (8, 52, c)
    ones, zeros = torch.ones(shape, device=DEV), torch.zeros(shape, device=DEV)
    probe = _fused(ones, torch.zeros(c, device=DEV), zeros, torch.ones(c, device=DEV),
                   torch.zeros(c, device=DEV), p)
    # recover the keep mask: rows are LayerNorm-ed, kept entries are the larger of two values per row
    keep = probe > probe.mean(-1, keepdim=True)
    rate = 1.0 - keep.float().mean().item()
    n = keep.numel()
    assert abs(rate - p) < 4 * (p * (1 - p) / n) ** 0.5, rate
    other = _fused(ones, torch.zeros(c, device=DEV), zeros, torch.ones(c, device=DEV),
                   torch.zeros(c, device=DEV), p, call_id=8)
    assert (keep != (other > other.mean(-1, keepdim=True))).float().mean() > 0.05  # per-call streams

    x, res, bias, gamma, beta, dy = mk(*shape), mk(*shape), mk(c), mk(c) + 1.0, mk(c), mk(*shape)
    args = [t.clone().requires_grad_(True) for t in (x, bias, res, gamma, beta)]
    out = _fused(*args, p)
    out.backward(dy)
    refs = [t.clone().requires_grad_(True) for t in (x, bias, res, gamma, beta)]
    dropped = (refs[0] + refs[1]) * keep.float() / (1 - p)
    exp = F.layer_norm(dropped + refs[2], (c,), refs[3], refs[4], 1e-12)
    exp.backward(dy)
    torch.testing.assert_close(out, exp, rtol=1e-4, atol=1e-4)
    for a, r, name in zip(args, refs, ("x", "bias", "residual", "gamma", "beta")):
        scale = max(1.0, r.grad.abs().max().item())
        torch.testing.assert_close(a.grad, r.grad, rtol=1e-3, atol=1e-4 * scale, msg=lambda m: name + ": " + m)


def test_dropout_seed_advances_per_forward():
    from situation3d_amd.qformer import advance_dropout_seed
    c = 256
    ones = torch.ones(4, 10, c, device=DEV)
    args = (ones, torch.zeros(c, device=DEV), torch.zeros_like(ones), torch.ones(c, device=DEV),
            torch.zeros(c, device=DEV), 0.3)
    a = _fused(*args)
    b = _fused(*args)
    assert torch.equal(a, b)  # same counter, same call id: same mask
    advance_dropout_seed(ones.device)
    assert not torch.equal(a, _fused(*args))


def test_bias_gelu_matches_torch():
    """erf-GELU of BertIntermediate (Qformer.py:311-313) with per-part biases, forward and backward."""
    from situation3d_amd.qformer import _bias_gelu
    g = torch.Generator().manual_seed(5)
    rows, cols, part = 48, 256, 16
    x = (torch.randn(rows, cols, generator=g) * 2).to(DEV)
    bias = torch.randn(rows // part, cols, generator=g).to(DEV)
    gy = torch.randn(rows, cols, generator=g).to(DEV)
    u = (x.double() + bias.double().repeat_interleave(part, 0)).requires_grad_(True)
    ref = F.gelu(u)
    ref.backward(gy.double())
    # tolerance: 1e-4 (north star); erff / expf differ from the double reference by ~1e-7
    torch.testing.assert_close(_bias_gelu(x, bias, part).double(), ref.detach(), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(_bias_gelu(x, bias, part, gy=gy).double(), u.grad, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("p_drop", [0.0, 0.1])
def test_ffn_pair_block_equals_two_single_blocks(p_drop):
    """_FFNPairBlockFn on a (2P, C) matrix == the query block on rows [0,P) and the text block on rows
    [P,2P) run separately through torch (same dropout masks: recovered from the kernel's own output)."""
    from situation3d_amd.qformer import _FFNPairBlockFn
    g = torch.Generator().manual_seed(11)
    P, C, I = 24, 128, 256
    mk = lambda *s_: (torch.randn(*s_, generator=g) * 0.3).to(DEV)
    params = [mk(I, C), mk(I), mk(I, C), mk(I), mk(C, I), mk(C), mk(C, I), mk(C), mk(C) + 1, mk(C), mk(C) + 1, mk(C)]
    x, dy = mk(2 * P, C), mk(2 * P, C)
    a = [t.clone().requires_grad_(True) for t in [x] + params]
    out = _FFNPairBlockFn.apply(*a, p_drop, 1e-12, 77)
    out.backward(dy)
    r = [t.clone().double().requires_grad_(True) for t in [x] + params]
    xr, (w1q, b1q, w1t, b1t, w2q, b2q, w2t, b2t, gq, bq, gt, bt) = r[0], r[1:]
    outs = []
    for rows, w1, b1, w2, b2, ga, be in ((slice(0, P), w1q, b1q, w2q, b2q, gq, bq), (slice(P, 2 * P), w1t, b1t, w2t, b2t, gt, bt)):
        h = F.linear(F.gelu(F.linear(xr[rows], w1, b1)), w2, b2)
        if p_drop > 0:  # recover the keep mask: probe the kernel with a constant input (same call id)
            from situation3d_amd.qformer import _ln_tail_fwd
            ones = torch.ones(2 * P, C, device=DEV)
            probe = _ln_tail_fwd(ones, torch.zeros(2, C, device=DEV), torch.zeros_like(ones), torch.ones(2, C, device=DEV),
                                 torch.zeros(2, C, device=DEV), p_drop, 1e-12, 77, P)[1]   # v = dropout(1)
            h = h * (probe[rows] > 0).double() / (1 - p_drop)
        outs.append(F.layer_norm(h + xr[rows], (C,), ga, be, 1e-12))
    ref = torch.cat(outs, 0)
    ref.backward(dy.double())
    torch.testing.assert_close(out.double(), ref.detach(), rtol=1e-4, atol=1e-4)
    for t, rr, name in zip(a, r, ["x", "w1q", "b1q", "w1t", "b1t", "w2q", "b2q", "w2t", "b2t", "gq", "bq", "gt", "bt"]):
        scale = max(1.0, rr.grad.abs().max().item())
        torch.testing.assert_close(t.grad.double(), rr.grad, rtol=1e-3, atol=1e-4 * scale, msg=lambda m: name + ": " + m)


@pytest.mark.parametrize("p_drop", [0.0, 0.1])
@pytest.mark.parametrize("rows,live,part_rows,slab_rows,pass_through", [
    (512, 416, 0, 416, False), (512, 512, 256, 0, False), (512, 256, 0, 256, True), (52, 52, 0, 0, False)])
def test_tails_add_the_slabs_of_a_split_dense_layer(p_drop, rows, live, part_rows, slab_rows, pass_through):
    """sig3d_dropout_add_ln_fwd_slabs / _bwd_slabs == the plain tails on the pre-summed operand (the slabs of
    sig3d_gemm16's split reductions are added while they are loaded), padding and pass-through rows included."""
    from situation3d_amd import qformer as Q
    cols, extra = 768, 4
    g = torch.Generator().manual_seed(rows + live + int(p_drop * 100))
    mk = lambda *s: torch.randn(*s, generator=g).to(DEV)
    parts = rows // part_rows if part_rows else 1
    x0, xs, res = mk(live, cols), mk(extra, live, cols), mk(rows, cols)
    bias, gamma, beta = mk(parts, cols), mk(parts, cols) + 1.0, mk(parts, cols)
    if parts == 1:
        bias, gamma, beta = bias[0], gamma[0], beta[0]
    out_a, v_a, st_a, m_a = Q._ln_tail_fwd(x0 + xs.sum(0), bias, res, gamma, beta, p_drop, 1e-12, 11, part_rows,
                                           pass_through=pass_through)
    out_b, v_b, st_b, m_b = Q._ln_tail_fwd(x0, bias, res, gamma, beta, p_drop, 1e-12, 11, part_rows,
                                           pass_through=pass_through, x_slabs=xs)
    torch.testing.assert_close(out_b, out_a, rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(v_b, v_a, rtol=2e-5, atol=2e-5)
    if m_a is not None:
        assert torch.equal(m_a[:live], m_b[:live])     # rows beyond the live ones have no mask
    # backward: dy = slab 0 + slabs on the rows below slab_rows
    n_slab = slab_rows if slab_rows else rows
    dy0, dys = mk(rows, cols), mk(extra, n_slab, cols)
    dy_sum = dy0.clone()
    dy_sum[:n_slab] += dys.sum(0)
    dx_a, dres_a, dp_a = Q._ln_tail_bwd(dy_sum, v_a, st_a, gamma, m_a, p_drop, part_rows, pass_through=pass_through)
    dx_b, dres_b, dp_b = Q._ln_tail_bwd(dy0, v_a, st_a, gamma, m_a, p_drop, part_rows, pass_through=pass_through,
                                        dy_slabs=dys, slab_rows=slab_rows)
    torch.testing.assert_close(dx_b, dx_a, rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(dres_b, dres_a, rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(dp_b, dp_a, rtol=2e-5, atol=2e-4)


def test_column_sum_multi_matches_the_single_launches():
    """sig3d_column_sum_multi: the Q-Former flush's seven kinds of column sums in one launch -- bit for bit what
    sig3d_column_sum gives job by job (same order of additions), ragged column counts and more jobs than one launch holds."""
    import ctypes
    from situation3d_amd import _lib as L
    g = torch.Generator().manual_seed(3)
    shapes = [(24, 256, 3072), (12, 416, 2304), (24, 4, 2304), (12, 7, 2304), (6, 256, 768), (6, 7, 2304), (1, 2048, 9216),
              (3, 5, 70), (1, 1, 1), (2, 33, 65), (1, 0, 17)]
    jobs, ref = [], []
    for parts, rows, cols in shapes:
        x = torch.randn(parts * rows, cols, generator=g).to(DEV)
        out = torch.full((parts, cols), float("nan"), device=DEV)
        one = torch.empty(parts, cols, device=DEV)
        L.call("sig3d_column_sum", parts, rows, cols, L.ptr(x), L.ptr(one), L.stream_ptr(torch.device(DEV)))
        jobs.append((x, parts, out))
        ref.append(one)
    L.column_sum_multi(torch.device(DEV), [j for j in jobs if j[0].shape[0] > 0 or True])
    for (x, parts, out), one, shp in zip(jobs, ref, shapes):
        assert torch.equal(out, one), shp
        if shp[1] > 0:
            torch.testing.assert_close(out.double(), x.view(parts, shp[1], shp[2]).double().sum(1), rtol=1e-5, atol=1e-4)
