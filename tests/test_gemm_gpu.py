"""sig3d_gemm (csrc/gemm.hip): the exact-f32 MFMA GEMM family behind the Q-Former's nn.Linear layers
(Qformer.py:116-118, 238, 305, 320) against float64 torch, through the C ABI: all three operand-layout
combinations, every tile, split K, batches with strides, bias / GELU / gelu' / accumulate epilogues, the
row sums that give a dW product its bias gradient, ragged and unaligned shapes (scalar load path)."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _gemm(amode, bmode, batch, m, n, k, A, lda, sa, B, ldb, sb, C, ldc, sc, bias=None, sbias=0, act=0, aux=None,
          accumulate=0, rowsum=None, srow=0, tile=0, ksplit=0):
    from situation3d_amd import _lib as L
    L.call("sig3d_gemm", amode, bmode, batch, m, n, k, L.ptr(A), lda, sa, L.ptr(B), ldb, sb, L.ptr(C), ldc, sc,
           L.ptr(bias), sbias, act, L.ptr(aux), accumulate, L.ptr(rowsum), srow, tile, ksplit,
           L.stream_ptr(torch.device(DEV)))


def _close(got, ref, tol=2e-5):
    ref = ref.to(got.dtype)
    err = float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-20))
    assert err < tol, err     # f32 accumulation of <= 3072 products: ~1e-6 relative to the largest entry


SHAPES = [(512, 2304, 768), (256, 768, 3072), (64, 64, 32), (100, 36, 70), (37, 129, 33), (5, 3, 2), (130, 260, 96)]


@pytest.mark.parametrize("m,n,k", SHAPES)
@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4])
def test_forward_product_bias_gelu(m, n, k, tile):
    """y = x W^T + b, gelu(y) with the pre-activation kept (BertIntermediate, Qformer.py:305-313)."""
    g = torch.Generator().manual_seed(m + n + k)
    x = torch.randn(m, k, generator=g).to(DEV)
    w = torch.randn(n, k, generator=g).to(DEV) * 0.1
    b = torch.randn(n, generator=g).to(DEV)
    ref = x.double() @ w.double().t() + b.double()
    out = torch.empty(m, n, device=DEV)
    _gemm(0, 0, 1, m, n, k, x, k, 0, w, k, 0, out, n, 0, bias=b, tile=tile, ksplit=1)
    _close(out, ref)
    pre = torch.empty(m, n, device=DEV)
    _gemm(0, 0, 1, m, n, k, x, k, 0, w, k, 0, out, n, 0, bias=b, act=1, aux=pre, tile=tile)
    _close(pre, ref)
    _close(out, torch.nn.functional.gelu(ref))


@pytest.mark.parametrize("m,n,k", SHAPES)
@pytest.mark.parametrize("ksplit", [0, 1, 2, 5])
def test_input_gradient_product_accumulates_and_splits_k(m, n, k, ksplit):
    """dx = dres + dy W (the residual path's gradient is the beta = 1 addend), K split through atomics."""
    g = torch.Generator().manual_seed(m * 3 + n + k)
    dy = torch.randn(m, k, generator=g).to(DEV)
    w = torch.randn(k, n, generator=g).to(DEV) * 0.1
    dres = torch.randn(m, n, generator=g).to(DEV)
    out = dres.clone()
    _gemm(0, 1, 1, m, n, k, dy, k, 0, w, n, 0, out, n, 0, accumulate=1, ksplit=ksplit)
    _close(out, dres.double() + dy.double() @ w.double())
    # times gelu'(pre): BertIntermediate backward
    pre = torch.randn(m, n, generator=g).to(DEV)
    out2 = torch.empty(m, n, device=DEV)
    _gemm(0, 1, 1, m, n, k, dy, k, 0, w, n, 0, out2, n, 0, act=2, aux=pre)
    p = pre.double().requires_grad_(True)
    torch.nn.functional.gelu(p).sum().backward()
    _close(out2, (dy.double() @ w.double()) * p.grad)


@pytest.mark.parametrize("m,n,k", SHAPES)
@pytest.mark.parametrize("tile", [0, 1, 4])
def test_weight_gradient_product_with_bias_gradient(m, n, k, tile):
    """dW = dy^T x and db = column sums of dy in the same launch (k = rows of dy and x)."""
    g = torch.Generator().manual_seed(m + n * 7 + k)
    dy = torch.randn(k, m, generator=g).to(DEV)
    x = torch.randn(k, n, generator=g).to(DEV)
    dw = torch.empty(m, n, device=DEV)
    db = torch.full((m,), 7.0, device=DEV)
    _gemm(1, 1, 1, m, n, k, dy, m, 0, x, n, 0, dw, n, 0, rowsum=db, tile=tile, ksplit=1)
    _close(dw, dy.double().t() @ x.double())
    _close(db, dy.double().sum(0), tol=1e-5)
    # split K: destinations hold zeros, partial sums meet through atomics
    dw.zero_()
    db.zero_()
    _gemm(1, 1, 1, m, n, k, dy, m, 0, x, n, 0, dw, n, 0, rowsum=db, accumulate=1, tile=tile, ksplit=3)
    _close(dw, dy.double().t() @ x.double())
    _close(db, dy.double().sum(0), tol=1e-5)


def test_batched_strided_operands_and_column_slices():
    """The two feed-forward branches as one batch of 2 (per-branch weights and biases), and operands /
    results that are column slices of wider matrices (row stride > width), like Q / K / V in the fused
    projection output."""
    g = torch.Generator().manual_seed(9)
    P, C, I = 96, 128, 256
    x = torch.randn(2 * P, C, generator=g).to(DEV)
    w = torch.randn(2, I, C, generator=g).to(DEV) * 0.1
    b = torch.randn(2, I, generator=g).to(DEV)
    out = torch.empty(2 * P, I, device=DEV)
    _gemm(0, 0, 2, P, I, C, x, C, P * C, w, C, I * C, out, I, P * I, bias=b, sbias=I)
    ref = torch.stack([x[:P].double() @ w[0].double().t() + b[0].double(), x[P:].double() @ w[1].double().t() + b[1].double()])
    _close(out.view(2, P, I), ref)
    db = torch.empty(2, I, device=DEV)
    dw = torch.empty(2, I, C, device=DEV)
    dy = torch.randn(2 * P, I, generator=g).to(DEV)
    _gemm(1, 1, 2, I, C, P, dy, I, P * I, x, C, P * C, dw, C, I * C, rowsum=db, srow=I)
    _close(dw[1], dy[P:].double().t() @ x[P:].double())
    _close(db, dy.view(2, P, I).double().sum(1), tol=1e-5)
    # slices: A = columns [32, 96) of a 160-wide matrix, C = columns [64, 192) of a 256-wide matrix
    wide = torch.randn(80, 160, generator=g).to(DEV)
    wt = torch.randn(128, 64, generator=g).to(DEV)
    big = torch.zeros(80, 256, device=DEV)
    from situation3d_amd import _lib as L
    L.call("sig3d_gemm", 0, 0, 1, 80, 128, 64, ctypes.c_void_p(wide.data_ptr() + 4 * 32), 160, 0, L.ptr(wt), 64, 0,
           ctypes.c_void_p(big.data_ptr() + 4 * 64), 256, 0, None, 0, 0, None, 0, None, 0, 0, 0,
           L.stream_ptr(torch.device(DEV)))
    _close(big[:, 64:192], wide[:, 32:96].double() @ wt.double().t())
    assert float(big[:, :64].abs().max()) == 0 and float(big[:, 192:].abs().max()) == 0


def test_bad_arguments_are_reported():
    from situation3d_amd._lib import Sig3dError
    x = torch.zeros(4, 4, device=DEV)
    with pytest.raises(Sig3dError):
        _gemm(1, 0, 1, 4, 4, 4, x, 4, 0, x, 4, 0, x, 4, 0)         # layout combination not supported
    with pytest.raises(Sig3dError):
        _gemm(0, 0, 1, 4, 4, 4, x, 4, 0, x, 4, 0, x, 4, 0, act=2)  # gelu' without the pre-activation


def _problem(**kw):
    from situation3d_amd import _lib as L
    p = L.GemmProblem()
    defaults = dict(amode=0, bmode=0, batch=1, m=0, n=0, k=0, m_last=None, k_last=None, A=None, lda=0, stride_a=0, B=None, ldb=0,
                    stride_b=0, C=None, ldc=0, stride_c=0, bias=None, stride_bias=0, act=0, aux=None, accumulate=0,
                    rowsum=None, stride_rowsum=0, tile=0, ksplit=0)
    defaults.update(kw)
    if defaults["m_last"] is None:
        defaults["m_last"] = defaults["m"]
    if defaults["k_last"] is None:
        defaults["k_last"] = defaults["k"]
    for k, v in defaults.items():
        if isinstance(v, torch.Tensor):
            v = v.data_ptr()
        setattr(p, k, v)
    return p


def _group(*problems):
    from situation3d_amd import _lib as L
    arr = (L.GemmProblem * len(problems))(*problems)
    L.call("sig3d_gemm_group", len(problems), arr, L.stream_ptr(torch.device(DEV)))


@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("rows", [(256, 160), (40, 9), (64, 64), (33, 1)])
def test_ragged_batch_of_two_branches(tile, rows):
    """Query branch (m rows) and text branch (m_last rows) of the feed-forward block as a batch of 2 with
    per-branch weights; the rows between m_last and m of the second element are never touched."""
    mq, mt = rows
    C, I = 96, 160
    g = torch.Generator().manual_seed(mq * 7 + mt)
    x = torch.randn(2 * mq, C, generator=g).to(DEV)
    w = torch.randn(2, I, C, generator=g).to(DEV) * 0.1
    b = torch.randn(2, I, generator=g).to(DEV)
    out = torch.full((2 * mq, I), 5.0, device=DEV)
    pre = torch.full((2 * mq, I), 6.0, device=DEV)
    _group(_problem(batch=2, m=mq, m_last=mt, n=I, k=C, A=x, lda=C, stride_a=mq * C, B=w, ldb=C, stride_b=I * C, C=out,
                    ldc=I, stride_c=mq * I, bias=b, stride_bias=I, act=1, aux=pre, tile=tile))
    ref0 = x[:mq].double() @ w[0].double().t() + b[0].double()
    ref1 = x[mq:mq + mt].double() @ w[1].double().t() + b[1].double()
    _close(pre[:mq], ref0)
    _close(pre[mq:mq + mt], ref1)
    _close(out[:mq], torch.nn.functional.gelu(ref0))
    _close(out[mq:mq + mt], torch.nn.functional.gelu(ref1))
    assert torch.all(out[mq + mt:] == 5.0) and torch.all(pre[mq + mt:] == 6.0)
    # the matching weight-gradient products: contraction over the (ragged) rows
    dy = torch.randn(2 * mq, I, generator=g).to(DEV)
    dw = torch.empty(2, I, C, device=DEV)
    db = torch.empty(2, I, device=DEV)
    # the contraction of the second element stops at its own row count (k_last): rows beyond are never read
    dy[mq + mt:] = float("nan")
    _group(_problem(amode=1, bmode=1, batch=2, m=I, n=C, k=mq, k_last=mt, A=dy, lda=I, stride_a=mq * I, B=x, ldb=C,
                    stride_b=mq * C, C=dw, ldc=C, stride_c=I * C, rowsum=db, stride_rowsum=I, tile=tile))
    _close(dw[0], dy[:mq].double().t() @ x[:mq].double())
    _close(dw[1], dy[mq:mq + mt].double().t() @ x[mq:mq + mt].double())
    _close(db[1], dy[mq:mq + mt].double().sum(0), tol=1e-5)


@pytest.mark.parametrize("tile", [0, 1, 2, 3])
def test_group_of_independent_products_in_one_launch(tile):
    """dX = dres + dY W (split K, atomics) and dW = dY^T X with db, two problems of different layouts in one
    launch -- what the backward pass of every Q-Former block issues."""
    g = torch.Generator().manual_seed(5)
    R, N, K = 416, 768, 3072
    dy = torch.randn(R, K, generator=g).to(DEV) * 0.1
    w = torch.randn(K, N, generator=g).to(DEV) * 0.1
    x = torch.randn(R, N, generator=g).to(DEV)
    dres = torch.randn(R, N, generator=g).to(DEV)
    gx = dres.clone()
    gw = torch.empty(K, N, device=DEV)
    gb = torch.empty(K, device=DEV)
    _group(_problem(amode=0, bmode=1, m=R, n=N, k=K, A=dy, lda=K, B=w, ldb=N, C=gx, ldc=N, accumulate=1, tile=tile),
           _problem(amode=1, bmode=1, m=K, n=N, k=R, A=dy, lda=K, B=x, ldb=N, C=gw, ldc=N, rowsum=gb, tile=tile))
    _close(gx, dres.double() + dy.double() @ w.double())
    _close(gw, dy.double().t() @ x.double())
    _close(gb, dy.double().sum(0), tol=1e-5)
