"""sig3d_gemm16 (csrc/gemm16_core.h, gemm16.hip): the exact-f32 MFMA GEMM the Q-Former's dense layers run on
(Qformer.py:116-118, 238, 305, 320 and their input-gradient products) against float64 torch, through the C ABI:
both weight layouts, the three tilings, split reductions into slabs, batches with strides, bias / GELU (pre-activation
kept) / times gelu' / addend (also in place) epilogues, ragged shapes (rows, columns and k that are no multiples of the
tiles or of the 32-deep chunk), the step's own shapes."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _close(got, ref, tol=2e-5):
    ref = ref.to(got.dtype)
    err = float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-20))
    assert err < tol, err     # f32 accumulation of <= 9216 products: ~1e-6 relative to the largest entry


def _run(bmode, batch, m, n, k, act=0, bias=False, addend=None, splits=1, config=0, seed=0, in_place=False):
    """Returns (C summed over its slabs, aux, reference, reference pre-activation)."""
    from situation3d_amd import _lib as L
    g = torch.Generator().manual_seed(1000 * m + 10 * n + k + seed)
    a = torch.randn(batch, m, k, generator=g).to(DEV)
    w = (torch.randn(batch, n, k, generator=g) * (1.0 / k ** 0.5)).to(DEV)          # (n, k) rows
    b = torch.randn(batch, n, generator=g).to(DEV) if bias else None
    add = torch.randn(batch, m, n, generator=g).to(DEV) if addend else None
    wmat = w if bmode == 0 else w.transpose(1, 2).contiguous()                        # bmode 1: (k, n) rows
    ref = a.double() @ w.double().transpose(1, 2)
    if bias:
        ref = ref + b.double()[:, None, :]
    pre_ref = ref.clone()
    aux = None
    if act == 1:
        aux = torch.full((batch, m, n), float("nan"), device=DEV)
        ref = torch.nn.functional.gelu(ref)
    elif act == 2:
        aux = (torch.randn(batch, m, n, generator=g) * 2).to(DEV)
        u = aux.double()
        ref = ref * (0.5 * (1 + torch.erf(u / 2 ** 0.5)) + u * torch.exp(-0.5 * u * u) / (2 * torch.pi) ** 0.5)
    if addend:
        ref = ref + add.double()
    c = add.clone() if in_place else torch.full((batch, m, n), float("nan"), device=DEV)
    slabs = torch.full((max(splits - 1, 1), batch, m, n), float("nan"), device=DEV)
    L.gemm16(torch.device(DEV), A=a, lda=k, stride_a=m * k, B=wmat, ldb=(k if bmode == 0 else n), stride_b=n * k,
             C=c, ldc=n, stride_c=m * n, C_slabs=slabs if splits > 1 else None, slab_stride=batch * m * n,
             bias=b, stride_bias=n, addend=(c if in_place else add), aux=aux, bmode=bmode, batch=batch, m=m, n=n, k=k,
             act=act, splits=splits, config=config)
    total = c.double()
    if splits > 1:
        total = total + slabs.double().sum(0)
    return total, aux, ref, pre_ref


SHAPES = [(416, 2304, 768), (256, 768, 3072), (64, 64, 32), (100, 36, 72), (37, 129, 36), (5, 4, 4), (130, 260, 96),
          (16, 16, 64), (33, 17 * 4, 100)]


@pytest.mark.parametrize("m,n,k", SHAPES)
@pytest.mark.parametrize("config", [0, 1, 2, 3])
def test_forward_product_bias_gelu(m, n, k, config):
    """y = x W^T + b; gelu(y) with the pre-activation kept (BertIntermediate, Qformer.py:305-313)."""
    out, _, ref, _ = _run(0, 1, m, n, k, bias=True, config=config)
    _close(out, ref)
    out, aux, ref, pre = _run(0, 1, m, n, k, act=1, bias=True, config=config)
    _close(aux, pre)
    _close(out, ref)


@pytest.mark.parametrize("m,n,k", [(m, (n + 3) // 4 * 4, k) for m, n, k in SHAPES])   # n-contiguous rows: n % 4 == 0
@pytest.mark.parametrize("config", [0, 1, 2, 3])
def test_input_gradient_product(m, n, k, config):
    """dx = dy W with the weight read along its rows (bmode 1), plain, times gelu'(pre), plus the residual gradient."""
    out, _, ref, _ = _run(1, 1, m, n, k, config=config)
    _close(out, ref)
    out, _, ref, _ = _run(1, 1, m, n, k, act=2, config=config)
    _close(out, ref)
    out, _, ref, _ = _run(1, 1, m, n, k, addend=True, config=config)
    _close(out, ref)
    out, _, ref, _ = _run(1, 1, m, n, k, addend=True, in_place=True, config=config)
    _close(out, ref)


@pytest.mark.parametrize("bmode", [0, 1])
@pytest.mark.parametrize("m,n,k,splits", [(416, 768, 768, 3), (256, 768, 3072, 5), (416, 768, 2304, 6), (256, 768, 768, 5),
                                          (100, 36, 200, 2), (37, 132, 96, 3), (64, 64, 64, 2), (50, 52, 2048, 8)])
def test_split_reduction_writes_slabs_that_sum_to_the_product(bmode, m, n, k, splits):
    """Every split writes its own slab (bias and addend go with split 0): nothing is pre-zeroed, nothing is atomic."""
    for config in (1, 2, 3):
        out, _, ref, _ = _run(bmode, 1, m, n, k, bias=True, addend=True, splits=splits, config=config)
        _close(out, ref)
        out, _, ref, _ = _run(bmode, 1, m, n, k, addend=True, in_place=True, splits=splits, config=config)
        _close(out, ref)


@pytest.mark.parametrize("bmode", [0, 1])
@pytest.mark.parametrize("config", [0])
def test_batched_feed_forward_pair(bmode, config):
    """The query branch and the text branch of a layer (different weights, biases) as one launch of batch 2."""
    out, aux, ref, pre = _run(bmode, 2, 256, 3072, 768, act=(1 if bmode == 0 else 2), bias=(bmode == 0), config=config)
    _close(out, ref)
    out, _, ref, _ = _run(bmode, 2, 256, 768, 3072, addend=True, splits=5, bias=(bmode == 0), config=config)
    _close(out, ref)


def test_heuristic_splits_are_usable():
    from situation3d_amd import _lib as L
    for bmode, batch, m, n, k in [(0, 1, 416, 768, 768), (0, 2, 256, 768, 3072), (1, 1, 416, 768, 2304),
                                  (0, 1, 256, 768, 768), (0, 1, 2048, 9216, 256), (1, 1, 2048, 256, 9216)]:
        s = L.gemm16_splits(bmode, batch, m, n, k)
        assert 1 <= s <= 8
        out, _, ref, _ = _run(bmode, batch, m, n, k, splits=s)
        _close(out, ref)
    assert L.gemm16_splits(0, 2, 256, 3072, 768, 1) == 1      # an activation needs the whole sum


def test_argument_errors_are_reported():
    from situation3d_amd import _lib as L
    a = torch.zeros(8, 6, device=DEV)
    with pytest.raises(L.Sig3dError):      # k % 4 != 0
        L.gemm16(torch.device(DEV), A=a, lda=6, B=a, ldb=6, C=a, ldc=8, m=8, n=8, k=6)
    b = torch.zeros(8, 8, device=DEV)
    with pytest.raises(L.Sig3dError):      # a split product needs slabs
        L.gemm16(torch.device(DEV), A=b, lda=8, B=b, ldb=8, C=b, ldc=8, m=8, n=8, k=8, splits=2)
    with pytest.raises(L.Sig3dError):      # ... and has no activation
        L.gemm16(torch.device(DEV), A=b, lda=8, B=b, ldb=8, C=b, ldc=8, C_slabs=b, m=8, n=8, k=64, splits=2, act=1)
