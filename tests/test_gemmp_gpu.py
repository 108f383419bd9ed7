"""sig3d_planes_split / sig3d_gemmp (csrc/gemmp_core.h, gemmp.hip): the f32 GEMM on the bf16 matrix cores over operands
that arrive as chunked bf16 planes -- the layer-batched weight gradients dW = dY^T X of the Q-Former's dense layers
(Qformer.py:116-118, 238, 305, 320) and the forward / input-gradient forms of the same layers -- against float64 torch,
through the C ABI: the three operand-orientation modes, the three tilings, ragged rows / reductions, batches with strides,
bias / GELU (pre-activation kept) / times gelu' / addend epilogues, planes of the result, split reductions."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def planes_ref(x, chunk_rows=None):
    """(..., R, C) f32 -> (..., C / 32, chunk_rows, 96) int16: x = p1 + p2 + p3, round to nearest even each time."""
    r, c = x.shape[-2:]
    chunk_rows = chunk_rows or r
    p1 = x.to(torch.bfloat16)
    r1 = x - p1.float()
    p2 = r1.to(torch.bfloat16)
    p3 = (r1 - p2.float()).to(torch.bfloat16)
    pl = torch.stack([p1, p2, p3], dim=-2)                                    # (..., R, 3, C)
    pl = pl.view(*x.shape[:-2], r, 3, c // 32, 32).movedim(-2, -4)            # (..., C/32, R, 3, 32)
    out = torch.zeros(*x.shape[:-2], c // 32, chunk_rows, 96, dtype=torch.int16, device=x.device)
    out[..., :r, :] = pl.reshape(*x.shape[:-2], c // 32, r, 96).view(torch.int16)
    return out


def planes_sum(planes, rows):
    """Inverse of planes_ref: (..., C / 32, chunk_rows, 96) int16 -> (..., rows, C) f32."""
    p = planes[..., :rows, :].contiguous().view(torch.bfloat16).float()
    p = p.view(*planes.shape[:-3], planes.shape[-3], rows, 3, 32)
    s = (p[..., 0, :] + p[..., 1, :]) + p[..., 2, :]                            # (..., C/32, rows, 32)
    return s.movedim(-3, -2).reshape(*planes.shape[:-3], rows, planes.shape[-3] * 32)


@pytest.mark.parametrize("batch,rows,cols,cap", [(1, 416, 768, 512), (3, 37, 64, 40), (2, 256, 3072, 256), (1, 1, 32, 8)])
def test_planes_split_is_exact_and_laid_out_in_chunks(batch, rows, cols, cap):
    from situation3d_amd import _lib as L
    g = torch.Generator().manual_seed(rows + cols)
    x = (torch.randn(batch, cap, cols, generator=g) * torch.logspace(-6, 3, cols)).to(DEV)
    planes = torch.full((batch, cols // 32, cap, 96), -1, dtype=torch.int16, device=DEV)
    L.planes_split(x, planes, rows=rows)
    ref = planes_ref(x[:, :rows], cap)
    assert torch.equal(planes[:, :, :rows], ref[:, :, :rows])
    assert bool((planes[:, :, rows:] == -1).all())                             # rows beyond are left alone
    assert torch.equal(planes_sum(planes, rows), x[:, :rows])                  # three bf16 terms ARE the f32 number


def _operands(modes, batch, m, n, k, g, ka=None):
    """f32 operands a (batch, m, k), w (batch, n, k) and their chunked planes in the orientation `modes` wants."""
    a = torch.randn(batch, m, k, generator=g).to(DEV)
    w = (torch.randn(batch, n, k, generator=g) * (1.0 / k ** 0.5)).to(DEV)
    xa = a.transpose(1, 2).contiguous() if modes == 2 else a                   # (k, m) rows for the weight gradient
    xb = w if modes == 0 else w.transpose(1, 2).contiguous()                   # (k, n) rows unless forward
    return a, w, planes_ref(xa), planes_ref(xb)


def _run(modes, batch, m, n, k, act=0, bias=False, addend=False, splits=1, config=0, seed=0, want_planes=False, want_c=True):
    from situation3d_amd import _lib as L
    g = torch.Generator().manual_seed(1000 * m + 10 * n + k + seed + modes)
    a, w, pa, pb = _operands(modes, batch, m, n, k, g)
    b = torch.randn(batch, n, generator=g).to(DEV) if bias else None
    add = torch.randn(batch, m, n, generator=g).to(DEV) if addend else None
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
    c = torch.full((batch, m, n), float("nan"), device=DEV) if want_c else None
    cp = torch.full((batch, n // 32, m, 96), -1, dtype=torch.int16, device=DEV) if want_planes else None
    work = counters = None
    if splits > 1:
        work = torch.empty(L.gemmp_work_floats(batch, m, n, splits, config), device=DEV)
        counters = torch.zeros(4096, dtype=torch.int32, device=DEV)
    ra, rb = pa.shape[-2], pb.shape[-2]
    for _ in range(2 if splits > 1 else 1):        # a split product leaves its counters at zero: a second launch must work
        L.gemmp(torch.device(DEV), A=pa, chunk_a=ra * 96, stride_a=pa[0].numel(), bytes_a=pa[0].numel() * 2,
                B=pb, chunk_b=rb * 96, stride_b=pb[0].numel(), bytes_b=pb[0].numel() * 2,
                C=c, ldc=n, stride_c=m * n, C_planes=cp, chunk_c=m * 96, stride_cp=(cp[0].numel() if want_planes else 0),
                bias=b, stride_bias=n, addend=add, aux=aux, work=work, counters=counters, modes=modes, batch=batch, m=m, n=n,
                k=k, act=act, splits=splits, config=config)
    if splits > 1:
        assert int(counters.abs().sum()) == 0
    return c, cp, aux, ref, pre_ref


def _close(got, ref, tol=2e-5):
    err = float((got.double() - ref).abs().max() / ref.abs().max().clamp_min(1e-20))
    assert err < tol, err


FWD_SHAPES = [(416, 2304, 768), (256, 768, 3072), (64, 64, 32), (104, 40, 96), (40, 136, 64), (8, 8, 32), (136, 264, 96)]


@pytest.mark.parametrize("m,n,k", FWD_SHAPES)
@pytest.mark.parametrize("config", [0, 1, 2, 3])
def test_forward_product_bias_gelu(m, n, k, config):
    c, _, aux, ref, pre = _run(0, 1, m, n, k, act=1, bias=True, config=config)
    _close(c, ref)
    _close(aux, pre)


@pytest.mark.parametrize("m,n,k", [(416, 768, 2304), (256, 3072, 768), (64, 64, 64), (104, 96, 32), (40, 160, 96), (8, 32, 32)])
@pytest.mark.parametrize("config", [0, 1, 2, 3])
def test_input_gradient_product_reads_the_weight_planes_transposed(m, n, k, config):
    c, _, _, ref, _ = _run(1, 1, m, n, k, addend=True, config=config)
    _close(c, ref)
    c, _, _, ref, _ = _run(1, 2, m, n, k, act=2, config=config, seed=3)
    _close(c, ref)


@pytest.mark.parametrize("m,n,k", [(768, 3072, 256), (2304, 768, 416), (768, 768, 208), (64, 64, 8), (96, 160, 40),
                                   (32, 32, 1000), (256, 9216 // 8, 2048)])
@pytest.mark.parametrize("config", [0, 1, 2, 3])
def test_weight_gradient_product_reduces_over_the_rows_of_both_operands(m, n, k, config):
    c, _, _, ref, _ = _run(2, 2, m, n, k, config=config)
    _close(c, ref)


def test_batches_advance_by_their_strides_and_only_live_rows_are_reduced():
    """The arena's use: per layer a (rows, cols) matrix of which the first `live` rows carry tokens; the second half of
    the rows belongs to another product (query branch / text branch of the feed-forward pair)."""
    from situation3d_amd import _lib as L
    g = torch.Generator().manual_seed(5)
    nl, rows, h, i_, p = 3, 512, 768, 3072, 256
    dy = torch.randn(nl, rows, h, generator=g).to(DEV)
    act = torch.randn(nl, rows, i_, generator=g).to(DEV)
    pdy = torch.empty(nl, h // 32, rows, 96, dtype=torch.int16, device=DEV)
    pact = torch.empty(nl, i_ // 32, rows, 96, dtype=torch.int16, device=DEV)
    L.planes_split(dy, pdy)
    L.planes_split(act, pact)
    gw = torch.full((nl, 2, h, i_), float("nan"), device=DEV)
    for half in range(2):
        off = half * p * 96
        L.gemmp(torch.device(DEV), A=pdy.data_ptr() + 2 * off, chunk_a=rows * 96, stride_a=pdy[0].numel(),
                bytes_a=(pdy[0].numel() - off) * 2, B=pact.data_ptr() + 2 * off, chunk_b=rows * 96, stride_b=pact[0].numel(),
                bytes_b=(pact[0].numel() - off) * 2, C=gw[:, half], ldc=i_, stride_c=2 * h * i_, modes=2, batch=nl, m=h, n=i_,
                k=p)
        ref = dy[:, half * p:(half + 1) * p].double().transpose(1, 2) @ act[:, half * p:(half + 1) * p].double()
        _close(gw[:, half], ref)


@pytest.mark.parametrize("modes,m,n,k", [(0, 416, 768, 768), (1, 256, 768, 3072), (0, 256, 768, 3072), (2, 128, 256, 512)])
@pytest.mark.parametrize("splits", [2, 3, 5])
@pytest.mark.parametrize("config", [1, 2, 3])
def test_split_reductions_meet_in_the_last_workgroup(modes, m, n, k, splits, config):
    c, _, _, ref, _ = _run(modes, 2 if modes != 2 else 1, m, n, k, bias=(modes == 0), addend=(modes == 1), splits=splits,
                           config=config)
    _close(c, ref)


def test_split_reduction_is_deterministic():
    a = _run(0, 1, 416, 768, 768, splits=3, config=2)[0]
    for _ in range(3):
        assert torch.equal(a, _run(0, 1, 416, 768, 768, splits=3, config=2)[0])


@pytest.mark.parametrize("modes,act", [(0, 1), (1, 2), (0, 0)])
def test_planes_of_the_result_are_the_result(modes, act):
    c, cp, _, ref, _ = _run(modes, 2, 256, 3072 if modes == 0 else 768, 768 if modes == 0 else 3072, act=act, bias=(modes == 0),
                            want_planes=True)
    _close(c, ref)
    assert torch.equal(planes_sum(cp, 256), c)
    only, cp2, _, _, _ = _run(modes, 2, 256, 3072 if modes == 0 else 768, 768 if modes == 0 else 3072, act=act,
                              bias=(modes == 0), want_planes=True, want_c=False)
    assert only is None and torch.equal(cp2, cp)


def test_six_bf16_products_over_planes_are_as_close_to_float64_as_an_f32_product():
    g = torch.Generator().manual_seed(11)
    m, n, k = 416, 2304, 768
    a, w, pa, pb = _operands(0, 1, m, n, k, g)
    from situation3d_amd import _lib as L
    c = torch.empty(1, m, n, device=DEV)
    L.gemmp(torch.device(DEV), A=pa, chunk_a=m * 96, stride_a=pa.numel(), bytes_a=pa.numel() * 2, B=pb, chunk_b=n * 96,
            stride_b=pb.numel(), bytes_b=pb.numel() * 2, C=c, ldc=n, stride_c=m * n, modes=0, batch=1, m=m, n=n, k=k)
    ref = a.double() @ w.double().transpose(1, 2)
    scale = float(ref.abs().max())
    own = float((c.double() - ref).abs().max()) / scale
    f32 = float(((a @ w.transpose(1, 2)).double() - ref).abs().max()) / scale
    assert own < 2e-6 and own < 2 * f32, (own, f32)


def test_bad_arguments_are_refused():
    from situation3d_amd import _lib as L
    t = torch.zeros(1, 4, 64, 96, dtype=torch.int16, device=DEV)
    c = torch.zeros(64, 64, device=DEV)
    kw = dict(A=t, chunk_a=64 * 96, stride_a=t.numel(), bytes_a=t.numel() * 2, B=t, chunk_b=64 * 96, stride_b=t.numel(),
              bytes_b=t.numel() * 2, C=c, ldc=64, stride_c=64 * 64, batch=1, m=64, n=64, k=128)
    with pytest.raises(L.Sig3dError):
        L.gemmp(torch.device(DEV), modes=0, **dict(kw, k=100))       # columns are the reduction index: chunks of 32
    with pytest.raises(L.Sig3dError):
        L.gemmp(torch.device(DEV), modes=3, **kw)
    with pytest.raises(L.Sig3dError):
        L.gemmp(torch.device(DEV), modes=0, splits=2, **kw)          # no work space
    with pytest.raises(L.Sig3dError):
        L.gemmp(torch.device(DEV), modes=0, **dict(kw, C=None))      # nothing requested
