"""Fused SharedMLP + max-pool kernels (csrc/shared_mlp.hip) against the layer-by-layer torch path
of the same module (Conv2d 1x1 -> BatchNorm2d(train) -> ReLU, then max over nsample;
lib/pointnet2/pytorch_utils.py:11-36, pointnet2_modules.py:251-262): outputs, input gradient,
every parameter gradient and the BatchNorm running statistics.  fp32 within 1e-4 (scaled).
"""
import copy
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _close(a, b, name, tol=1e-4):
    scale = max(1.0, b.abs().max().item())
    torch.testing.assert_close(a, b, rtol=1e-3, atol=tol * scale, msg=lambda m: name + ": " + m)


@pytest.mark.parametrize("b,chans,p,s", [(2, [6, 64, 64, 128], 100, 64), (3, [131, 128, 128, 256], 64, 32),
                                         (2, [259, 128, 128, 256], 33, 16), (1, [9, 32], 5, 7),
                                         (2, [35, 64, 32], 17, 12),
                                         (1, [6, 32, 64], 300, 64)])  # rows of 19200 positions: full 8192-chunks + tail
@pytest.mark.parametrize("library_gemm", [False, True])
def test_fused_mlp_max_matches_torch(b, chans, p, s, library_gemm):
    from situation3d_amd.pointnet2 import fused_mlp
    from situation3d_amd.pointnet2.pytorch_utils import SharedMLP
    torch.manual_seed(sum(chans) + p)
    mlp = SharedMLP(list(chans), bn=True).to(DEV).train()
    with torch.no_grad():
        for m in mlp.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.2)
                m.running_mean.normal_(0, 0.1)
    ref = copy.deepcopy(mlp)
    x = torch.randn(b, chans[0], p, s, device=DEV)
    x1 = x.clone().requires_grad_(True)
    x2 = x.clone().requires_grad_(True)
    g = torch.randn(b, chans[-1], p, device=DEV)

    assert fused_mlp.can_fuse(mlp, x1)
    # library_gemm=False: MFMA layer kernels; True: torch.bmm convolutions + the same BN / pool kernels
    out = fused_mlp.fused_mlp_max(mlp, x1, library_gemm=library_gemm)
    (out * g).sum().backward()
    exp = torch.max(ref(x2), dim=3)[0]
    (exp * g).sum().backward()

    _close(out, exp.detach(), "output")
    _close(x1.grad, x2.grad, "input grad")
    for (n1, p1), (_, p2) in zip(mlp.named_parameters(), ref.named_parameters()):
        _close(p1.grad, p2.grad, n1)
    for (n1, b1), (_, b2) in zip(mlp.named_buffers(), ref.named_buffers()):
        _close(b1.float(), b2.float(), n1, tol=1e-5)


def test_fused_path_is_taken_by_sa_module_in_train_and_inference():
    from situation3d_amd.pointnet2 import fused_mlp, pointnet2_modules
    from util import feats, scene
    mod = pointnet2_modules.PointnetSAModuleVotes(npoint=64, radius=0.8, nsample=16, mlp=[5, 32, 64],
                                                  use_xyz=True).to(DEV)
    xyz, f = scene(2, 500, seed=1).to(DEV), feats(2, 5, 500).to(DEV)
    calls = []
    orig, orig_min, orig_c = fused_mlp.fused_mlp_max, fused_mlp.MIN_POSITIONS, fused_mlp.fused_sa_compact
    fused_mlp.fused_mlp_max = lambda m, x, **k: calls.append(1) or orig(m, x, **k)
    # training mode above MIN_POSITIONS runs over the distinct neighbours only: also a fused path
    fused_mlp.fused_sa_compact = lambda *a, **k: calls.append(1) or orig_c(*a, **k)
    fused_mlp.MIN_POSITIONS = 0
    try:
        mod.train()
        _, a, _ = mod(xyz, f)
        assert calls == [1]
        mod.eval()
        _, bb, _ = mod(xyz, f)  # eval mode WITH autograd: the unfused torch path (running stats)
        assert calls == [1]
        with torch.no_grad():
            _, cc, _ = mod(xyz, f)  # eval mode, no gradient wanted: the fused inference path
        assert calls == [1, 1]
        torch.testing.assert_close(cc, bb, rtol=1e-4, atol=1e-4)
    finally:
        fused_mlp.fused_mlp_max, fused_mlp.MIN_POSITIONS, fused_mlp.fused_sa_compact = orig, orig_min, orig_c
    assert a.shape == bb.shape == (2, 64, 64)


@pytest.mark.parametrize("s", [64, 32, 16, 24])
def test_bn_relu_maxpool_first_maximum_on_ties(s):
    """Ball-query padding repeats the first neighbour, so tied maxima are the common case: the arg-max
    must be the FIRST maximal sample (F.max_pool2d semantics), the value max(relu(y*scale+shift))."""
    import torch.nn.functional as F
    from situation3d_amd import _lib as L
    b, c, p = 2, 5, 37
    g = torch.Generator().manual_seed(s)
    y = torch.randn(b, c, p, s, generator=g)
    y[..., s // 3:] = y[..., :1]            # padded tail = copies of sample 0
    y[:, :, ::3, 1] = y[:, :, ::3, 0]       # and an early tie
    scale = torch.tensor([1.0, 0.5, -1.0, 2.0, 1.0])
    shift = torch.tensor([0.0, 0.1, 0.2, -50.0, 0.3])   # channel 3: everything below zero after BN
    z = F.relu(y * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    exp, idx = F.max_pool2d(z, [1, s], return_indices=True)     # CPU: first maximum
    out = torch.empty(b, c, p, device=DEV)
    arg = torch.empty(b, c, p, dtype=torch.int32, device=DEV)
    yd, sd, hd = y.to(DEV), scale.to(DEV), shift.to(DEV)
    L.call("sig3d_bn_relu_maxpool", b, c, p, s, L.ptr(yd), L.ptr(sd), L.ptr(hd), L.ptr(out), L.ptr(arg),
           L.stream_ptr())
    assert torch.equal(out.cpu(), exp.squeeze(-1))
    assert torch.equal(arg.cpu().long(), idx.squeeze(-1) % s)


@pytest.mark.parametrize("c_feat,mlp,npoint,nsample,radius", [(3, [3, 64, 64, 128], 1024, 64, 0.2),
                                                               (128, [128, 128, 128, 256], 1024, 32, 0.4),
                                                               (36, [36, 32, 64], 2048, 32, 0.3)])
def test_compact_sa_level_matches_dense(c_feat, mlp, npoint, nsample, radius):
    """One SA level over the DISTINCT neighbours only (csrc/compact.hip) against the dense fused path:
    outputs, input-feature gradient, every weight / BatchNorm gradient and the running statistics.
    Same arithmetic up to summation order: 2e-5 relative to the largest magnitude (observed ~1e-6)."""
    import copy
    from situation3d_amd.pointnet2 import fused_mlp
    from situation3d_amd.pointnet2.pointnet2_modules import PointnetSAModuleVotes
    torch.manual_seed(7)
    b, n = 2, 6000
    xyz = (torch.rand(b, n, 3) * torch.tensor([8.0, 8.0, 3.0])).to(DEV)
    feats = torch.randn(b, c_feat, n).to(DEV)
    sa = PointnetSAModuleVotes(npoint=npoint, radius=radius, nsample=nsample, mlp=list(mlp), use_xyz=True,
                               normalize_xyz=True).to(DEV).train()
    sa2 = copy.deepcopy(sa)
    g = None
    results = []
    for module, compact in ((sa, True), (sa2, False)):
        fused_mlp.COMPACT = compact
        try:
            f = feats.clone().requires_grad_(True)
            new_xyz, out, inds = module(xyz, f)
            if g is None:
                g = torch.randn_like(out)
            (out * g).sum().backward()
        finally:
            fused_mlp.COMPACT = True
        results.append((out.detach(), f.grad, [p.grad for p in module.parameters()],
                        [bf.clone() for bf in module.buffers()]))
    (o1, f1, p1, b1), (o2, f2, p2, b2) = results

    def close(a, b_, what):
        err = (a - b_).abs().max().item()
        assert err <= 2e-5 * max(1.0, b_.abs().max().item()), (what, err)
    close(o1, o2, "output")
    close(f1, f2, "feature gradient")
    for i, (a, b_) in enumerate(zip(p1, p2)):
        close(a, b_, "parameter gradient %d" % i)
    for i, (a, b_) in enumerate(zip(b1, b2)):
        close(a.float(), b_.float(), "buffer %d" % i)
    # the fused inference path (eval mode, no gradient) takes the compact route too
    sa.eval()
    sa2.load_state_dict(sa.state_dict())
    sa2.eval()
    with torch.no_grad():
        e1 = sa(xyz, feats)[1]
        fused_mlp.COMPACT = False
        try:
            e2 = sa2(xyz, feats)[1]
        finally:
            fused_mlp.COMPACT = True
    close(e1, e2, "eval output")


@pytest.mark.parametrize("b,m,ns", [(3, 300, 16), (2, 257, 64), (2, 130, 32), (1, 61, 8), (2, 90, 12), (1, 2048, 64),
                                    (1, 8200, 16)])
def test_compact_lists_structure(b, m, ns):
    """sig3d_compact_neighbour_lists against a direct construction from the padded lists: the wave-per-row kernel
    (nsample 8 / 16 / 32 / 64, m <= 8192: rows not a multiple of the rows per wave, full and single-entry rows) and
    the thread-per-row one (any nsample, any m)."""
    from situation3d_amd.pointnet2 import fused_mlp
    g = torch.Generator().manual_seed(3 + m)
    n = 4 * m + 100
    ks = torch.randint(1, ns + 1, (b, m), generator=g)
    ks[:, 0] = ns                                      # a full row
    ks[:, -1] = 1                                      # a row that is all padding
    idx = torch.zeros(b, m, ns, dtype=torch.int32)
    for bi in range(b):
        for j in range(m):
            k = int(ks[bi, j])
            hits = torch.sort(torch.randperm(n, generator=g)[:k]).values.int()
            idx[bi, j, :k] = hits
            idx[bi, j, k:] = hits[0]
    cl = fused_mlp.compact_lists(idx.to(DEV))
    cidx, cent, mult, seg, nact = [t.cpu() for t in cl.tensors()]
    for bi in range(b):
        u = 0
        for j in range(m):
            row = idx[bi, j]
            k = 1 + int((row[1:] != row[0]).sum())
            assert seg[bi, j] == u
            assert torch.equal(cidx[bi, u:u + k], row[:k]) and (cent[bi, u:u + k] == j).all()
            assert mult[bi, u] == ns - k + 1 and (mult[bi, u + 1:u + k] == 1).all()
            u += k
        assert seg[bi, m] == u and nact[bi] == u


def test_compact_mode_is_declined_for_full_neighbour_lists():
    """The per-layer decision is taken from the first batch: sparse lists -> compact, full lists -> dense."""
    from situation3d_amd.pointnet2.pointnet2_modules import PointnetSAModuleVotes
    torch.manual_seed(1)
    b, n = 2, 6000
    xyz = (torch.rand(b, n, 3) * torch.tensor([8.0, 8.0, 3.0])).to(DEV)
    feats = torch.randn(b, 3, n).to(DEV)
    for radius, expect in ((0.2, True), (3.0, False)):
        sa = PointnetSAModuleVotes(npoint=1024, radius=radius, nsample=64, mlp=[3, 32, 32], use_xyz=True).to(DEV).train()
        out = sa(xyz, feats)[1]
        assert sa._compact_decision is expect and torch.isfinite(out).all()


@pytest.mark.parametrize("npoint,nsample,radius", [(333, 6, 0.25), (500, 7, 0.05), (64, 16, 0.6)])
def test_compact_sa_level_odd_shapes(npoint, nsample, radius, monkeypatch):
    """Compact mode on shapes outside the vector paths: positions per row not a multiple of 4, partial last
    tiles, centres without any neighbour (all-zero list = point 0 repeated), nearly full lists."""
    import copy
    from situation3d_amd.pointnet2 import fused_mlp
    from situation3d_amd.pointnet2.pointnet2_modules import PointnetSAModuleVotes
    monkeypatch.setattr(fused_mlp, "MIN_POSITIONS", 0)
    monkeypatch.setattr(fused_mlp, "COMPACT_MIN_POSITIONS", 0)
    monkeypatch.setattr(PointnetSAModuleVotes, "COMPACT_MAX_FRACTION", 1.0)
    torch.manual_seed(11)
    b, n = 3, 1500
    xyz = (torch.rand(b, n, 3) * torch.tensor([4.0, 4.0, 2.0])).to(DEV)
    xyz[:, :200] += 100.0          # far-away points: their own ball only
    feats = torch.randn(b, 5, n).to(DEV)
    sa = PointnetSAModuleVotes(npoint=npoint, radius=radius, nsample=nsample, mlp=[5, 32, 64], use_xyz=True).to(DEV).train()
    sa2 = copy.deepcopy(sa)
    g, res = None, []
    for module, compact in ((sa, True), (sa2, False)):
        monkeypatch.setattr(fused_mlp, "COMPACT", compact)
        f = feats.clone().requires_grad_(True)
        out = module(xyz, f)[1]
        g = torch.randn_like(out) if g is None else g
        (out * g).sum().backward()
        res.append((out.detach(), f.grad, [p.grad for p in module.parameters()]))
    assert sa._compact_decision is True and getattr(sa2, "_compact_decision", None) is None
    (o1, f1, p1), (o2, f2, p2) = res
    for a, b_, what in [(o1, o2, "out"), (f1, f2, "dfeat")] + [(x, y, "dparam") for x, y in zip(p1, p2)]:
        err = (a - b_).abs().max().item()
        assert err <= 3e-5 * max(1.0, b_.abs().max().item()), (what, err)


@pytest.mark.parametrize("b,n,m,ns,c,cout,normalize", [(2, 500, 64, 32, 128, 128, False), (1, 300, 33, 16, 256, 128, True),
                                                       (3, 200, 20, 7, 32, 40, False), (2, 1024, 128, 32, 64, 256, True)])
def test_first_layer_with_the_grouped_operand_gathered_on_load(b, n, m, ns, c, cout, normalize):
    """sig3d_mlp_layer0_gather_fwd == sig3d_query_group_fused_pm followed by
    sig3d_mlp_layer_fwd on the stored tensor: raw conv output and BatchNorm sums, dense lists."""
    import ctypes
    from situation3d_amd import _lib as L
    from util import scene
    g = torch.Generator().manual_seed(b * n + c)
    xyz = scene(b, n, seed=n + c).to(DEV)
    new_xyz = xyz[:, :m].contiguous()
    idx = torch.randint(0, n, (b, m, ns), generator=g, dtype=torch.int32).to(DEV)
    feat_pm = torch.randn(b, n, c, generator=g).to(DEV)
    w = (torch.randn(cout, c + 3, generator=g) * 0.2).to(DEV)
    radius = 0.7
    e = m * ns
    s = L.stream_ptr(torch.device(DEV))
    grouped = torch.empty(b, c + 3, m, ns, device=DEV)
    L.call("sig3d_query_group_fused_pm", b, n, m, c, c, ns, 1, int(normalize), ctypes.c_float(radius), L.ptr(xyz),
           L.ptr(new_xyz), L.ptr(feat_pm), L.ptr(idx), L.ptr(grouped), s)
    y_ref = torch.empty(b, cout, e, device=DEV)
    st_ref = torch.zeros(2, cout, dtype=torch.float64, device=DEV)
    L.call("sig3d_mlp_layer_fwd", b, c + 3, cout, e, L.ptr(grouped), L.ptr(w), L.ptr(None), L.ptr(None), L.ptr(y_ref),
           L.ptr(st_ref[0]), L.ptr(st_ref[1]), 0, s)
    y = torch.full((b, cout, e), 7.0, device=DEV)
    st = torch.zeros(2, cout, dtype=torch.float64, device=DEV)
    L.call("sig3d_mlp_layer0_gather_fwd", b, n, m, ns, c, cout, int(normalize), ctypes.c_float(radius), L.ptr(xyz),
           L.ptr(new_xyz), L.ptr(feat_pm), L.ptr(idx), L.ptr(w), L.ptr(y), L.ptr(st[0]), L.ptr(st[1]), 0, L.ptr(None),
           L.ptr(None), L.ptr(None), s)
    exact = torch.einsum("oc,bce->boe", w.double(), grouped.view(b, c + 3, e).double())
    _close(y_ref, exact.float(), "stored-tensor kernel vs float64")
    _close(y, exact.float(), "gathering kernel vs float64")
    _close(st.float(), st_ref.float(), "BatchNorm sums", tol=1e-5)
    # backward products of the same layer: dW with the operand gathered again, dX scattered in the epilogue
    dy = torch.randn(b, cout, e, generator=g).to(DEV)
    dw_ref = torch.zeros(cout, c + 3, device=DEV)
    L.call("sig3d_mlp_layer_dw", b, c + 3, cout, e, L.ptr(dy), L.ptr(grouped), L.ptr(None), L.ptr(None), L.ptr(dw_ref), 0, s)
    dw = torch.full((cout, c + 3), 3.0, device=DEV)
    L.call("sig3d_mlp_layer0_gather_dw", b, n, m, ns, c, cout, int(normalize), ctypes.c_float(radius), L.ptr(xyz),
           L.ptr(new_xyz), L.ptr(feat_pm), L.ptr(idx), L.ptr(dy), L.ptr(dw), 0, L.ptr(None), L.ptr(None), s)
    dw_exact = torch.einsum("boe,bce->oc", dy.double(), grouped.view(b, c + 3, e).double())
    _close(dw_ref, dw_exact.float(), "stored-tensor dW vs float64")
    _close(dw, dw_exact.float(), "gathering dW vs float64")
    wt = w.t().contiguous()
    gpm = torch.zeros(b, n, c, device=DEV)
    L.call("sig3d_mlp_layer0_scatter_dx", b, n, m, ns, c, cout, L.ptr(idx), L.ptr(dy), L.ptr(wt), L.ptr(gpm), L.ptr(None), s)
    dx = torch.einsum("co,boe->bce", wt.double(), dy.double())[:, 3:]            # (b, c, e): feature rows only
    exp = torch.zeros(b, n, c, dtype=torch.float64, device=DEV)
    exp.scatter_add_(1, idx.view(b, e, 1).long().expand(b, e, c), dx.transpose(1, 2).contiguous())
    _close(gpm, exp.float(), "scattered input gradient")


@pytest.mark.parametrize("mode", ["dense", "compact", "dense-eval"])
def test_sa_level_without_the_grouped_tensor_matches_the_stored_form(mode, monkeypatch):
    """PointnetSAModuleVotes on the MFMA path with the first SharedMLP layer gathering its operand on load
    (fused_mlp.fused_sa_dense / fused_sa_compact) against the same module with the
    grouped tensor stored (fused_mlp.GATHER_L0 = False): features out, feature gradient in, every parameter gradient and
    the BatchNorm running statistics."""
    from situation3d_amd.pointnet2 import fused_mlp, pointnet2_modules
    from util import scene
    b, n, c, npoint, ns = 2, 4096, 128, 1024, 64
    torch.manual_seed(11)
    mod = pointnet2_modules.PointnetSAModuleVotes(npoint=npoint, radius=0.9 if mode != "compact" else 0.25, nsample=ns,
                                                  mlp=[c, 128, 128, 256], use_xyz=True, normalize_xyz=True).to(DEV)
    ref = copy.deepcopy(mod)
    if mode == "dense-eval":
        mod.eval(); ref.eval()
    xyz = scene(b, n, seed=3).to(DEV)
    feats = torch.randn(b, c, n, device=DEV)
    g = torch.randn(b, 256, npoint, device=DEV)
    monkeypatch.setattr(fused_mlp, "COMPACT", mode == "compact")

    def run(m, gather):
        monkeypatch.setattr(fused_mlp, "GATHER_L0", gather)
        f = feats.clone().requires_grad_(mode != "dense-eval")
        with torch.set_grad_enabled(mode != "dense-eval"):
            new_xyz, out, inds = m(xyz, f)
            if mode != "dense-eval":
                (out * g).sum().backward()
        return out.detach(), (f.grad if mode != "dense-eval" else None), inds

    calls = []
    orig = fused_mlp._lib.call
    monkeypatch.setattr(fused_mlp._lib, "call", lambda name, *a: (calls.append(name), orig(name, *a))[1])
    out1, gf1, i1 = run(mod, True)
    assert "sig3d_mlp_layer0_gather_fwd" in calls
    if mode == "compact":   # no grouped tensor in the forward pass; the backward pass re-materialises the few MB of
        # DISTINCT neighbours for the streaming weight gradient (recompute in backward) or,
        # with fused_mlp.DW_REGROUP = False, gathers them inside the weight-gradient kernel; the input gradient is scattered
        fwd_end = calls.index("sig3d_bn_relu_maxpool_compact") if "sig3d_bn_relu_maxpool_compact" in calls else \
            max(i for i, nm in enumerate(calls) if nm.startswith("sig3d_bn_relu_maxpool"))
        assert not any(nm.startswith("sig3d_query_group") for nm in calls[:fwd_end + 1])
        assert "sig3d_mlp_layer0_gather_dw" in calls or \
            calls.index("sig3d_query_group_compact") < calls.index("sig3d_mlp_layer_dw_stream_nofold", calls.index("sig3d_query_group_compact"))
        assert "sig3d_mlp_layer0_scatter_dx_w" in calls
    elif mode == "dense":   # forward without it; the backward re-materialises it (faster than gathering twice more)
        assert calls.index("sig3d_query_group_fused_pm") > calls.index("sig3d_bn_relu_maxpool")
    else:
        assert not any(nm.startswith("sig3d_query_group") for nm in calls)
    if mode == "compact":
        assert mod._compact_decision is True
    calls.clear()
    out2, gf2, i2 = run(ref, False)
    assert "sig3d_mlp_layer0_gather_fwd" not in calls and any(nm.startswith("sig3d_query_group") for nm in calls)
    assert torch.equal(i1, i2)
    _close(out1, out2, "features")
    if mode != "dense-eval":
        # the two forms sum the first layer in different orders: a max-pool winner that leads by one ulp can
        # change, which moves the gradient of ONE (channel, centre) to another neighbour -- all input channels of
        # two source points.  Everything else must agree to 1e-4.
        bad = (gf1 - gf2).abs() > 1e-4 * max(1.0, float(gf2.abs().max())) + 1e-3 * gf2.abs()
        assert bad.float().mean() < 1e-3, bad.float().mean()
        assert bad.any(dim=1).sum() <= 8                      # ... confined to a handful of source points
        for (n1, p1), (_, p2) in zip(mod.named_parameters(), ref.named_parameters()):
            rel = float((p1.grad - p2.grad).norm() / p2.grad.norm())       # (a changed winner moves single terms)
            assert rel < 1e-3, (n1, rel)
        for (n1, b1), (_, b2) in zip(mod.named_buffers(), ref.named_buffers()):
            _close(b1.float(), b2.float(), n1, tol=1e-5)


@pytest.mark.parametrize("s,c,p", [(64, 5, 37), (32, 128, 70), (16, 40, 512), (24, 33, 19)])
def test_bn_relu_maxpool_pm_writes_both_layouts(s, c, p):
    """sig3d_bn_relu_maxpool_pm == sig3d_bn_relu_maxpool (values and first-maximum indices, bit for bit) plus the
    point-major copy (b, p, c) the next level / the Q-Former reads -- tiles that overhang c and p included."""
    from situation3d_amd import _lib as L
    b = 2
    g = torch.Generator().manual_seed(s + c)
    y = torch.randn(b, c, p, s, generator=g)
    y[..., s // 3:] = y[..., :1]
    scale = (torch.rand(c, generator=g) + 0.5) * torch.where(torch.arange(c) % 7 == 3, -1.0, 1.0)
    shift = torch.randn(c, generator=g) * 0.3
    yd, sd, hd = y.to(DEV), scale.to(DEV), shift.to(DEV)
    out0, arg0 = torch.empty(b, c, p, device=DEV), torch.empty(b, c, p, dtype=torch.int32, device=DEV)
    L.call("sig3d_bn_relu_maxpool", b, c, p, s, L.ptr(yd), L.ptr(sd), L.ptr(hd), L.ptr(out0), L.ptr(arg0), L.stream_ptr())
    out1, arg1 = torch.full((b, c, p), -7.0, device=DEV), torch.full((b, c, p), -7, dtype=torch.int32, device=DEV)
    pm = torch.full((b, p, c), -7.0, device=DEV)
    L.call("sig3d_bn_relu_maxpool_pm", b, c, p, s, p * s, L.ptr(yd), L.ptr(sd), L.ptr(hd), L.ptr(None), L.ptr(out1),
           L.ptr(arg1), L.ptr(pm), L.stream_ptr())
    assert torch.equal(out1, out0) and torch.equal(arg1, arg0)
    assert torch.equal(pm, out0.transpose(1, 2))


def test_bn_relu_maxpool_pm_compact_lists():
    from situation3d_amd import _lib as L
    b, c, p, ns = 2, 70, 45, 8
    g = torch.Generator().manual_seed(4)
    counts = torch.randint(1, ns + 1, (b, p), generator=g)
    seg = torch.zeros(b, p + 1, dtype=torch.int32)
    seg[:, 1:] = counts.cumsum(1)
    e = p * ns
    y = torch.randn(b, c, e, generator=g)
    scale, shift = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.2
    yd, sd, hd, segd = y.to(DEV), scale.to(DEV), shift.to(DEV), seg.to(DEV)
    out0, arg0 = torch.empty(b, c, p, device=DEV), torch.empty(b, c, p, dtype=torch.int32, device=DEV)
    L.call("sig3d_bn_relu_maxpool_compact", b, c, p, e, L.ptr(yd), L.ptr(sd), L.ptr(hd), L.ptr(segd), L.ptr(out0),
           L.ptr(arg0), L.stream_ptr())
    out1, arg1 = torch.empty_like(out0), torch.empty_like(arg0)
    pm = torch.empty(b, p, c, device=DEV)
    L.call("sig3d_bn_relu_maxpool_pm", b, c, p, ns, e, L.ptr(yd), L.ptr(sd), L.ptr(hd), L.ptr(segd), L.ptr(out1),
           L.ptr(arg1), L.ptr(pm), L.stream_ptr())
    assert torch.equal(out1, out0) and torch.equal(arg1, arg0) and torch.equal(pm, out0.transpose(1, 2))


@pytest.mark.parametrize("second", ["compact-gather", "dense-gather", "library"])
def test_chained_levels_hand_features_over_point_major_without_transposes(second, monkeypatch):
    """Two stacked SA levels with emit_point_major: the first level's pooling kernel writes the (B, npoint, C) twin,
    the second level gathers from it (MFMA path: first layer gathering on load; small level: point-major grouping)
    and returns its feature gradient point-major -- same outputs bit for bit and the same gradients as the plain
    chain, with NO transpose launch in either pass."""
    import copy
    from situation3d_amd.pointnet2 import fused_mlp
    from situation3d_amd.pointnet2.pointnet2_modules import PointnetSAModuleVotes
    from util import scene
    torch.manual_seed(5)
    b, n = 2, 5000
    xyz = scene(b, n, seed=9).to(DEV)
    feats = torch.randn(b, 3, n, device=DEV)
    r2, ns2, m2 = {"compact-gather": (0.15, 32, 2048), "dense-gather": (0.9, 64, 1024), "library": (0.8, 16, 256)}[second]
    monkeypatch.setattr(fused_mlp, "COMPACT", second == "compact-gather")
    l1 = PointnetSAModuleVotes(npoint=2048, radius=0.3, nsample=16, mlp=[3, 32, 64], use_xyz=True, normalize_xyz=True).to(DEV)
    l2 = PointnetSAModuleVotes(npoint=m2, radius=r2, nsample=ns2, mlp=[64, 64, 128], use_xyz=True, normalize_xyz=True).to(DEV)
    r1, r2m = copy.deepcopy(l1), copy.deepcopy(l2)
    l1.emit_point_major = l2.emit_point_major = True
    G = torch.randn(b, m2, 128, device=DEV)

    def run(a, c, pm):
        f = feats.clone().requires_grad_(True)
        x1, f1, _ = a(xyz, f)
        x2, f2, _ = c(x1, f1)
        if pm:
            tw = fused_mlp.point_major_of(f2)
            assert tw is not None and torch.equal(tw, f2.transpose(1, 2))
            (tw * G).sum().backward()                      # the consumer reads the point-major twin (the Q-Former does)
        else:
            (f2.transpose(1, 2) * G).sum().backward()
        return f2.detach(), f.grad, [p.grad for p in list(a.parameters()) + list(c.parameters())]

    calls = []
    orig = fused_mlp._lib.call
    monkeypatch.setattr(fused_mlp._lib, "call", lambda name, *a: (calls.append(name), orig(name, *a))[1])
    out1, gf1, gp1 = run(l1, l2, True)
    assert calls.count("sig3d_bn_relu_maxpool_pm") == 2
    # no transpose launch in either direction: the point-major output gradient is turned inside the top layer's
    # statistics pass of each level
    assert calls.count("sig3d_transpose_cn") == 0, calls
    assert calls.count("sig3d_bn_relu_bwd_top_from_pm") == 2
    calls.clear()
    out2, gf2, gp2 = run(r1, r2m, False)
    assert "sig3d_bn_relu_maxpool_pm" not in calls
    assert torch.equal(out1, out2)
    _close(gf1, gf2, "input feature gradient", tol=1e-4)
    for a, c in zip(gp1, gp2):
        _close(a, c, "parameter gradient", tol=1e-4)


@pytest.mark.parametrize("compact", [False, True])
def test_top_layer_backward_from_a_point_major_pool_gradient(compact):
    """sig3d_bn_relu_bwd_top_from_pm: the point-major pool gradient -> its channel-major copy + the top layer's
    statistics in one launch; followed by sig3d_bn_relu_bwd[_compact](accumulate = 2) it must give the dY and the
    sums of the plain (channel-major gradient) call."""
    from situation3d_amd import _lib as L
    b, c, p, ns = 2, 70, 150, 16
    g = torch.Generator().manual_seed(9 + compact)
    e = p * ns
    y = torch.randn(b, c, e, generator=g).to(DEV)
    scale = (torch.rand(c, generator=g) + 0.5).to(DEV)
    shift = (torch.randn(c, generator=g) * 0.2).to(DEV)
    mean = (torch.randn(c, generator=g) * 0.1).to(DEV)
    invstd = (torch.rand(c, generator=g) + 0.5).to(DEV)
    dout = torch.randn(b, c, p, generator=g).to(DEV)
    dout_pm = dout.transpose(1, 2).contiguous()
    if compact:
        counts = torch.randint(1, ns + 1, (b, p), generator=g)
        seg = torch.zeros(b, p + 1, dtype=torch.int32)
        seg[:, 1:] = counts.cumsum(1)
        n_act = seg[:, -1].clone().to(DEV)
        arg = (torch.rand(b, c, p, generator=g) * counts[:, None, :]).floor().to(torch.int32).to(DEV)
        cent = torch.zeros(b, e, dtype=torch.int32)
        mult = torch.ones(b, e)
        for bi in range(b):
            cent[bi, :int(seg[bi, -1])] = torch.repeat_interleave(torch.arange(p, dtype=torch.int32), counts[bi])
            mult[bi, seg[bi, :-1].long()] = (ns - counts[bi] + 1).float()
        seg, cent, mult = seg.to(DEV), cent.to(DEV), mult.to(DEV)
    else:
        arg = torch.randint(0, ns, (b, c, p), generator=g, dtype=torch.int32).to(DEV)

    def run(from_pm):
        sums = torch.zeros(2, c, dtype=torch.float64, device=DEV)
        dY = torch.zeros(b, c, e, device=DEV)
        d_cm, acc = dout, 1
        if from_pm:
            d_cm = torch.full((b, c, p), 7.0, device=DEV)
            L.call("sig3d_bn_relu_bwd_top_from_pm", b, c, p, ns, e, L.ptr(dout_pm), L.ptr(arg), L.ptr(y), L.ptr(scale),
                   L.ptr(shift), L.ptr(mean), L.ptr(invstd), L.ptr(seg if compact else None), L.ptr(d_cm), L.ptr(sums[0]),
                   L.ptr(sums[1]), 1, L.stream_ptr())
            assert torch.equal(d_cm, dout)
            acc = 2
        if compact:
            L.call("sig3d_bn_relu_bwd_compact", b, c, e, p, L.ptr(None), L.ptr(d_cm), L.ptr(arg), L.ptr(y), L.ptr(scale),
                   L.ptr(shift), L.ptr(mean), L.ptr(invstd), L.ptr(sums[0]), L.ptr(sums[1]), L.ptr(dY), acc, L.ptr(n_act),
                   L.ptr(mult), L.ptr(cent), L.ptr(seg), L.stream_ptr())
        else:
            L.call("sig3d_bn_relu_bwd", b, c, e, ns, L.ptr(None), L.ptr(d_cm), L.ptr(arg), L.ptr(y), L.ptr(scale), L.ptr(shift),
                   L.ptr(mean), L.ptr(invstd), L.ptr(sums[0]), L.ptr(sums[1]), L.ptr(dY), acc, L.stream_ptr())
        return dY, sums

    dY0, s0 = run(False)
    dY1, s1 = run(True)
    torch.testing.assert_close(s1, s0, rtol=1e-6, atol=1e-6)
    if compact:   # positions beyond n_act are undefined in compact mode
        for bi in range(b):
            n = int(n_act[bi])
            torch.testing.assert_close(dY1[bi, :, :n], dY0[bi, :, :n], rtol=1e-5, atol=1e-6)
    else:
        torch.testing.assert_close(dY1, dY0, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("compact", [False, True])
@pytest.mark.parametrize("c,cout,normalize", [(3, 64, True), (1, 128, False), (5, 64, True)])
def test_first_layer_from_the_raw_scan_matches_group_then_conv(compact, c, cout, normalize):
    """sig3d_sa_first_layer_fwd / _dw (SA1: the column of QueryAndGroup formed in registers from point-major scan
    rows) against sig3d_query_group_fused + a torch 1x1 convolution: pre-activations, batch statistics (weighted by
    the multiplicity on compact lists) and the weight gradient."""
    from situation3d_amd import _lib as L
    from situation3d_amd.pointnet2 import _ext, fused_mlp
    from util import scene
    b, n, m, ns, radius = 2, 3000, 300, 16, 0.35
    g = torch.Generator().manual_seed(c * 10 + cout)
    pts = torch.cat([scene(b, n, seed=4), torch.rand(b, n, c, generator=g)], -1).to(DEV).contiguous()
    xyz = pts[..., :3].contiguous()
    feats = pts[..., 3:].transpose(1, 2).contiguous()
    inds = _ext.furthest_point_sampling(xyz, m)
    new_xyz = torch.gather(xyz, 1, inds.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    idx = _ext.ball_query(new_xyz, xyz, radius, ns)
    w = (torch.randn(cout, 3 + c, generator=g) * 0.3).to(DEV)
    e = m * ns
    grouped = torch.empty(b, 3 + c, m, ns, device=DEV)
    L.call("sig3d_query_group_fused", b, n, m, c, ns, 1, int(normalize), ctypes.c_float(radius), L.ptr(xyz), L.ptr(new_xyz),
           L.ptr(feats), L.ptr(idx), L.ptr(grouped), L.stream_ptr())
    ref_y = torch.einsum("oc,bce->boe", w.double(), grouped.view(b, 3 + c, e).double())
    dY = torch.randn(b, cout, e, generator=g).to(DEV)
    if compact:
        cl = fused_mlp.compact_lists(idx)
        cidx, cent, mult, seg, nact = cl.tensors()
        lists, cptr, nptr, mptr = cidx, L.ptr(cent), L.ptr(nact), L.ptr(mult)
    else:
        lists, cptr, nptr, mptr = idx.view(b, e), L.ptr(None), L.ptr(None), L.ptr(None)
    y = torch.full((b, cout, e), float("nan"), device=DEV)
    st = torch.zeros(2, cout, dtype=torch.float64, device=DEV)
    L.call("sig3d_sa_first_layer_fwd", b, n, m, ns, 3 + c, 3 + c, cout, int(normalize), ctypes.c_float(radius), L.ptr(pts),
           L.ptr(new_xyz), L.ptr(lists), cptr, nptr, mptr, L.ptr(w), L.ptr(y), L.ptr(st[0]), L.ptr(st[1]), 0, L.stream_ptr())
    dW = torch.full((cout, 3 + c), float("nan"), device=DEV)
    L.call("sig3d_sa_first_layer_dw", b, n, m, ns, 3 + c, 3 + c, cout, int(normalize), ctypes.c_float(radius), L.ptr(pts),
           L.ptr(new_xyz), L.ptr(lists), cptr, nptr, L.ptr(dY), L.ptr(dW), 0, L.stream_ptr())
    if not compact:
        torch.testing.assert_close(y.double(), ref_y, rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(st[0], ref_y.sum((0, 2)), rtol=1e-5, atol=1e-3)
        torch.testing.assert_close(st[1], (ref_y ** 2).sum((0, 2)), rtol=1e-5, atol=1e-3)
        ref_dw = torch.einsum("boe,bce->oc", dY.double(), grouped.view(b, 3 + c, e).double())
        torch.testing.assert_close(dW.double(), ref_dw, rtol=1e-4, atol=1e-3)
    else:
        # position u of the compact lists = column (centre_of[u], first occurrence of cidx[u]) of the dense tensor
        ref_dw = torch.zeros(cout, 3 + c, dtype=torch.float64, device=DEV)
        s1 = torch.zeros(cout, dtype=torch.float64, device=DEV)
        s2 = torch.zeros(cout, dtype=torch.float64, device=DEV)
        for bi in range(b):
            na = int(nact[bi])
            cols = torch.empty(na, dtype=torch.long, device=DEV)
            row_idx = idx[bi][cent[bi, :na].long()]                         # (na, ns) lists of the positions' centres
            first = (row_idx == cidx[bi, :na, None]).float().argmax(1)      # first sample holding that neighbour
            cols = cent[bi, :na].long() * ns + first
            ry = ref_y[bi][:, cols]
            torch.testing.assert_close(y[bi, :, :na].double(), ry, rtol=1e-5, atol=1e-5)
            s1 += (ry * mult[bi, :na].double()).sum(1)
            s2 += (ry ** 2 * mult[bi, :na].double()).sum(1)
            ref_dw += dY[bi, :, :na].double() @ grouped.view(b, 3 + c, e)[bi][:, cols].double().t()
        torch.testing.assert_close(st[0], s1, rtol=1e-5, atol=1e-3)
        torch.testing.assert_close(st[1], s2, rtol=1e-5, atol=1e-3)
        torch.testing.assert_close(dW.double(), ref_dw, rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("mode", ["compact", "dense", "dense-eval"])
def test_sa1_reading_the_raw_scan_matches_the_stored_grouped_tensor(mode, monkeypatch):
    """PointnetSAModuleVotes on a (B, N, 6) scan: the first SharedMLP layer formed from point-major rows
    (fused_mlp.attach_scan / first_layer_scan; no colour transpose, no grouped tensor) against the stored form
    (fused_mlp.FIRST_L0 = False): pooled features, every parameter gradient, BatchNorm running statistics."""
    import copy
    from situation3d_amd.pointnet2 import fused_mlp
    from situation3d_amd.pointnet2.pointnet2_modules import PointnetSAModuleVotes
    from util import scene
    torch.manual_seed(3)
    b, n = 2, 9000
    pc = torch.cat([scene(b, n, seed=5), torch.rand(b, n, 3)], -1).to(DEV).contiguous()
    xyz = pc[..., :3].contiguous()
    monkeypatch.setattr(fused_mlp, "COMPACT", mode == "compact")
    sa = PointnetSAModuleVotes(npoint=1024, radius=0.25 if mode == "compact" else 0.6, nsample=64, mlp=[3, 64, 64, 128],
                               use_xyz=True, normalize_xyz=True).to(DEV)
    ref = copy.deepcopy(sa)
    if mode == "dense-eval":
        sa.eval(); ref.eval()
    G = torch.randn(b, 128, 1024, device=DEV)
    calls = []
    orig = fused_mlp._lib.call
    monkeypatch.setattr(fused_mlp._lib, "call", lambda name, *a: (calls.append(name), orig(name, *a))[1])

    def run(m, scan):
        monkeypatch.setattr(fused_mlp, "FIRST_L0", scan)
        monkeypatch.setattr(fused_mlp, "FIRST_L0_DENSE", scan)     # dense lists take the scan path on request only
        f = fused_mlp.attach_scan(pc[..., 3:].transpose(1, 2), pc)
        with torch.set_grad_enabled(mode != "dense-eval"):
            _, out, inds = m(xyz, f)
            if mode != "dense-eval":
                (out * G).sum().backward()
        return out.detach(), inds

    o1, i1 = run(sa, True)
    assert "sig3d_sa_first_layer_fwd" in calls and not any(nm.startswith("sig3d_query_group") for nm in calls)
    if mode != "dense-eval":
        assert "sig3d_sa_first_layer_dw" in calls
    calls.clear()
    o2, i2 = run(ref, False)
    assert "sig3d_sa_first_layer_fwd" not in calls and any(nm.startswith("sig3d_query_group") for nm in calls)
    assert torch.equal(i1, i2)
    _close(o1, o2, "pooled features", tol=2e-5)
    if mode != "dense-eval":
        for (n1, p1), (_, p2) in zip(sa.named_parameters(), ref.named_parameters()):
            rel = float((p1.grad - p2.grad).norm() / p2.grad.norm().clamp_min(1e-12))
            assert rel < 1e-3, (n1, rel)      # a max-pool winner that leads by an ulp may change (see the gather test)
        for (n1, b1), (_, b2) in zip(sa.named_buffers(), ref.named_buffers()):
            _close(b1.float(), b2.float(), n1, tol=1e-5)


@pytest.mark.parametrize("b,cin,cout,e,compact,prologue", [(8, 128, 128, 4096, False, True), (8, 259, 128, 4096, False, False),
                                                          (8, 131, 128, 8192, False, False), (8, 128, 256, 8192, False, True),
                                                          (8, 64, 128, 20000, True, True), (3, 64, 64, 1000, True, True),
                                                          (2, 35, 64, 132, False, True), (8, 128, 128, 2048, True, True),
                                                          (1, 6, 32, 36, True, False)])
def test_streaming_weight_gradient_matches_the_row_per_lane_kernel_and_float64(b, cin, cout, e, compact, prologue):
    """sig3d_mlp_layer_dw_stream (k-streaming split product, slabs folded in a fixed order) against sig3d_mlp_layer_dw /
    _compact (f32 atomics) and a float64 einsum: dense rows, compact rows with ragged live counts (one of them 0 .. a
    few positions), with and without the previous layer's BatchNorm + ReLU on load; and it is deterministic."""
    from situation3d_amd import _lib as L
    g = torch.Generator().manual_seed(cin * 7 + cout + e)
    dY = torch.randn(b, cout, e, generator=g).to(DEV)
    x = torch.randn(b, cin, e, generator=g).to(DEV)
    ps = (torch.rand(cin, generator=g) + 0.5).to(DEV) if prologue else None
    pb = torch.randn(cin, generator=g).to(DEV) if prologue else None
    n_act = None
    if compact:
        live = [e, max(e // 3, 1), 5, 0, e - 1, e // 2, 33, e][:b]
        n_act = torch.tensor(live, dtype=torch.int32, device=DEV)
    st = L.stream_ptr(DEV)
    work = torch.empty(max(int(L.load().sig3d_mlp_layer_dw_stream_work_floats(b, cin, cout, e)), 4), device=DEV)
    outs = []
    with torch.cuda.device(DEV):
        for _ in range(2):
            dW = torch.full((cout, cin), float("nan"), device=DEV)
            L.call("sig3d_mlp_layer_dw_stream", b, cin, cout, e, L.ptr(dY), L.ptr(x), L.ptr(ps), L.ptr(pb), L.ptr(n_act),
                   L.ptr(dW), L.ptr(work), st)
            outs.append(dW)
        old = torch.empty(cout, cin, device=DEV)
        if compact:
            L.call("sig3d_mlp_layer_dw_compact", b, cin, cout, e, L.ptr(dY), L.ptr(x), L.ptr(ps), L.ptr(pb), L.ptr(old), 0,
                   L.ptr(n_act), st)
        else:
            L.call("sig3d_mlp_layer_dw", b, cin, cout, e, L.ptr(dY), L.ptr(x), L.ptr(ps), L.ptr(pb), L.ptr(old), 0, st)
    assert torch.equal(outs[0], outs[1])
    # the same product without its fold + the fold of several results in one launch (what a SharedMLP stack's backward
    # pass does): bit for bit the folded product, whatever else shares the launch
    with torch.cuda.device(DEV):
        dW2 = torch.full((cout, cin), float("nan"), device=DEV)
        work2 = torch.empty_like(work)
        L.call("sig3d_mlp_layer_dw_stream_nofold", b, cin, cout, e, L.ptr(dY), L.ptr(x), L.ptr(ps), L.ptr(pb), L.ptr(n_act),
               L.ptr(dW2), L.ptr(work2), st)
        n_work = int(L.load().sig3d_mlp_layer_dw_stream_work_floats(b, cin, cout, e))
        slab = (cout * cin + 3) // 4 * 4
        other = torch.ones(40, device=DEV)
        other_slabs = torch.arange(3 * 40, dtype=torch.float32, device=DEV)
        src64 = torch.randn(700, dtype=torch.float64, device=DEV)
        dst32 = torch.full((700,), float("nan"), device=DEV)
        L.sum_slabs_multi(torch.device(DEV), [(other, other_slabs, 40, 40, 3), (dW2, work2, cout * cin, slab, n_work // slab)],
                          convert=(src64, dst32))
    assert torch.equal(dW2, outs[0])
    assert torch.equal(other, 1 + other_slabs.view(3, 40).sum(0))
    assert torch.equal(dst32, src64.float())            # the conversion riding along
    a = x.double()
    if prologue:
        a = torch.relu(a * ps.double()[None, :, None] + pb.double()[None, :, None])
    d = dY.double()
    if compact:
        mask = (torch.arange(e, device=DEV)[None, :] < n_act[:, None]).double()[:, None, :]
        a, d = a * mask, d * mask
    ref = torch.einsum("boe,bce->oc", d, a)
    scale = max(1.0, float(ref.abs().max()))
    assert float((outs[0].double() - ref).abs().max()) < 2e-5 * scale
    assert float((old.double() - ref).abs().max()) < 1e-4 * scale


@pytest.mark.parametrize("b,cin,cout,e,live,prologue", [(8, 64, 128, 131072, 14400, True), (8, 64, 64, 131072, 47000, True),
                                                        (8, 128, 256, 32768, 1300, True), (8, 128, 128, 32768, 5400, True),
                                                        (3, 96, 128, 4096, 700, False), (2, 35, 64, 132, 100, True),
                                                        (2, 64, 40, 1024, 300, True)])
def test_weight_and_input_gradient_in_one_launch_match_the_two_launches(b, cin, cout, e, live, prologue):
    """sig3d_mlp_layer_dw_dx (a compact layer's weight gradient and input gradient as two workgroup ranges of one
    launch) against sig3d_mlp_layer_dw_stream_nofold + sig3d_mlp_layer_dx: bit for bit, ragged live counts, shapes that
    fall back to the two launches included (odd channel counts)."""
    from situation3d_amd import _lib as L
    g = torch.Generator().manual_seed(cin + cout + e)
    dY = torch.randn(b, cout, e, generator=g).to(DEV)
    x = torch.randn(b, cin, e, generator=g).to(DEV)
    w = torch.randn(cout, cin, generator=g).to(DEV)
    ps = (torch.rand(cin, generator=g) + 0.5).to(DEV) if prologue else None
    pb = torch.randn(cin, generator=g).to(DEV) if prologue else None
    lives = [live, max(live // 3, 1), 5, 0, live - 1, live // 2, 33, live][:b]
    n_act = torch.tensor(lives, dtype=torch.int32, device=DEV)
    st = L.stream_ptr(DEV)
    n_work = int(L.load().sig3d_mlp_layer_dw_stream_work_floats(b, cin, cout, e))
    res = []
    with torch.cuda.device(DEV):
        for one in (False, True):
            dW = torch.full((cout, cin), float("nan"), device=DEV)
            dA = torch.zeros(b, cin, e, device=DEV)          # (columns past a sample's live count are not written)
            work = torch.full((max(n_work, 4),), float("nan"), device=DEV)
            if one:
                L.call("sig3d_mlp_layer_dw_dx", b, cin, cout, e, L.ptr(dY), L.ptr(x), L.ptr(ps), L.ptr(pb), L.ptr(n_act),
                       L.ptr(w), L.ptr(dW), L.ptr(work), L.ptr(dA), st)
            else:
                L.call("sig3d_mlp_layer_dw_stream_nofold", b, cin, cout, e, L.ptr(dY), L.ptr(x), L.ptr(ps), L.ptr(pb),
                       L.ptr(n_act), L.ptr(dW), L.ptr(work), st)
                L.call("sig3d_mlp_layer_dx", b, cin, cout, e, L.ptr(dY), L.ptr(w), L.ptr(dA), L.ptr(n_act), st)
            res.append((dW, work[:n_work].clone(), dA))
    for a, c, name in zip(res[0], res[1], ("dW", "slabs", "dA")):
        assert torch.equal(torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(c, nan=-7.0)), name
    ref = torch.einsum("oc,boe->bce", w.double(), dY.double())
    for i, n in enumerate(lives):
        assert float((res[1][2][i, :, :n].double() - ref[i, :, :n]).abs().max() if n else 0.0) < 1e-3
