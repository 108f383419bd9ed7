"""csrc/small_mlp.hip: the small dense layers around the Q-Former as single launches, against the torch modules they
stand for (same parameters): outputs and every gradient."""
import copy

import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("rows_shape,cin,hid,cout,x_grad", [((8, 256), 3, 128, 256, False), ((8, 256), 2, 128, 256, True),
                                                          ((37,), 3, 128, 256, True), ((3, 5), 4, 64, 96, False)])
def test_pos_embed_add_matches_the_sequential(rows_shape, cin, hid, cout, x_grad):
    """tok_feat + Linear(GELU(Linear(pos))) (sqa_module.py:274-278, :319-321) as one forward and two backward launches."""
    from situation3d_amd.model import build_pos_embed
    from situation3d_amd.small_mlp import _covered, pos_embed_add
    torch.manual_seed(cin * 7 + hid)
    seq = nn.Sequential(nn.Linear(cin, hid), nn.GELU(), nn.Linear(hid, cout)).to(DEV)
    ref = copy.deepcopy(seq)
    if (cin, hid, cout) == (3, 128, 256):
        assert [tuple(p.shape) for p in build_pos_embed(3, 256).parameters()] == [tuple(p.shape) for p in seq.parameters()]
    x = (torch.randn(*rows_shape, cin, device=DEV) * 2).requires_grad_(x_grad)
    xr = x.detach().clone().requires_grad_(x_grad)
    res = torch.randn(*rows_shape, cout, device=DEV, requires_grad=True)
    rr = res.detach().clone().requires_grad_(True)
    G = torch.randn(*rows_shape, cout, device=DEV)
    assert _covered(seq, x, res)
    out = pos_embed_add(seq, x, res)
    want = rr + ref(xr)
    # 1e-5 relative to the tensor's scale: f32 dot products of 128 terms in a different order
    torch.testing.assert_close(out, want, rtol=1e-5, atol=1e-5 * float(want.detach().abs().max()))
    (out * G).sum().backward()
    (want * G).sum().backward()
    torch.testing.assert_close(res.grad, rr.grad, rtol=0, atol=0)
    for (n, p), q in zip(seq.named_parameters(), ref.parameters()):
        torch.testing.assert_close(p.grad, q.grad, rtol=1e-4, atol=1e-5 * max(1.0, float(q.grad.abs().max())), msg=lambda m: n + ": " + m)
    if x_grad:
        torch.testing.assert_close(x.grad, xr.grad, rtol=1e-4, atol=1e-5 * max(1.0, float(xr.grad.abs().max())))


def test_pos_embed_add_falls_back_for_what_the_kernels_do_not_cover():
    from situation3d_amd.small_mlp import _covered, pos_embed_add
    seq = nn.Sequential(nn.Linear(3, 128), nn.GELU(approximate="tanh"), nn.Linear(128, 256)).to(DEV)
    x, res = torch.randn(4, 3, device=DEV), torch.randn(4, 256, device=DEV)
    assert not _covered(seq, x, res)
    torch.testing.assert_close(pos_embed_add(seq, x, res), res + seq(x))
    wide = nn.Sequential(nn.Linear(3, 256), nn.GELU(), nn.Linear(256, 256)).to(DEV)
    assert not _covered(wide, x, res)
    cpu = nn.Sequential(nn.Linear(3, 128), nn.GELU(), nn.Linear(128, 256))
    assert not _covered(cpu, x.cpu(), res.cpu())
    torch.testing.assert_close(pos_embed_add(cpu, x.cpu(), res.cpu()), res.cpu() + cpu(x.cpu()))


def test_situational_transform_folded_into_the_positional_mlp_is_bit_identical():
    """sig3d_pos_mlp_fwd_posed forms x = R(q)^T (p - t) inside the positional MLP's launch (temp.py:86-97 +
    sqa_module.py:274-278, 319-321 as one kernel): the re-encoded positions and the tokens must equal the two-launch
    path bit for bit, forward and (parameter) gradients; with a gradient wanted for the pose the fold steps aside."""
    import torch
    from situation3d_amd.model import build_pos_embed
    from situation3d_amd.situational import situational_transform
    from situation3d_amd.small_mlp import pos_embed_add, posed_pos_embed_add
    dev = "cuda:0"
    g = torch.Generator().manual_seed(3)
    b, t = 8, 256
    pose = torch.randn(b, 7, generator=g).to(dev)
    pts = (torch.rand(b, t, 3, generator=g) * 8).to(dev)
    feat = torch.randn(b, t, 256, generator=g).to(dev).requires_grad_(True)
    mlp = build_pos_embed(3, 256).to(dev)
    sit = situational_transform(pose, pts, inverse=True)
    ref = pos_embed_add(mlp, sit, feat)
    G = torch.randn(b, t, 256, generator=g).to(dev)
    (ref * G).sum().backward()
    ref_grads = [p.grad.clone() for p in mlp.parameters()] + [feat.grad.clone()]
    mlp.zero_grad(set_to_none=True)
    feat.grad = None
    fused = posed_pos_embed_add(mlp, pose, pts, feat, inverse=True)
    assert fused is not None
    tokens, sit2 = fused
    assert torch.equal(sit2, sit) and torch.equal(tokens, ref)
    (tokens * G).sum().backward()
    for a, r in zip([p.grad for p in mlp.parameters()] + [feat.grad], ref_grads):
        assert float((a - r).norm() / r.norm()) < 1e-5              # float atomics: the order of the sums varies
    assert posed_pos_embed_add(mlp, pose.clone().requires_grad_(True), pts, feat) is None
    assert posed_pos_embed_add(build_pos_embed(2, 256).to(dev), pose, pts, feat) is None
