"""CPU suite, part 1: the oracle against the committed golden vectors and against independent
properties.  The goldens were produced by the REFERENCE's own Python code imported in the build
container (tests/golden/make_golden.py); nothing here reads /root/reference.
"""
import os

import numpy as np
import pytest
import torch

from util import scene

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return {k: v for k, v in np.load(os.path.join(GOLD, name), allow_pickle=False).items()}


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


# ---- oracle regression pins ------------------------------------------------------------------
def test_oracle_ops_match_committed_vectors(oracle):
    g = _load("pointnet2_ops.npz")
    xyz = _t(g["fps.xyz"])
    assert torch.equal(oracle.furthest_point_sampling(xyz, 300), _t(g["fps.idx"]))
    assert torch.equal(oracle.ball_query(_t(g["bq.new_xyz"]), xyz, 0.5, 16), _t(g["bq.idx"]))
    d2, i3 = oracle.three_nn(xyz[:, :200].contiguous(), _t(g["bq.new_xyz"]))
    assert torch.equal(i3, _t(g["nn.idx"])) and torch.equal(d2, _t(g["nn.dist2"]))


def test_reference_three_interpolate_vector(oracle):
    """The reference's only op test (lib/pointnet2/pointnet2_test.py:18-30): idx [[0,1,2],[1,2,3]],
    weights [[1,1,1],[2,2,2]]; its gradcheck (atol=rtol=1e-1) is reproduced analytically."""
    g = _load("pointnet2_ops.npz")
    pts, idx, w = _t(g["ti.points"]), _t(g["ti.idx"]), _t(g["ti.weight"])
    out = oracle.three_interpolate(pts, idx, w)
    assert torch.equal(out, _t(g["ti.out"]))
    exp = torch.stack([pts[0, :, 0] + pts[0, :, 1] + pts[0, :, 2],
                       2 * (pts[0, :, 1] + pts[0, :, 2] + pts[0, :, 3])], -1)[None]
    torch.testing.assert_close(out, exp, rtol=1e-6, atol=1e-6)
    go = torch.ones(1, 2, 2)
    grad = oracle.three_interpolate_grad(go, idx, w, 4)
    torch.testing.assert_close(grad, torch.tensor([[[1., 3., 3., 2.]] * 2]), rtol=1e-1, atol=1e-1)


# ---- independent properties -----------------------------------------------------
def test_fps_properties(oracle):
    xyz = scene(2, 2000, seed=3)
    idx = oracle.furthest_point_sampling(xyz, 64).long()
    assert (idx[:, 0] == 0).all()
    for b in range(2):
        sel = idx[b]
        assert sel.unique().numel() == 64
        # maximin: every pick maximises the distance to the already-picked set
        for j in (1, 5, 63):
            d = torch.cdist(xyz[b], xyz[b, sel[:j]]).min(1).values
            assert torch.isclose(d[sel[j]], d.max(), rtol=1e-5)


def test_fps_tie_break_is_reference_order(oracle):
    """Four coincident far points: the winner is fixed by the reference's strided scan + tree
    (sampling_gpu.cu:95-168): minimal key (bitrev9(k mod 512), k div 512)."""
    n = 1500
    xyz = torch.zeros(1, n, 3) + 0.5
    ks = [700, 5, 1029, 260]
    xyz[0, ks] = torch.tensor([5.0, 5.0, 1.0])
    idx = oracle.furthest_point_sampling(xyz, 2)

    def key(k):
        return (int("{:09b}".format(k % 512)[::-1], 2), k // 512)

    assert idx[0, 1].item() == min(ks, key=key)


def test_ball_query_properties(oracle):
    xyz = scene(2, 3000, seed=5)
    new_xyz = xyz[:, :100].contiguous()
    r, ns = 0.6, 24
    idx = oracle.ball_query(new_xyz, xyz, r, ns).long()
    d2 = torch.cdist(new_xyz.double(), xyz.double()) ** 2
    for b in range(2):
        for j in range(0, 100, 7):
            inside = (d2[b, j] < r * r - 1e-9).nonzero().flatten()
            row = idx[b, j]
            k = min(ns, inside.numel())
            assert torch.equal(row[:k], inside[:k])           # first ns in index order
            assert (row[k:] == row[0]).all()                  # padded with the first hit


def test_three_nn_matches_topk(oracle):
    unknown, known = scene(2, 300, seed=6), scene(2, 90, seed=7)
    d2, i3 = oracle.three_nn(unknown, known)
    diff = unknown.double()[:, :, None, :] - known.double()[:, None, :, :]
    ref = (diff * diff).sum(-1).topk(3, largest=False)
    assert torch.equal(i3.long(), ref.indices)
    torch.testing.assert_close(d2.double(), ref.values, rtol=1e-5, atol=1e-6)


def test_group_gather_match_torch_and_autograd(oracle):
    g = torch.Generator().manual_seed(8)
    pts = torch.rand(2, 5, 50, generator=g, dtype=torch.float32)
    idx = torch.randint(0, 50, (2, 7, 4), generator=g, dtype=torch.int32)
    out = oracle.group_points(pts, idx)
    ref = torch.gather(pts[:, :, None, :].expand(2, 5, 7, 50), 3, idx.long()[:, None].expand(2, 5, 7, 4))
    assert torch.equal(out, ref)
    go = torch.rand(2, 5, 7, 4, generator=g)
    p = pts.clone().requires_grad_(True)
    (torch.gather(p[:, :, None, :].expand(2, 5, 7, 50), 3, idx.long()[:, None].expand(2, 5, 7, 4)) * go).sum().backward()
    torch.testing.assert_close(oracle.group_points_grad(go, idx, 50), p.grad, rtol=1e-5, atol=1e-6)


def test_situational_transform_known_answers(oracle):
    from scipy.spatial.transform import Rotation
    pts = scene(3, 40, seed=9)
    ident = torch.tensor([[1., 2., 3., 0., 0., 0., 1.]]).repeat(3, 1)
    torch.testing.assert_close(oracle.situational_transform(ident, pts), pts + ident[:, None, :3])
    s = 2 ** -0.5
    rotz = torch.tensor([[0., 0., 0., 0., 0., s, s]]).repeat(3, 1)  # +90 deg about z
    out = oracle.situational_transform(rotz, pts)
    torch.testing.assert_close(out, torch.stack([-pts[..., 1], pts[..., 0], pts[..., 2]], -1),
                               rtol=1e-6, atol=1e-6)
    q = Rotation.random(3, random_state=1).as_quat().astype(np.float32)  # xyzw, unit
    pose = torch.cat([torch.zeros(3, 3), _t(q)], 1).contiguous()
    R = _t(Rotation.from_quat(q).as_matrix().astype(np.float32))
    torch.testing.assert_close(oracle.pose_to_matrix(pose)[:, :3, :3], R, rtol=1e-5, atol=1e-6)


# ---- Q-Former oracle vs the reference's Qformer.py (golden) ------------------------------------
def _qformer_setup():
    from oracle import qformer_ref
    g = _load("qformer_small.npz")
    c = g["config"]
    cfg = dict(vocab_size=int(c[0]), hidden_size=int(c[1]), num_hidden_layers=int(c[2]),
               num_attention_heads=int(c[3]), intermediate_size=int(c[4]),
               max_position_embeddings=int(c[5]), encoder_width=int(c[6]),
               cross_attention_freq=int(c[7]), query_length=int(c[8]), add_cross_attention=True,
               layer_norm_eps=1e-12)
    sd = {k[len("state."):]: _t(v) for k, v in g.items() if k.startswith("state.")}
    return qformer_ref, g, cfg, sd


def test_qformer_oracle_matches_reference_queries_only():
    Q, g, cfg, sd = _qformer_setup()
    sd = {k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in sd.items()}
    query = _t(g["query_embeds"]).requires_grad_(True)
    enc = _t(g["encoder_hidden_states"]).requires_grad_(True)
    out, states = Q.bert_model(sd, cfg, query_embeds=query, encoder_hidden_states=enc,
                               encoder_attention_mask=_t(g["encoder_attention_mask"]),
                               return_all=True)
    # fp32 activations within 1e-4 (north star); same torch CPU ops => observed ~1e-6
    torch.testing.assert_close(out, _t(g["last_hidden_state"]), rtol=1e-4, atol=1e-4)
    for i, s in enumerate(states):
        torch.testing.assert_close(s, _t(g["hidden_states.%d" % i]), rtol=1e-4, atol=1e-4)
    (out * _t(g["G"])).sum().backward()
    torch.testing.assert_close(query.grad, _t(g["grad_query_embeds"]), rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(enc.grad, _t(g["grad_encoder_hidden_states"]), rtol=1e-3, atol=1e-4)
    for k, v in g.items():
        if k.startswith("grad.") and not k.startswith("grad_"):
            name = k[len("grad."):]
            torch.testing.assert_close(sd[name].grad, _t(v), rtol=1e-3, atol=1e-4, msg=name)


def test_qformer_oracle_matches_reference_with_text():
    Q, g, cfg, sd = _qformer_setup()
    out = Q.bert_model(sd, cfg, query_embeds=_t(g["query_embeds"]), input_ids=_t(g["t.input_ids"]),
                       attention_mask=_t(g["t.attention_mask"]),
                       encoder_hidden_states=_t(g["encoder_hidden_states"]),
                       encoder_attention_mask=_t(g["encoder_attention_mask"]))
    torch.testing.assert_close(out, _t(g["t.last_hidden_state"]), rtol=1e-4, atol=1e-4)
