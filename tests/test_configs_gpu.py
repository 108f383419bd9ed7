"""BASELINE.json configs at their full sizes against the oracle:
  config 2  SQA3D forward-only, 40 000 points, 32 queries + 20 question tokens, B = 4, eval mode
            (fused-inference SharedMLP path, running BatchNorm statistics) -- every output of
            forward(data_dict) against the same weights pushed through oracle/pointnet2_oracle.c +
            oracle/qformer_ref.py (bench.oracle_forward);
  config 3  the B = 8 train-mode forward + loss the bench times (dropout off), same comparison;
  config 5  the 3D-LLM shape: Blip2PointQFormer, B = 4, d_enc 1408, Nk = 80 000 (forward; key-split
            attention) and the reference's Nk = 5000 (forward + backward) against qformer_ref + blip2_ref;
  full-size Q-Former: 768 wide / 12 layers / 12 heads, B = 8, 32 + 20 tokens, Nk = 256, forward + backward
            against qformer_ref (which the reference's own Qformer.py pins: tests/test_oracle_golden.py).
fp32 activations within 1e-4 (north star), relative to the tensor's scale; tolerances written at each assert.
"""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
# gradient bounds = ~2x the worst value observed on MI355X (SIG3D_TEST_REPORT=1 pytest -s prints them)
ENC_GRAD_TOL = 8e-3       # point-encoder weights: 1 M positions x 8 scenes through BatchNorm batch statistics and
                          # max-pool winners, float atomics vs MKL order; observed 2.4e-3 .. 3.7e-3 from run to run
ENC_GRAD_F64_TOL = 6e-3   # the same gradients against the FLOAT64 run of the same graph (same indices), relative L2 per
                          # parameter: observed worst 2.8e-3 for the HIP path, 4.1e-3 for the float32 oracle (MKL) -- the
                          # 8e-3 above is mostly the ORACLE's distance from the truth, not the kernels' 
QF_GRAD_TOL = 2e-5        # Q-Former / head parameters and inputs: observed 1.6e-6 .. 5.2e-6 (was 1e-3 until round 3)


def _rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


_OBSERVED = {}


def _within(what, r, bound):
    """assert r < bound; the worst value seen per class of check is printed by `pytest -s` when SIG3D_TEST_REPORT is
    set (how the bounds below were chosen: ~2x the observed worst)."""
    import os
    key = what.split(":")[0]
    _OBSERVED[key] = max(_OBSERVED.get(key, 0.0), r)
    if os.environ.get("SIG3D_TEST_REPORT"):
        print("[observed] %-40s worst %.3g (bound %.3g)" % (key, _OBSERVED[key], bound))
    assert r < bound, "%s: relative max error %.3g (bound %.3g)" % (what, r, bound)


def _is_key_bias(name):
    """softmax(q (k + b)^T) does not depend on a key bias b (it shifts every score of a row by q.b), so
    d loss / d key.bias is EXACTLY zero; what both implementations return is rounding noise (~1e-9) that
    cannot be compared relatively."""
    return name.endswith("self.key.bias")


def _sig3d(train, batch_size, seed):
    import bench
    from situation3d_amd.model import SIG3DQFormer
    torch.manual_seed(seed)
    model = SIG3DQFormer(num_answers=bench.NUM_ANSWERS)
    # non-trivial BatchNorm running statistics / affine so that eval mode is exercised
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
    model = model.train() if train else model.eval()
    bench._without_dropout(model)
    cpu_model = copy.deepcopy(model)
    batch = bench.synthetic_batch(batch_size, bench.N_POINTS, seed + 2, "cpu")
    return bench, model.to(DEV), cpu_model, batch


def _compare_outputs(out, ref):
    # indices first: the token positions are FPS picks of FPS picks -- bit-exact or everything below is noise
    assert torch.equal(out["scene_positions"].cpu(), ref["scene_positions"]), "SA4 centres differ"
    for key, tol in (("att_feat_pre", 1e-4), ("att_feat_ori", 1e-4), ("answer_scores", 1e-4), ("aux_scores", 1e-4)):
        r = _rel(out[key].detach().cpu(), ref[key].detach())
        assert r < tol, "%s: relative max error %.3g" % (key, r)     # 1e-4: north-star fp32 bar
    r = abs(float(out["loss"]) - float(ref["loss"])) / abs(float(ref["loss"]))
    assert r < 1e-4, "loss: relative error %.3g" % r


class _Ext64:
    """The nine ops for a float64 run of the oracle pipeline: the index ops go through the C oracle on the float32
    coordinates (identical FPS picks and neighbour lists), the data movement happens in the tensor's own dtype."""

    def __getattr__(self, name):          # everything not redefined below (three_interpolate ...: unused here)
        from oracle import pointnet2_ref
        return getattr(pointnet2_ref, name)

    def furthest_point_sampling(self, xyz, m):
        from oracle import pointnet2_ref
        return pointnet2_ref.furthest_point_sampling(xyz.float().contiguous(), m)

    def ball_query(self, new_xyz, xyz, radius, nsample):
        from oracle import pointnet2_ref
        return pointnet2_ref.ball_query(new_xyz.float().contiguous(), xyz.float().contiguous(), radius, nsample)

    def gather_points(self, p, idx):
        return torch.gather(p, 2, idx.long().unsqueeze(1).expand(-1, p.shape[1], -1))

    def gather_points_grad(self, g, idx, n):
        return torch.zeros(g.shape[0], g.shape[1], n, dtype=g.dtype).scatter_add_(
            2, idx.long().unsqueeze(1).expand(-1, g.shape[1], -1), g)

    def group_points(self, p, idx):
        b, c, _ = p.shape
        _, m, s = idx.shape
        return torch.gather(p, 2, idx.long().view(b, 1, m * s).expand(-1, c, -1)).view(b, c, m, s)

    def group_points_grad(self, g, idx, n):
        b, c, m, s = g.shape
        return torch.zeros(b, c, n, dtype=g.dtype).scatter_add_(
            2, idx.long().view(b, 1, m * s).expand(-1, c, -1), g.reshape(b, c, m * s))

    def pose_to_matrix(self, pose):
        from oracle import pointnet2_ref
        return pointnet2_ref.pose_to_matrix(pose.float().contiguous()).to(pose.dtype)


def _to64(x):
    if isinstance(x, dict):
        return {k: _to64(v) for k, v in x.items()}
    return x.double() if torch.is_tensor(x) and x.is_floating_point() else x


def test_config2_sqa3d_forward_only_b4_40k_eval_matches_oracle():
    bench, model, cpu_model, batch = _sig3d(train=False, batch_size=4, seed=21)
    from situation3d_amd.trainer import get_loss
    with torch.no_grad():
        ref = bench.oracle_forward(cpu_model, batch)
        out = model(bench.to_device(batch, DEV))
        get_loss(out)
    _compare_outputs(out, ref)


def test_config3_sqa3d_train_forward_b8_40k_matches_oracle():
    """Train mode (BatchNorm batch statistics over 8 scenes; compact set abstraction where the lists are
    mostly padding), forward + loss, and the gradient of the loss w.r.t. a few parameters."""
    bench, model, cpu_model, batch = _sig3d(train=True, batch_size=8, seed=31)
    from situation3d_amd.trainer import get_loss
    ref = bench.oracle_forward(cpu_model, batch, backward=True)
    out = model(bench.to_device(batch, DEV))
    loss, out = get_loss(out)
    loss.backward()
    _compare_outputs(out, ref)
    cpu_params = dict(cpu_model.named_parameters())
    checked = 0
    for name, p in model.named_parameters():
        if name in ("query_tokens", "answer_cls.3.weight", "Qformer.bert.encoder.layer.0.crossattention.self.key.weight",
                    "Qformer.bert.encoder.layer.11.output_query.dense.weight", "pos_embed.0.weight",
                    "encoder.sa4.mlp_module.layer2.conv.weight", "encoder.sa1.mlp_module.layer0.conv.weight",
                    "encoder.sa2.mlp_module.layer1.bn.bn.weight"):
            r = _rel(p.grad.cpu(), cpu_params[name].grad)
            # both sides are f32; a weight gradient of SA1 sums 1 M positions x 8 scenes through BatchNorm's
            # batch statistics in a different order (MFMA tiles + float atomics vs MKL)
            enc = name.startswith("encoder.")
            _within(("config3 encoder grad: " if enc else "config3 head/Q-Former grad: ") + name, r, ENC_GRAD_TOL if enc else QF_GRAD_TOL)
            checked += 1
    assert checked == 8
    # ---- float64 adjudication.  The encoder's weight gradients pass through BatchNorm batch
    # statistics over 1 M positions and through max-pool winners, which a rounding difference can flip: two float32
    # implementations differ by 0.2-1 % there, and so does EACH of them from float64.  The same graph in float64 (same
    # indices) says who is right: the HIP path must be as close to it as the float32 oracle is.
    model64 = copy.deepcopy(cpu_model).double()
    model64.zero_grad(set_to_none=True)
    bench.oracle_forward(model64, _to64(batch), backward=True, ext=_Ext64())
    p64 = dict(model64.named_parameters())
    hip_l2, mkl_l2, n_enc = 0.0, 0.0, 0
    for name, p in model.named_parameters():
        if not name.startswith("encoder.") or p.grad is None:
            continue
        g64 = p64[name].grad
        scale = float(g64.norm())
        e_hip = float((p.grad.cpu().double() - g64).norm()) / scale
        e_mkl = float((cpu_params[name].grad.double() - g64).norm()) / scale
        hip_l2, mkl_l2, n_enc = max(hip_l2, e_hip), max(mkl_l2, e_mkl), n_enc + 1
        _within("config3 encoder grad vs float64 (L2): " + name, e_hip, ENC_GRAD_F64_TOL)
    assert n_enc >= 36
    _within("config3 encoder grads vs float64, worst L2 of the HIP path", hip_l2, ENC_GRAD_F64_TOL)
    _within("config3 encoder grads vs float64, worst L2 of the float32 ORACLE", mkl_l2, 1.0)     # reported, not bounded
    # as close to float64 as the other float32 implementation is (worst parameter against worst parameter)
    assert hip_l2 <= 2.0 * mkl_l2 + 1e-4, "HIP path %.3g vs float32 oracle %.3g from float64" % (hip_l2, mkl_l2)


def _blip2(nk, b, seed):
    from oracle import blip2_ref, qformer_ref
    from situation3d_amd.blip2 import Blip2PointQFormer
    torch.manual_seed(seed)
    model = Blip2PointQFormer().eval()       # 32 queries, d_enc 1408, bert-base Q-Former, t5_proj 768 -> 2048
    g = torch.Generator().manual_seed(seed + 1)
    samples = {"pc_feat": torch.randn(b, nk, 1408, generator=g),
               "pc": torch.randint(0, 256, (b, nk, 3), generator=g).float()}

    def reference(requires_grad):
        feat = samples["pc_feat"].clone().requires_grad_(requires_grad)
        # blip2_ref is the loop-for-loop restatement of blip2_t5.py:106-118 (it allocates a second (B,Nk,1408) tensor)
        enc = blip2_ref.add_position_embedding(feat, samples["pc"], model.pos_embedding, 0.01)
        sd = dict(model.Qformer.bert.state_dict())
        sd.update(dict(model.Qformer.bert.named_parameters()))
        c = model.Qformer.config
        cfg = dict(num_hidden_layers=c.num_hidden_layers, num_attention_heads=c.num_attention_heads,
                   layer_norm_eps=c.layer_norm_eps, add_cross_attention=True, cross_attention_freq=c.cross_attention_freq)
        hidden = qformer_ref.bert_model(sd, cfg, query_embeds=model.query_tokens.expand(b, -1, -1),
                                        encoder_hidden_states=enc, encoder_attention_mask=torch.ones(b, nk))
        return feat, hidden, model.t5_proj(hidden)

    return model, samples, reference


def test_config5_blip2_shape_nk80000_forward_matches_oracle():
    """B = 4, Nk = 80 000 point tokens of width 1408 (1.8 GB of features): key-split attention forward."""
    model, samples, reference = _blip2(80000, 4, seed=51)
    with torch.no_grad():
        _, hidden, t5 = reference(False)
        gpu = copy.deepcopy(model).to(DEV)
        out = gpu({k: v.to(DEV) for k, v in samples.items()})
    assert _rel(out["query_output"].cpu(), hidden) < 1e-4          # north-star fp32 bar
    assert _rel(out["inputs_t5"].cpu(), t5) < 1e-4


def test_config5_blip2_reference_shape_nk5000_forward_backward_matches_oracle():
    """The reference's own token count (threedvqa_datasets.py:72-79), B = 2, forward + backward incl. the
    gradient w.r.t. the point features and the cross-attention key / value weights of the six cross layers."""
    model, samples, reference = _blip2(5000, 2, seed=52)
    G = torch.randn(2, 32, 2048, generator=torch.Generator().manual_seed(5))
    feat, hidden, t5 = reference(True)
    (t5 * G).sum().backward()
    ref_grads = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    gpu = copy.deepcopy(model).to(DEV)
    gpu.zero_grad(set_to_none=True)
    f = samples["pc_feat"].to(DEV).requires_grad_(True)
    out = gpu({"pc_feat": f, "pc": samples["pc"].to(DEV)})
    (out["inputs_t5"] * G.to(DEV)).sum().backward()
    assert _rel(out["inputs_t5"].detach().cpu(), t5.detach()) < 1e-4
    _within("config5 d/d pc_feat (Nk=5000)", _rel(f.grad.cpu(), feat.grad), QF_GRAD_TOL)   # 1408-long dot products, 12 layers deep
    n = 0
    for name, p in gpu.named_parameters():
        if ("crossattention.self" in name or name in ("query_tokens", "t5_proj.weight")) and not _is_key_bias(name):
            r = _rel(p.grad.cpu(), ref_grads[name])
            _within("config5 cross-attention grad: " + name, r, QF_GRAD_TOL)
            n += 1
    assert n >= 6 * 5


def test_config5_blip2_shape_nk80000_forward_backward_matches_oracle():
    """BASELINE config 5's token count in BOTH directions (the largest tested backward was Nk = 5000):
    B = 1, Nk = 80 000 point tokens of width 1408 -- the backward takes the register-resident one-query-tile
    attention kernel (`attention_bwd_kernel<64, ONEQT>`, Qformer.py:185-227) and the K = 80 000 weight-gradient
    products of the six cross-attention key / value projections."""
    model, samples, reference = _blip2(80000, 1, seed=53)
    G = torch.randn(1, 32, 2048, generator=torch.Generator().manual_seed(6))
    feat, hidden, t5 = reference(True)
    (t5 * G).sum().backward()
    ref_grads = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    gpu = copy.deepcopy(model).to(DEV)
    gpu.zero_grad(set_to_none=True)
    f = samples["pc_feat"].to(DEV).requires_grad_(True)
    out = gpu({"pc_feat": f, "pc": samples["pc"].to(DEV)})
    (out["inputs_t5"] * G.to(DEV)).sum().backward()
    assert _rel(out["inputs_t5"].detach().cpu(), t5.detach()) < 1e-4      # north-star fp32 bar
    _within("config5 d/d pc_feat (Nk=80000)", _rel(f.grad.cpu(), feat.grad), QF_GRAD_TOL)
    n = 0
    for name, p in gpu.named_parameters():
        if ("crossattention.self" in name or name in ("query_tokens", "t5_proj.weight")) and not _is_key_bias(name):
            r = _rel(p.grad.cpu(), ref_grads[name])
            _within("config5 cross-attention grad: " + name, r, QF_GRAD_TOL)
            n += 1
    assert n >= 6 * 5


def test_a_reused_encoder_buffer_is_split_again_every_forward():
    """Qformer.bert handed the SAME encoder tensor object (>= 8192 rows: the planes path) refilled in ways that do not
    advance its autograd version -- a `.data` write, as a static hipGraph input or a raw-pointer kernel would do: the
    second forward must see the new contents (the planes of the encoder tokens live for one encoder forward only;
    a process-wide cache keyed on identity + `_version` returned the previous batch's planes here)."""
    from situation3d_amd import qformer
    from situation3d_amd.blip2 import Blip2PointQFormer
    torch.manual_seed(61)
    small = dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                 max_position_embeddings=64, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    model = Blip2PointQFormer(point_width=96, qformer_overrides=small).to(DEV).eval()
    g = torch.Generator().manual_seed(3)
    a, b_ = torch.randn(2, 4608, 96, generator=g).to(DEV), torch.randn(2, 4608, 96, generator=g).to(DEV)
    assert qformer._big_source(a, 2 * 128)
    q = model.query_tokens.expand(2, -1, -1)

    def run(enc):
        with torch.no_grad():
            return model.Qformer.bert(query_embeds=q, encoder_hidden_states=enc,
                                      encoder_attention_mask=torch.ones(2, enc.shape[1], device=DEV)).last_hidden_state
    ref_a, ref_b = run(a.clone()), run(b_.clone())
    assert not torch.allclose(ref_a, ref_b, atol=1e-3)
    buf = a.clone()
    version = buf._version
    assert torch.equal(run(buf), ref_a)
    buf.data.copy_(b_)                       # refilled behind autograd's back
    assert buf._version == version
    assert torch.equal(run(buf), ref_b)


@pytest.mark.parametrize("own_gemm", [0, 64, 127])
def test_full_size_qformer_forward_backward_matches_oracle(own_gemm, monkeypatch):
    """768 wide / 12 layers / 12 heads / 6 cross-attention layers, B = 8, 32 queries + 20 question tokens,
    256 scene tokens of width 256: the Q-Former the bench times, two-segment layout, against qformer_ref --
    last hidden state, per-parameter gradients of every layer, gradient of the scene tokens.
    own_gemm: the dense layers on sig3d_gemm16 (SIG3D_QF_GEMM=1: split reductions, slabs added by the LayerNorm
    tails, slabs handed from block to block in the backward pass) instead of the vendor library -- same bounds
    (64: one product, the input gradient of the projections; 127: all seven)."""
    from oracle import qformer_ref
    from situation3d_amd import qformer as qformer_mod
    from situation3d_amd.qformer import init_Qformer
    monkeypatch.setattr(qformer_mod, "OWN_GEMM", own_gemm != 0)      # 0: the library; 64: the default products; 127: all
    monkeypatch.setattr(qformer_mod, "OWN_MASK", own_gemm if own_gemm else 127)
    torch.manual_seed(61)
    qf, query_tokens = init_Qformer(32, 256)
    qf.eval()
    g = torch.Generator().manual_seed(62)
    with torch.no_grad():
        for p in qf.parameters():
            if p.dim() == 1:
                p.add_(torch.randn(p.shape, generator=g) * 0.05)
    B, T = 8, 20
    enc = torch.randn(B, 256, 256, generator=g)
    ids = torch.randint(1000, 30000, (B, T), generator=g)
    att = torch.ones(B, 32 + T, dtype=torch.long)
    att[3, -5:] = 0                                            # a padded question
    G = torch.randn(B, 32 + T, 768, generator=g)
    c = qf.config
    cfg = dict(num_hidden_layers=c.num_hidden_layers, num_attention_heads=c.num_attention_heads,
               layer_norm_eps=c.layer_norm_eps, add_cross_attention=True, cross_attention_freq=c.cross_attention_freq)
    sd = dict(qf.bert.state_dict())
    sd.update(dict(qf.bert.named_parameters()))
    e_ref = enc.clone().requires_grad_(True)
    q_ref = query_tokens.detach().clone().requires_grad_(True)
    ref = qformer_ref.bert_model(sd, cfg, query_embeds=q_ref.expand(B, -1, -1), input_ids=ids, attention_mask=att,
                                 encoder_hidden_states=e_ref)
    (ref * G).sum().backward()
    ref_grads = {n: p.grad.clone() for n, p in qf.bert.named_parameters() if p.grad is not None}
    gpu = copy.deepcopy(qf).to(DEV)
    gpu.zero_grad(set_to_none=True)
    e = enc.to(DEV).requires_grad_(True)
    q = query_tokens.detach().to(DEV).requires_grad_(True)
    out = gpu.bert(query_embeds=q.expand(B, -1, -1), input_ids=ids.to(DEV), attention_mask=att.to(DEV),
                   encoder_hidden_states=e, return_dict=True).last_hidden_state
    (out * G.to(DEV)).sum().backward()
    assert _rel(out.detach().cpu(), ref.detach()) < 1e-4           # north-star fp32 bar
    _within("full-size Q-Former d/d scene tokens", _rel(e.grad.cpu(), e_ref.grad), QF_GRAD_TOL)
    _within("full-size Q-Former d/d query tokens", _rel(q.grad.cpu(), q_ref.grad), QF_GRAD_TOL)
    worst, n = 0.0, 0
    for name, p in gpu.bert.named_parameters():
        if _is_key_bias(name):
            assert float(p.grad.abs().max()) < 1e-5 and float(ref_grads[name].abs().max()) < 1e-5
            continue
        if name == "embeddings.word_embeddings.weight":
            rows = ids.unique()
            r = _rel(p.grad[rows.to(DEV)].cpu(), ref_grads[name][rows])
        else:
            r = _rel(p.grad.cpu(), ref_grads[name])
        worst, n = max(worst, r), n + 1
        _within("full-size Q-Former grad: " + name, r, QF_GRAD_TOL)
    assert n > 230, n
