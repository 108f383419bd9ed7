"""Generate tests/golden/*.npz -- run ONLY in the build container, where /root/reference exists.

    python tests/golden/make_golden.py

What is pinned and how:
  * pointnet2_modules_*.npz : the REFERENCE's own Python (lib/pointnet2/pointnet2_utils.py,
    pointnet2_modules.py, pytorch_utils.py, imported from /root/reference, unmodified) is run
    on CPU with `pointnet2._ext` bound to the C oracle (the reference's native ops have no CPU
    branch).  Inputs, module weights, outputs and gradients are stored.
  * qformer_*.npz : the REFERENCE's Qformer.py (BertModel) is imported with five in-process
    compatibility shims for transformers 5.x (listed below), built from a small local BertConfig
    and run on CPU in eval mode.  state_dict, inputs, outputs, per-layer hidden states and
    gradients are stored.
  * pointnet2_ops.npz : the oracle's own outputs on seeded inputs (regression pin for the C
    restatement) + the reference's only op test vector (pointnet2_test.py:18-30).
Only data (inputs / expected outputs) is written; no reference source text is copied.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import pointnet2_ref as oracle  # noqa: E402
from util import feats, scene  # noqa: E402


def _np(t):
    return t.detach().cpu().numpy()


def _save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print("wrote %s (%.1f KiB)" % (path, os.path.getsize(path) / 1024))


# ------------------------------------------------------------------------------------------
def import_reference_pointnet2():
    """Bind the oracle as `pointnet2._ext`, then import the reference's Python modules."""
    pkg = types.ModuleType("pointnet2")
    ext = types.ModuleType("pointnet2._ext")
    for fn in ("gather_points", "gather_points_grad", "furthest_point_sampling", "three_nn",
               "three_interpolate", "three_interpolate_grad", "ball_query", "group_points",
               "group_points_grad"):
        setattr(ext, fn, getattr(oracle, fn))
    pkg._ext = ext
    sys.modules["pointnet2"] = pkg
    sys.modules["pointnet2._ext"] = ext
    sys.path.insert(0, os.path.join(REF, "lib", "pointnet2"))
    import pointnet2_modules  # noqa: the reference's file
    import pointnet2_utils  # noqa
    return pointnet2_modules, pointnet2_utils


def _module_io(module, args, out_index, seed):
    """Run fwd + bwd of loss = sum(out * G); return dict of arrays."""
    module.train()
    outs = module(*args)
    out = outs[out_index] if isinstance(outs, tuple) else outs
    G = torch.randn(out.shape, generator=torch.Generator().manual_seed(seed))
    (out * G).sum().backward()
    rec = {"G": _np(G)}
    for i, o in enumerate(outs if isinstance(outs, tuple) else (outs,)):
        if o is not None:
            rec["out%d" % i] = _np(o)
    for k, v in module.named_parameters():
        rec["grad." + k] = _np(v.grad)
    for k, v in module.state_dict().items():  # AFTER forward: BN running stats updated
        rec["state_after." + k] = _np(v)
    return rec


def golden_pointnet2_modules():
    M, U = import_reference_pointnet2()
    # (1) the reference's __main__ smoke (pointnet2_modules.py:504-523): seed 1, B=2, N=9,
    #     SA-MSG npoint=2 radii [5,10] nsamples [6,3].  The reference passes features shaped
    #     (2,9,6) (C=9 channels over N=6 points) while its indices address N=9 points -- an
    #     out-of-bounds read in its own smoke; here features are (2,6,9) with mlps [[6,3],[6,6]].
    torch.manual_seed(1)
    xyz = torch.randn(2, 9, 3, requires_grad=True)
    f = torch.randn(2, 6, 9, requires_grad=True)
    mod = M.PointnetSAModuleMSG(npoint=2, radii=[5.0, 10.0], nsamples=[6, 3], mlps=[[6, 3], [6, 6]])
    state0 = {k: v.clone() for k, v in mod.state_dict().items()}
    rec = _module_io(mod, (xyz, f), 1, seed=11)
    rec.update({"xyz": _np(xyz), "features": _np(f), "grad_xyz": _np(xyz.grad),
                "grad_features": _np(f.grad)})
    rec.update({"state." + k: _np(v) for k, v in state0.items()})
    _save("pointnet2_modules_msg_smoke.npz", **rec)

    # (2) BASELINE config 1: one synthetic scene, 4096 pts, SA1 (2048 / 0.2 / 64 / [3,64,64,128])
    torch.manual_seed(2)
    xyz = scene(1, 4096, seed=21)
    f = feats(1, 3, 4096, seed=22).requires_grad_(True)
    mod = M.PointnetSAModuleVotes(npoint=2048, radius=0.2, nsample=64, mlp=[3, 64, 64, 128],
                                  use_xyz=True, normalize_xyz=True)
    state0 = {k: v.clone() for k, v in mod.state_dict().items()}
    rec = _module_io(mod, (xyz, f), 1, seed=12)
    rec.update({"xyz": _np(xyz), "features": _np(f), "grad_features": _np(f.grad)})
    rec.update({"state." + k: _np(v) for k, v in state0.items()})
    rec["out1"] = rec["out1"].astype(np.float32)
    _save("pointnet2_modules_sa1_4096.npz", **rec)

    # (3) SA-Votes with avg and rbf pooling, no normalisation, ragged sizes
    for pooling in ("avg", "rbf"):
        torch.manual_seed(3)
        xyz = scene(2, 300, seed=31, dup=40)
        f = feats(2, 5, 300, seed=32).requires_grad_(True)
        mod = M.PointnetSAModuleVotes(npoint=37, radius=0.9, nsample=12, mlp=[5, 16, 8],
                                      use_xyz=True, pooling=pooling)
        state0 = {k: v.clone() for k, v in mod.state_dict().items()}
        rec = _module_io(mod, (xyz, f), 1, seed=13)
        rec.update({"xyz": _np(xyz), "features": _np(f), "grad_features": _np(f.grad)})
        rec.update({"state." + k: _np(v) for k, v in state0.items()})
        _save("pointnet2_modules_votes_%s.npz" % pooling, **rec)

    # (4) feature propagation (pointnet2_modules.py:376-421)
    torch.manual_seed(4)
    unknown = scene(2, 128, seed=41)
    known = scene(2, 32, seed=42)
    uf = feats(2, 7, 128, seed=43).requires_grad_(True)
    kf = feats(2, 11, 32, seed=44).requires_grad_(True)
    mod = M.PointnetFPModule(mlp=[18, 16, 16])
    state0 = {k: v.clone() for k, v in mod.state_dict().items()}
    rec = _module_io(mod, (unknown, known, uf, kf), 0, seed=14)
    rec.update({"unknown": _np(unknown), "known": _np(known), "unknow_feats": _np(uf),
                "known_feats": _np(kf), "grad_unknow_feats": _np(uf.grad),
                "grad_known_feats": _np(kf.grad)})
    rec.update({"state." + k: _np(v) for k, v in state0.items()})
    _save("pointnet2_modules_fp.npz", **rec)


# ------------------------------------------------------------------------------------------
def golden_pointnet2_ops():
    rec = {}
    xyz = scene(2, 1500, seed=51, dup=200, zero_tail=64)
    rec["fps.xyz"] = _np(xyz)
    rec["fps.idx"] = _np(oracle.furthest_point_sampling(xyz, 300))
    new_xyz = oracle.gather_points(xyz.transpose(1, 2).contiguous(),
                                   torch.from_numpy(rec["fps.idx"])).transpose(1, 2).contiguous()
    rec["bq.new_xyz"] = _np(new_xyz)
    rec["bq.idx"] = _np(oracle.ball_query(new_xyz, xyz, 0.5, 16))
    d2, i3 = oracle.three_nn(xyz[:, :200].contiguous(), new_xyz)
    rec["nn.dist2"], rec["nn.idx"] = _np(d2), _np(i3)
    # the reference's only op-level test vector: pointnet2_test.py:18-30
    rec["ti.idx"] = np.array([[[0, 1, 2], [1, 2, 3]]], dtype=np.int32)
    rec["ti.weight"] = np.array([[[1, 1, 1], [2, 2, 2]]], dtype=np.float32)
    pts = torch.randn(1, 2, 4, generator=torch.Generator().manual_seed(52))
    rec["ti.points"] = _np(pts)
    rec["ti.out"] = _np(oracle.three_interpolate(pts, torch.from_numpy(rec["ti.idx"]),
                                                 torch.from_numpy(rec["ti.weight"])))
    _save("pointnet2_ops.npz", **rec)


# ------------------------------------------------------------------------------------------
def import_reference_qformer():
    """transformers 5.x compatibility shims (the reference pins 4.31, environment.yml:274):
    (1,2) apply_chunking_to_forward / prune_linear_layer moved to transformers.pytorch_utils,
    (3) find_pruneable_heads_and_indices removed, (4) init_weights protocol changed,
    (5) get_head_mask removed.  None of them touches the arithmetic."""
    import transformers
    import transformers.modeling_utils as mu
    import transformers.pytorch_utils as pu
    mu.apply_chunking_to_forward = pu.apply_chunking_to_forward
    mu.prune_linear_layer = pu.prune_linear_layer
    mu.find_pruneable_heads_and_indices = lambda *a, **k: (set(), None)
    sys.path.insert(0, os.path.join(REF, "3DLLM_BLIP2-base"))
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "ref_qformer", os.path.join(REF, "3DLLM_BLIP2-base/lavis/models/blip2_models/Qformer.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.BertPreTrainedModel.init_weights = lambda self: self.apply(self._init_weights)
    mod.BertModel.get_head_mask = lambda self, hm, n, *a, **k: [None] * n
    return mod, transformers


# a handful of weight gradients is enough to pin the backward pass (fixture size)
GRAD_KEYS = {
    "embeddings.LayerNorm.weight", "embeddings.word_embeddings.weight",
    "encoder.layer.0.attention.self.query.weight", "encoder.layer.0.attention.self.key.bias",
    "encoder.layer.0.crossattention.self.key.weight", "encoder.layer.0.crossattention.self.value.bias",
    "encoder.layer.0.crossattention.output.dense.weight",
    "encoder.layer.0.intermediate_query.dense.weight", "encoder.layer.1.output_query.LayerNorm.weight",
    "encoder.layer.1.intermediate.dense.weight", "encoder.layer.1.attention.self.value.weight",
}


def golden_qformer():
    Q, transformers = import_reference_qformer()
    from transformers.models.bert.configuration_bert import BertConfig
    cfg = BertConfig(vocab_size=60, hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                     intermediate_size=192, max_position_embeddings=40,
                     hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
    cfg.encoder_width = 96
    cfg.add_cross_attention = True
    cfg.cross_attention_freq = 2
    cfg.query_length = 8
    torch.manual_seed(7)
    model = Q.BertModel(cfg, add_pooling_layer=False)
    model.eval()
    # non-trivial biases / LayerNorm affine so every parameter is exercised
    g = torch.Generator().manual_seed(70)
    with torch.no_grad():
        for p in model.parameters():
            if p.dim() == 1:
                p.add_(torch.randn(p.shape, generator=g) * 0.05)
    state = {k: _np(v) for k, v in model.state_dict().items()}

    B, Nq, Nk, T = 2, 8, 77, 6
    query = (torch.randn(B, Nq, 128, generator=g) * 0.5).requires_grad_(True)
    enc = torch.randn(B, Nk, 96, generator=g).requires_grad_(True)
    enc_mask = torch.ones(B, Nk, dtype=torch.long)
    enc_mask[1, 50:] = 0  # padded point tokens in sample 1
    G = torch.randn(B, Nq, 128, generator=g)

    # (a) Blip2T5-style call: queries only (blip2_t5.py:121-127)
    out = model(query_embeds=query, encoder_hidden_states=enc, encoder_attention_mask=enc_mask,
                output_hidden_states=True, return_dict=True)
    (out.last_hidden_state * G).sum().backward()
    rec = {"state." + k: v for k, v in state.items()}
    rec.update({"query_embeds": _np(query), "encoder_hidden_states": _np(enc),
                "encoder_attention_mask": _np(enc_mask), "G": _np(G),
                "last_hidden_state": _np(out.last_hidden_state),
                "grad_query_embeds": _np(query.grad), "grad_encoder_hidden_states": _np(enc.grad)})
    for i, hsv in enumerate(out.hidden_states):
        rec["hidden_states.%d" % i] = _np(hsv)
    for k, v in model.named_parameters():
        if v.grad is not None and k in GRAD_KEYS:
            rec["grad." + k] = _np(v.grad)
    model.zero_grad()
    query.grad = None
    enc.grad = None

    # (b) queries + question tokens in one self-attention (blip2_qformer.py:190-197 style):
    #     input_ids (B,T) with padding, attention_mask over [queries | text]
    ids = torch.randint(1, 60, (B, T), generator=g)
    txt_mask = torch.ones(B, T, dtype=torch.long)
    txt_mask[0, 4:] = 0
    ids[0, 4:] = 0
    attn = torch.cat([torch.ones(B, Nq, dtype=torch.long), txt_mask], 1)
    G2 = torch.randn(B, Nq + T, 128, generator=g)
    out2 = model(input_ids=ids, attention_mask=attn, query_embeds=query, encoder_hidden_states=enc,
                 encoder_attention_mask=enc_mask, return_dict=True)
    (out2.last_hidden_state * G2).sum().backward()
    rec.update({"t.input_ids": _np(ids), "t.attention_mask": _np(attn), "t.G": _np(G2),
                "t.last_hidden_state": _np(out2.last_hidden_state),
                "t.grad_query_embeds": _np(query.grad),
                "t.grad_encoder_hidden_states": _np(enc.grad)})
    for k, v in model.named_parameters():
        if v.grad is not None and k in GRAD_KEYS:
            rec["t.grad." + k] = _np(v.grad)
    rec["config"] = np.array([cfg.vocab_size, cfg.hidden_size, cfg.num_hidden_layers,
                              cfg.num_attention_heads, cfg.intermediate_size,
                              cfg.max_position_embeddings, cfg.encoder_width,
                              cfg.cross_attention_freq, cfg.query_length])
    rec["transformers_version"] = np.array(transformers.__version__)
    _save("qformer_small.npz", **rec)


if __name__ == "__main__":
    assert os.path.isdir(REF), "golden vectors can only be generated where /root/reference exists"
    torch.set_num_threads(4)
    golden_pointnet2_ops()
    golden_pointnet2_modules()
    golden_qformer()
