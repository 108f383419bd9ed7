"""Blip2T5 point branch: the position-embedding gather+add kernel
against the loop-for-loop CPU restatement of blip2_t5.py:106-118 (bit-exact: one rounded multiply
and one rounded add per element, like `pc_embeds + 0.01 * all_pcs`), and the forward(samples)
contract of the module."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("b,n,c", [(2, 50, 1408), (1, 5000, 1408), (3, 7, 30)])
def test_pos_embed_add_matches_reference_loops(b, n, c):
    from oracle import blip2_ref
    from situation3d_amd.blip2 import add_position_embedding, sinusoid_table
    g = torch.Generator().manual_seed(n + c)
    feat = torch.randn(b, n, c, generator=g)
    pc = torch.randint(0, 256, (b, n, 3), generator=g).float()
    pc[0, -1] = 0.0  # zero-padded points (threedvqa_datasets.py:77-79)
    table = sinusoid_table(256, c // 3)
    ref = blip2_ref.add_position_embedding(feat, pc, table, 0.01)
    got = add_position_embedding(feat.to(DEV), pc.to(DEV), table.to(DEV), 0.01).cpu()
    assert torch.equal(got, ref)
    if c % 3:  # channels beyond 3*(c//3) are passed through untouched (the ":1407" slice)
        assert torch.equal(got[..., 3 * (c // 3):], feat[..., 3 * (c // 3):])


def test_sinusoid_table_properties():
    from situation3d_amd.blip2 import sinusoid_table
    t = sinusoid_table(256, 469)
    assert t.shape == (256, 469)
    assert torch.all(t[0, 0::2] == 0) and torch.all(t[0, 1::2] == 1)  # sin(0), cos(0) interleaved
    torch.testing.assert_close(t[3, 0], torch.sin(torch.tensor(3.0)))


def test_blip2_point_qformer_forward_samples_contract():
    """forward(samples) -> dict with "loss" (base_task.py:63-65) and the T5 inputs (blip2_t5.py:128)."""
    from situation3d_amd.blip2 import Blip2PointQFormer
    small = dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                 hidden_dropout_prob=0.0)
    torch.manual_seed(0)
    head = lambda inputs_t5, atts, samples: inputs_t5.pow(2).mean()  # stand-in for the frozen T5
    m = Blip2PointQFormer(num_query_token=8, point_width=96, t5_hidden=64, language_head=head,
                          qformer_overrides=small).to(DEV).train()
    keys = set(m.Qformer.bert.state_dict().keys())
    assert not any("word_embeddings" in k or ".intermediate." in k for k in keys)  # blip2_t5.py:63-69
    samples = {"pc_feat": torch.randn(2, 300, 96, device=DEV),
               "pc": torch.randint(0, 256, (2, 300, 3), device=DEV).float()}
    out = m(samples)
    assert out["inputs_t5"].shape == (2, 8, 64) and out["atts_t5"].shape == (2, 8)
    out["loss"].backward()
    assert m.query_tokens.grad is not None and torch.isfinite(m.query_tokens.grad).all()
    assert m.t5_proj.weight.grad.abs().sum() > 0
