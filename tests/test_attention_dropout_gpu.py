"""Attention-probability dropout inside the fused attention kernels (Qformer.py:219).

The keep mask is hash(device counter, call id, (b,h,q,k)); it is recovered here with V = identity
(out[q][k] = keep * P / (1-p) > 0 iff kept) and the kernels are then checked, forward and backward,
against torch's `softmax -> dropout(mask) -> matmul` with that same mask.  Also: drop rate,
determinism for a fixed seed, fresh masks after advance_dropout_seed, eval-mode determinism."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _tm(t):  # (b,h,n,d) -> token-major (b,n,h*d)
    return t.permute(0, 2, 1, 3).reshape(t.shape[0], t.shape[2], -1).contiguous()


@pytest.mark.parametrize("nq,nk", [(40, 64), (32, 64), (20, 64),     # <= 64 x 64: the small-problem backward's 2 x 2 form
                                   (32, 256), (20, 200),              # 32 x <= 1024: its one-query-block form
                                   (40, 200)])                        # neither: the generic backward
def test_attention_dropout_matches_torch_with_recovered_mask(nq, nk):
    from situation3d_amd.qformer import fused_attention
    b, h, p = 2, 3, 0.25
    g = torch.Generator().manual_seed(0)
    q = torch.randn(b, h, nq, 64, generator=g).to(DEV)
    k = torch.randn(b, h, nk, 64, generator=g).to(DEV)

    def probe_mask(call_id):      # V = 64 keys' worth of identity at a time: out[q][j] = keep * P / (1 - p) of key 64 c + j
        cols = []
        for c in range((nk + 63) // 64):
            sel = torch.zeros(nk, 64, device=DEV)
            n_here = min(64, nk - 64 * c)
            sel[64 * c + torch.arange(n_here, device=DEV), torch.arange(n_here, device=DEV)] = 1.0
            o = fused_attention(_tm(q), _tm(k), _tm(sel.expand(b, h, nk, 64).contiguous()), None, h, p, call_id=call_id)
            cols.append(o.view(b, nq, h, 64).permute(0, 2, 1, 3)[..., :n_here])
        return torch.cat(cols, -1)                                            # (b,h,nq,nk)

    probe = probe_mask(11)
    keep = probe > 0
    rate = 1.0 - keep.float().mean().item()
    assert abs(rate - p) < 4 * math.sqrt(p * (1 - p) / keep.numel()), rate
    assert torch.equal(probe, probe_mask(11))                                 # fixed seed: same mask
    assert not torch.equal(keep, probe_mask(12) > 0)                          # per-module streams

    v = torch.randn(b, h, nk, 64, generator=g).to(DEV)
    go = torch.randn(b, nq, h * 64, generator=g).to(DEV)
    qt, kt, vt = (_tm(t).requires_grad_(True) for t in (q, k, v))
    out = fused_attention(qt, kt, vt, None, h, p, call_id=11)
    out.backward(go)

    q64, k64, v64 = (t.double().requires_grad_(True) for t in (q, k, v))
    probs = torch.softmax(torch.matmul(q64, k64.transpose(-1, -2)) / 8.0, -1)
    ctx = torch.matmul(probs * keep.double() / (1 - p), v64)
    ref = ctx.permute(0, 2, 1, 3).reshape(b, nq, h * 64)
    (ref * go.double()).sum().backward()
    torch.testing.assert_close(out.double(), ref, rtol=1e-4, atol=1e-4)
    untm = lambda t, n: t.view(b, n, h, 64).permute(0, 2, 1, 3).double()
    torch.testing.assert_close(untm(qt.grad, nq), q64.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(untm(kt.grad, nk), k64.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(untm(vt.grad, nk), v64.grad, rtol=1e-4, atol=1e-4)


def test_qformer_train_mode_uses_both_dropouts_and_eval_is_deterministic():
    from situation3d_amd.qformer import init_Qformer
    torch.manual_seed(0)
    qf, qt = init_Qformer(8, 96, hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                          intermediate_size=256, hidden_dropout_prob=0.1,
                          attention_probs_dropout_prob=0.1)
    qf, qt = qf.to(DEV), qt.detach().to(DEV)
    enc = torch.randn(2, 50, 96, device=DEV)
    run = lambda: qf.bert(query_embeds=qt.expand(2, -1, -1), encoder_hidden_states=enc,
                          return_dict=True).last_hidden_state
    qf.train()
    a, b = run(), run()
    assert not torch.equal(a, b)  # the seed advances once per training forward
    qf.eval()
    c, d = run(), run()
    assert torch.equal(c, d)
    assert torch.isfinite(a).all() and (a - c).abs().mean() > 0
