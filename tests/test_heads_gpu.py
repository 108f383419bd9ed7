"""csrc/heads.hip behind situation3d_amd.heads.pooled_heads: mean over the query rows + the auxiliary and the answer MLP
(sqa_module.py: aux_reg / answer_cls on the pooled Q-Former output) against the torch modules in float64 -- scores, the
gradient of the row matrix and all eight parameter gradients; dropout: the same keep bits forward and backward."""
import copy

import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _heads(hidden, n_ans, seed):
    torch.manual_seed(seed)
    aux = nn.Sequential(nn.Linear(hidden, hidden), nn.GELU(), nn.Linear(hidden, 7))
    ans = nn.Sequential(nn.Linear(hidden, hidden), nn.GELU(), nn.Dropout(0.1), nn.Linear(hidden, n_ans))
    return aux.to(DEV), ans.to(DEV)


def _close(a, b, name, tol=3e-6):
    err = float((a.detach().double() - b.detach().double()).abs().max())
    assert err <= tol * max(1.0, float(b.detach().abs().max())), (name, err)


@pytest.mark.parametrize("b,q,hidden,n_ans,tail", [(8, 32, 768, 706, 256), (1, 32, 768, 706, 0), (16, 4, 64, 10, 3),
                                                  (5, 7, 256, 33, 0), (9, 32, 768, 706, 100)])
def test_scores_and_gradients_match_the_modules(b, q, hidden, n_ans, tail):
    from situation3d_amd import heads
    aux, ans = _heads(hidden, n_ans, b + q)
    ans[2].p = 0.0
    ref_aux, ref_ans = copy.deepcopy(aux).double(), copy.deepcopy(ans).double()
    rows = torch.randn(b * q + tail, hidden, device=DEV, requires_grad=True)
    assert heads.covered(rows, b, q, aux, ans)
    s_aux, s_ans = heads.pooled_heads(rows, b, q, aux, ans)
    rows64 = rows.detach().double().requires_grad_(True)
    pooled = rows64[:b * q].view(b, q, hidden).mean(1)
    r_aux, r_ans = ref_aux(pooled), ref_ans(pooled)
    _close(s_aux, r_aux, "aux scores")
    _close(s_ans, r_ans, "answer scores")
    g1, g2 = torch.randn_like(s_aux), torch.randn_like(s_ans)
    torch.autograd.backward([s_aux, s_ans], [g1, g2])
    torch.autograd.backward([r_aux, r_ans], [g1.double(), g2.double()])
    _close(rows.grad, rows64.grad, "d rows")
    assert float(rows.grad[b * q:].abs().sum()) == 0 if tail else True
    for (n, p), (_, pr) in zip(list(aux.named_parameters()) + list(ans.named_parameters()),
                               list(ref_aux.named_parameters()) + list(ref_ans.named_parameters())):
        _close(p.grad, pr.grad, n)


def test_dropout_keeps_the_same_elements_forward_and_backward():
    from situation3d_amd import heads
    from situation3d_amd.qformer import advance_dropout_seed
    b, q, hidden, n_ans = 8, 32, 768, 706
    aux, ans = _heads(hidden, n_ans, 3)
    aux.train(); ans.train()
    rows = torch.randn(b * q, hidden, device=DEV, requires_grad=True)
    seen = []
    orig = heads._lib.call

    def spy(name, *a):
        rc = orig(name, *a)
        if name == "sig3d_pooled_heads_fwd":
            seen.append(a)
        return rc
    heads._lib.call = spy
    try:
        s_aux, s_ans = heads.pooled_heads(rows, b, q, aux, ans)
    finally:
        heads._lib.call = orig
    fn = s_ans.grad_fn
    pooled, pre, h = fn.saved_tensors[0], fn.saved_tensors[1], fn.saved_tensors[2]
    act = torch.nn.functional.gelu(pre[1])
    kept = h[1] != 0
    frac = float(kept.float().mean())
    assert 0.87 < frac < 0.93, frac
    torch.testing.assert_close(h[1][kept], (act / 0.9)[kept], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(h[0], torch.nn.functional.gelu(pre[0]), rtol=1e-5, atol=1e-6)   # no dropout in aux_reg
    s_ans.sum().backward()
    # d pre = (W2^T 1) * gelu'(pre) * keep / 0.9: zero exactly where the forward dropped
    g_b1 = ans[0].bias.grad          # = sum over samples of d pre
    w2 = ans[3].weight
    dh = w2.sum(0)                   # d h for every sample
    u = pre[1]
    gelu_grad = 0.5 * (1 + torch.erf(u / 2 ** 0.5)) + u * torch.exp(-0.5 * u * u) / (2 * torch.pi) ** 0.5
    dpre = dh[None, :] * gelu_grad * kept.float() / 0.9
    torch.testing.assert_close(g_b1, dpre.sum(0), rtol=1e-4, atol=1e-5)
    # a new forward pass (new seed) drops other elements
    advance_dropout_seed(DEV)
    s2 = heads.pooled_heads(rows, b, q, aux, ans)[1]
    kept2 = s2.grad_fn.saved_tensors[2][1] != 0
    assert float((kept2 != kept).float().mean()) > 0.05


def test_model_takes_the_fused_heads_and_the_switch_restores_the_modules(monkeypatch):
    import bench
    from situation3d_amd import heads
    from situation3d_amd.model import SIG3DQFormer
    torch.manual_seed(0)
    model = SIG3DQFormer(num_answers=bench.NUM_ANSWERS, qformer_overrides=dict(num_hidden_layers=2)).to(DEV).eval()
    batch = bench.synthetic_batch(2, 4096, 3, DEV)
    calls = []
    orig = heads._lib.call

    def spy(name, *a):
        calls.append(name)
        return orig(name, *a)
    monkeypatch.setattr(heads._lib, "call", spy)
    with torch.no_grad():
        out = model(dict(batch))
        assert "sig3d_pooled_heads_fwd" in calls
        a1, c1 = out["aux_scores"].clone(), out["answer_scores"].clone()
        monkeypatch.setattr(heads, "ENABLED", False)
        del calls[:]
        out = model(dict(batch))
        assert "sig3d_pooled_heads_fwd" not in calls
    torch.testing.assert_close(a1, out["aux_scores"], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(c1, out["answer_scores"], rtol=1e-4, atol=1e-5)
