"""FlatAdamW (csrc/optim.hip) == clip_grad_value_ + torch.optim.AdamW + zero_grad
(lib/solver.py:618-627, situation3d/train/train.py:226-238), step for step."""
import copy

import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _net():
    torch.manual_seed(0)
    return nn.Sequential(nn.Linear(37, 64), nn.LayerNorm(64), nn.GELU(), nn.Linear(64, 5)).to(DEV)


def _groups(m, wd):
    decay = [p for n, p in m.named_parameters() if "bias" not in n]
    no_decay = [p for n, p in m.named_parameters() if "bias" in n]
    return [{"params": decay, "weight_decay": wd}, {"params": no_decay, "weight_decay": 0.0}]


def test_flat_adamw_matches_torch_adamw_with_value_clip():
    from situation3d_amd.optim import FlatAdamW
    a, b = _net(), None
    b = copy.deepcopy(a)
    ref = torch.optim.AdamW(_groups(a, 0.05), lr=1e-2, betas=(0.9, 0.999), eps=1e-8)
    opt = FlatAdamW(_groups(b, 0.05), lr=1e-2, betas=(0.9, 0.999), eps=1e-8, clip_value=0.05)
    g = torch.Generator().manual_seed(1)
    for step in range(25):
        x = torch.randn(16, 37, generator=g).to(DEV)
        y = torch.randn(16, 5, generator=g).to(DEV)
        ref.zero_grad(set_to_none=False)
        (10 * (a(x) - y).pow(2).mean()).backward()
        nn.utils.clip_grad_value_(a.parameters(), 0.05)
        ref.step()
        (10 * (b(x) - y).pow(2).mean()).backward()   # no zero_grad: fused into the previous step()
        opt.step()
    for (n, p), q in zip(a.named_parameters(), b.parameters()):
        torch.testing.assert_close(q, p, rtol=1e-5, atol=1e-6, msg=lambda m: n + ": " + m)
        assert q.grad.abs().max() == 0  # zeroed by step()
    # parameters are views of one flat buffer per group; state_dict keys/shapes unchanged
    assert [k for k in b.state_dict()] == [k for k in a.state_dict()]


def test_flat_adamw_inside_graph_replay():
    from situation3d_amd.optim import FlatAdamW
    work = torch.cuda.Stream()
    with torch.cuda.stream(work):
        a = _net()
        b = copy.deepcopy(a)
        ref = FlatAdamW(_groups(a, 0.0), lr=1e-2, clip_value=1.0)
        opt = FlatAdamW(_groups(b, 0.0), lr=1e-2, clip_value=1.0)
        x = torch.randn(8, 37, device=DEV)
        y = torch.randn(8, 5, device=DEV)
        for _ in range(2):  # warm-up (both models identically)
            for m, o in ((a, ref), (b, opt)):
                (m(x) - y).pow(2).mean().backward()
                o.step()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=work):
            (b(x) - y).pow(2).mean().backward()
            opt.step()
        for _ in range(5):
            (a(x) - y).pow(2).mean().backward()
            ref.step()
            graph.replay()
        torch.cuda.synchronize()
        for p, q in zip(a.parameters(), b.parameters()):
            torch.testing.assert_close(q, p, rtol=1e-5, atol=1e-6)
