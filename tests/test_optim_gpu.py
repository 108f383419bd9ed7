"""FlatAdamW (csrc/optim.hip) == clip_grad_value_ + torch.optim.AdamW + zero_grad(set_to_none)
(lib/solver.py:618-627, situation3d/train/train.py:226-238), step for step -- through the
pointer-table kernel (scattered gradients), the gather + flat kernel (data-parallel path) and
inside a hipGraph replay."""
import copy

import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _net():
    torch.manual_seed(0)
    # 300 x 300 = 90000 elements: more than one 65536-element chunk
    return nn.Sequential(nn.Linear(37, 300), nn.LayerNorm(300), nn.GELU(), nn.Linear(300, 300),
                         nn.GELU(), nn.Linear(300, 5)).to(DEV)


def _groups(m, wd):
    decay = [p for n, p in m.named_parameters() if "bias" not in n]
    no_decay = [p for n, p in m.named_parameters() if "bias" in n]
    return [{"params": decay, "weight_decay": wd}, {"params": no_decay, "weight_decay": 0.0}]


@pytest.mark.parametrize("gather", [False, True])
def test_flat_adamw_matches_torch_adamw_with_value_clip(gather):
    from situation3d_amd.optim import FlatAdamW
    a = _net()
    b = copy.deepcopy(a)
    ref = torch.optim.AdamW(_groups(a, 0.05), lr=1e-2, betas=(0.9, 0.999), eps=1e-8)
    opt = FlatAdamW(_groups(b, 0.05), lr=1e-2, betas=(0.9, 0.999), eps=1e-8, clip_value=0.05)
    g = torch.Generator().manual_seed(1)
    for step in range(25):
        x = torch.randn(16, 37, generator=g).to(DEV)
        y = torch.randn(16, 5, generator=g).to(DEV)
        ref.zero_grad(set_to_none=True)
        (10 * (a(x) - y).pow(2).mean()).backward()
        nn.utils.clip_grad_value_(a.parameters(), 0.05)
        ref.step()
        (10 * (b(x) - y).pow(2).mean()).backward()
        if gather:
            opt.gather_grads()          # what the data-parallel step does before the all-reduce
            assert all(p.grad is None for p in b.parameters())
        opt.step()
        assert all(p.grad is None for p in b.parameters())  # zero_grad(set_to_none=True) semantics
    # 25 steps of lr = 1e-2 move a weight by up to 0.25; the two implementations differ in the
    # rounding of the bias corrections (float powf vs Python double) => a few 1e-6 after 25 steps
    for (n, p), q in zip(a.named_parameters(), b.parameters()):
        torch.testing.assert_close(q, p, rtol=1e-4, atol=2e-5, msg=lambda m: n + ": " + m)
    assert list(b.state_dict()) == list(a.state_dict())  # keys / shapes untouched by the flat storage


def test_flat_adamw_skips_parameters_without_gradient():
    from situation3d_amd.optim import FlatAdamW
    a = _net()
    extra = nn.Linear(4, 4).to(DEV)
    before = extra.weight.detach().clone()
    opt = FlatAdamW([{"params": list(a.parameters()) + list(extra.parameters()), "weight_decay": 0.1}],
                    lr=1e-2)
    (a(torch.randn(4, 37, device=DEV)).sum()).backward()
    opt.step()
    assert torch.equal(extra.weight, before)  # torch.optim.AdamW also skips p.grad is None


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
