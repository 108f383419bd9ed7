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


@pytest.mark.parametrize("direction", ["torch->flat", "flat->torch", "flat->flat"])
def test_flat_adamw_state_dict_round_trip_with_torch_adamw(direction):
    """lib/solver.py:657 saves optimizer.state_dict(), situation3d/train/train.py:262 loads it: a
    save / resume cycle must carry both moments and the step count, in torch.optim.AdamW's layout."""
    from situation3d_amd.optim import FlatAdamW
    g = torch.Generator().manual_seed(3)
    xs = [torch.randn(16, 37, generator=g).to(DEV) for _ in range(8)]
    ys = [torch.randn(16, 5, generator=g).to(DEV) for _ in range(8)]

    def make(kind, net):
        if kind == "torch":
            return torch.optim.AdamW(_groups(net, 0.05), lr=1e-2)
        return FlatAdamW(_groups(net, 0.05), lr=1e-2, clip_value=0.0)

    def run(net, opt, lo, hi):
        for i in range(lo, hi):
            opt.zero_grad(set_to_none=True)
            (10 * (net(xs[i]) - ys[i]).pow(2).mean()).backward()
            opt.step()

    src_kind, dst_kind = direction.split("->")
    # uninterrupted reference run: 8 steps of torch AdamW
    ref_net = _net()
    ref = torch.optim.AdamW(_groups(ref_net, 0.05), lr=1e-2)
    run(ref_net, ref, 0, 8)
    # 4 steps with the source optimizer, checkpoint, 4 more with a FRESH destination optimizer
    net = _net()
    src = make(src_kind, net)
    run(net, src, 0, 4)
    ckpt = {"model": copy.deepcopy(net.state_dict()), "optimizer": copy.deepcopy(src.state_dict())}
    assert len(ckpt["optimizer"]["state"]) == len(list(net.parameters()))
    st0 = ckpt["optimizer"]["state"][0]
    assert set(st0) >= {"step", "exp_avg", "exp_avg_sq"} and float(st0["step"]) == 4.0
    net2 = _net()
    net2.load_state_dict(ckpt["model"])
    dst = make(dst_kind, net2)
    dst.load_state_dict(ckpt["optimizer"])
    run(net2, dst, 4, 8)
    for (n, p), q in zip(ref_net.named_parameters(), net2.parameters()):
        torch.testing.assert_close(q, p, rtol=1e-4, atol=1e-5, msg=lambda m: n + ": " + m)


def test_flat_adamw_gathered_and_bucketed_paths_leave_dead_parameters_alone():
    """A parameter without a gradient is neither decayed nor stepped on ANY path (torch.optim.AdamW skips
    p.grad is None; the reference's DDP runs with find_unused_parameters=True, runner_base.py:91-93)."""
    from situation3d_amd.ddp import GradBucketReducer
    from situation3d_amd.optim import FlatAdamW
    for mode in ("gathered", "bucketed"):
        a = _net()
        extra = nn.Linear(300, 300).to(DEV)       # > one chunk, never used in the loss
        ref = copy.deepcopy(a)
        opt = FlatAdamW([{"params": list(a.parameters()) + list(extra.parameters()), "weight_decay": 0.1}],
                        lr=1e-2, clip_value=1.0)
        before = extra.weight.detach().clone()
        ropt = torch.optim.AdamW(ref.parameters(), lr=1e-2, weight_decay=0.1)
        reducer = GradBucketReducer.from_flat(opt.flat_grad_buffers(), bucket_bytes=1 << 18)
        assert len(reducer.buckets) > 2
        for _ in range(3):
            x = torch.randn(4, 37, device=DEV)
            a(x).sum().backward()
            opt.gather_grads()
            if mode == "gathered":
                opt.step()
            else:
                opt.step_after(reducer)
            ropt.zero_grad(set_to_none=True)
            ref(x).sum().backward()
            nn.utils.clip_grad_value_(ref.parameters(), 1.0)
            ropt.step()
        assert torch.equal(extra.weight, before), mode
        for p, q in zip(ref.parameters(), a.parameters()):
            torch.testing.assert_close(q, p, rtol=1e-4, atol=1e-5)


def test_flat_adamw_learning_rate_schedule_reaches_captured_replays():
    """StepLR-style changes of param_groups['lr'] (lib/solver.py:239-247) must be honoured by a captured
    step: the kernels read the learning rate from a device scalar that sync_lr() refreshes."""
    from situation3d_amd.optim import FlatAdamW
    work = torch.cuda.Stream()
    with torch.cuda.stream(work):
        a = _net()
        b = copy.deepcopy(a)
        ref = torch.optim.AdamW(_groups(a, 0.01), lr=1e-2)
        opt = FlatAdamW(_groups(b, 0.01), lr=1e-2, clip_value=0.0)
        sched_ref = torch.optim.lr_scheduler.StepLR(ref, step_size=2, gamma=0.1)
        sched = torch.optim.lr_scheduler.StepLR(opt, step_size=2, gamma=0.1)
        x = torch.randn(8, 37, device=DEV)
        y = torch.randn(8, 5, device=DEV)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=work):
            (b(x) - y).pow(2).mean().backward()
            opt.step()
        # the capture itself executed nothing: both models are still identical
        for _ in range(6):
            ref.zero_grad(set_to_none=True)
            (a(x) - y).pow(2).mean().backward()
            ref.step()
            sched_ref.step()
            opt.sync_lr()
            graph.replay()
            sched.step()
        torch.cuda.synchronize()
        assert opt.param_groups[0]["lr"] == pytest.approx(1e-5)
        for p, q in zip(a.parameters(), b.parameters()):
            torch.testing.assert_close(q, p, rtol=1e-4, atol=1e-6)
