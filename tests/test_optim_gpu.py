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


def _tiny_sig3d():
    from situation3d_amd.model import SIG3DQFormer
    torch.manual_seed(0)
    small = dict(hidden_size=128, num_hidden_layers=4, num_attention_heads=2, intermediate_size=256,
                 max_position_embeddings=64, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    return SIG3DQFormer(num_answers=16, qformer_overrides=small, vocab_size=100).to(DEV).train()


def _sig3d_batch(seed, b=2, n=3000):
    g = torch.Generator().manual_seed(seed)
    xyz = torch.rand(b, n, 3, generator=g) * torch.tensor([8.0, 8.0, 3.0])
    pose = torch.cat([torch.rand(b, 3, generator=g), torch.tensor([[0.0, 0.0, 0.6, 0.8]]).expand(b, 4)], 1)
    ans = torch.zeros(b, 16)
    ans[:, 3] = 1.0
    return {"point_clouds": torch.cat([xyz, torch.rand(b, n, 3, generator=g)], -1).to(DEV),
            "auxiliary_task": pose.to(DEV),
            "q_feat": {"input_ids": torch.randint(1, 100, (b, 20), generator=g).to(DEV),
                       "attention_mask": torch.ones(b, 20, dtype=torch.long, device=DEV)},
            "answer_cat_scores": ans.to(DEV)}


@pytest.mark.parametrize("direction", ["torch->flat", "flat->torch"])
def test_build_optimizer_checkpoints_are_independent_of_the_flat_storage_order(direction):
    """trainer.build_optimizer(name="flat_adamw") lays the Q-Former's parameters out kind-major
    (all layers' query/key/value, then all out-projections, ...) -- far from named_parameters() order -- and every
    768 x 768 matrix has the same shape, so a checkpoint indexed by STORAGE position would silently hand the
    moments to the wrong parameters.  The groups keep named_parameters() order: a torch.optim.AdamW checkpoint
    (lib/solver.py:657, train.py:262) loads into FlatAdamW and back, compared per parameter NAME."""
    from situation3d_amd.trainer import build_optimizer, train_step
    src_name, dst_name = [{"torch": "adamw", "flat": "flat_adamw"}[k] for k in direction.split("->")]
    model = _tiny_sig3d()
    src = build_optimizer(model, name=src_name, lr=1e-3)
    for i in range(3):
        train_step(model, src, _sig3d_batch(10 + i))
    ckpt = copy.deepcopy(src.state_dict())
    # the moments of the source, by parameter name (torch's convention: position in the concatenated group lists)
    names = {id(p): n for n, p in model.named_parameters()}
    by_name, i = {}, 0
    for group in src.param_groups:
        for p in group["params"]:
            if i in ckpt["state"]:
                by_name[names[id(p)]] = ckpt["state"][i]
            i += 1
    assert len(by_name) > 60
    model2 = _tiny_sig3d()
    model2.load_state_dict(model.state_dict())
    dst = build_optimizer(model2, name=dst_name, lr=1e-3)
    # same group lists in both optimizers: named_parameters() order, whatever the flat layout is
    order = lambda opt, m: [[{id(p): n for n, p in m.named_parameters()}[id(p)] for p in g["params"]]
                            for g in opt.param_groups]
    assert order(src, model) == order(dst, model2)
    dst.load_state_dict(ckpt)
    back = dst.state_dict()
    names2 = {id(p): n for n, p in model2.named_parameters()}
    i, seen = 0, 0
    for group in dst.param_groups:
        for p in group["params"]:
            n = names2[id(p)]
            if n in by_name:
                st = back["state"][i]
                assert float(st["step"]) == 3.0
                torch.testing.assert_close(st["exp_avg"].to(DEV), by_name[n]["exp_avg"].to(DEV), rtol=0, atol=0,
                                           msg=lambda m: n + " exp_avg: " + m)
                torch.testing.assert_close(st["exp_avg_sq"].to(DEV), by_name[n]["exp_avg_sq"].to(DEV), rtol=0, atol=0)
                seen += 1
            i += 1
    assert seen == len(by_name)
    # and the next update uses each parameter's OWN moments: one step of both optimizers on identical synthetic
    # gradients (real backward passes differ in float-atomic summation order, which Adam's normalisation amplifies)
    ref = build_optimizer(model, name=src_name, lr=1e-3)
    ref.load_state_dict(ckpt)
    g = torch.Generator().manual_seed(5)
    for (n, p), q in zip(model.named_parameters(), model2.parameters()):
        if n in by_name:
            grad = (torch.rand(p.shape, generator=g) - 0.5).to(DEV) * 0.1      # inside the value clip
            p.grad, q.grad = grad.clone(), grad.clone()
    ref.step()
    dst.step()
    for (n, p), q in zip(model.named_parameters(), model2.parameters()):
        torch.testing.assert_close(q, p, rtol=1e-5, atol=2e-6, msg=lambda m: n + ": " + m)


def test_flat_adamw_load_state_dict_rejects_a_differently_ordered_checkpoint():
    from situation3d_amd.optim import FlatAdamW
    net = _net()
    opt = FlatAdamW(_groups(net, 0.05), lr=1e-2)
    (net(torch.randn(4, 37, device=DEV)).sum()).backward()
    opt.step()
    sd = opt.state_dict()
    sd["state"][0], sd["state"][2] = sd["state"][2], sd["state"][0]     # (300, 37) <-> (300, 300) moments swapped
    with pytest.raises(ValueError, match="orders its parameters differently"):
        opt.load_state_dict(sd)


@pytest.mark.parametrize("gather", [False, True])
def test_flat_adamw_with_a_bounded_grid_matches_the_full_grid(gather):
    """sig3d_adamw_table_bounded: k workgroups walking the chunk table produce the same bits as one workgroup per
    chunk (k = 1: a single workgroup does the whole model; k = 3: chunks interleave)."""
    from situation3d_amd.optim import FlatAdamW
    nets = [_net() for _ in range(3)]
    opts = [FlatAdamW(_groups(m, 0.05), lr=1e-2, clip_value=0.05, max_workgroups=k) for m, k in zip(nets, (None, 1, 3))]
    g = torch.Generator().manual_seed(5)
    for step in range(6):
        x = torch.randn(16, 37, generator=g).to(DEV)
        y = torch.randn(16, 5, generator=g).to(DEV)
        for m, o in zip(nets, opts):
            (10 * (m(x) - y).pow(2).mean()).backward()
            if gather:
                o.gather_grads()
            o.step()
    for m in nets[1:]:
        for p, q in zip(nets[0].parameters(), m.parameters()):
            assert torch.equal(p, q)
