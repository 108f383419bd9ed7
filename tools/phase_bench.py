"""Phase timings of the B=8 / 40k step (developer tool): where does the GPU time go?

Eager launches on one work stream, GPU time per phase from events around each phase; host-bound
phases show up as wall >> sum of kernel time, so the second number per phase is the rocprof-style
busy time measured by re-running the phase inside a hipGraph where possible (not done here).
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from situation3d_amd.geometry import GeometryPlan  # noqa: E402
from situation3d_amd.model import SIG3DQFormer  # noqa: E402
from situation3d_amd.trainer import build_optimizer, get_loss  # noqa: E402


def timed(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def graph_timed(fn, iters=10):
    """Capture fn once, replay: GPU-side time without host launch overhead."""
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=torch.cuda.current_stream()):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    dev = torch.device("cuda:0")
    work = torch.cuda.Stream()
    with torch.cuda.stream(work):
        torch.manual_seed(0)
        model = SIG3DQFormer(num_answers=bench.NUM_ANSWERS).to(dev).train()
        batch = bench.synthetic_batch(8, 40000, 1, dev)
        pc = batch["point_clouds"]
        xyz = pc[..., :3].contiguous()
        feats = pc[..., 3:].transpose(1, 2).contiguous()
        plan = GeometryPlan(8, 40000, model.encoder.LEVELS, dev)
        print("geometry chain            %.3f ms" % graph_timed(lambda: plan.compute(xyz)))

        enc = model.encoder
        levels = [enc.sa1, enc.sa2, enc.sa3, enc.sa4]
        cur_xyz, cur_f = xyz, feats
        for i, sa in enumerate(levels):
            g = plan.level(i)
            x_in, f_in = cur_xyz, cur_f.detach().requires_grad_(i > 0)

            def fwd():
                return sa(x_in, f_in, geometry=g)

            def fwdbwd():
                o = sa(x_in, f_in, geometry=g)[1]
                grads = torch.autograd.grad(o.sum(), [p for p in sa.parameters()] + ([f_in] if i > 0 else []))
                return grads

            print("SA%d fwd %.3f ms | fwd+bwd %.3f ms (graph replay)" % (i + 1, graph_timed(fwd), graph_timed(fwdbwd)))
            with torch.no_grad():
                cur_xyz, cur_f, _ = sa(cur_xyz, cur_f, geometry=g)

        tokens = torch.randn(8, 256, 256, device=dev, requires_grad=True)
        q = batch["q_feat"]
        ones = torch.ones(8, 32, dtype=torch.long, device=dev)
        att = torch.cat([ones, q["attention_mask"]], 1)

        def qf():
            return model.Qformer.bert(query_embeds=model.query_tokens.expand(8, -1, -1), input_ids=q["input_ids"],
                                      attention_mask=att, encoder_hidden_states=tokens, return_dict=True).last_hidden_state

        def qfb():
            o = qf()
            return torch.autograd.grad(o.sum(), [p for p in model.Qformer.parameters()] + [tokens, model.query_tokens],
                                       allow_unused=True)

        print("Q-Former fwd %.3f ms | fwd+bwd %.3f ms (graph replay)" % (graph_timed(qf), graph_timed(qfb)))
        opt = build_optimizer(model)

        def full():
            opt.zero_grad(set_to_none=True)
            d = dict(batch)
            d["geometry_plan"] = plan
            out = model(d)
            loss, _ = get_loss(out)
            loss.backward()
            torch.nn.utils.clip_grad_value_(model.parameters(), 1.0)
            opt.step()

        for _ in range(2):
            full()
        print("full step w/ plan (graph)  %.3f ms" % graph_timed(full))

        def optstep():
            torch.nn.utils.clip_grad_value_(model.parameters(), 1.0)
            opt.step()
        print("clip + AdamW (graph)       %.3f ms" % graph_timed(optstep))


if __name__ == "__main__":
    main()
