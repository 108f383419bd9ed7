"""True concurrent timeline of one replayed training step (hipGraph, two branches).

rocprofv3 serialises the two branches of the graph (a 9 ms step takes 24 ms under --kernel-trace), so the
interference between the geometry-prefetch branch and the main branch cannot be read off a profile.  This tool
captures wall-clock marks (sig3d_timestamp, situation3d_amd/timeline.py) INTO the graph: at the phase
boundaries of the main branch (module forward hooks; tensor hooks for the backward pass) and after every
stage of the geometry branch.  It prints the phases of the main branch with and without the geometry branch
beside them.

python tools/branch_timeline.py [--no-prefetch] [--steps 20]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import bench  # noqa: E402
from situation3d_amd import timeline  # noqa: E402
from situation3d_amd.graph_step import GraphedTrainStep  # noqa: E402
from situation3d_amd.model import SIG3DQFormer  # noqa: E402
from situation3d_amd.trainer import build_optimizer  # noqa: E402


def instrument(model):
    mods = [("sa%d" % i, getattr(model.encoder, "sa%d" % i)) for i in (1, 2, 3, 4)]
    mods += [("encoder", model.encoder), ("pos_embed", model.pos_embed), ("qformer", model.Qformer)]
    mods += [("qf.layer%d" % i, l) for i, l in enumerate(model.Qformer.bert.encoder.layer) if i % 3 == 2]

    def tensors(o):
        if torch.is_tensor(o):
            yield o
        elif isinstance(o, (list, tuple)):
            for x in o:
                yield from tensors(x)
        elif isinstance(o, dict):
            for x in o.values():
                yield from tensors(x)
        elif hasattr(o, "__dict__"):
            for x in vars(o).values():
                yield from tensors(x)

    for name, mod in mods:
        def hook(m, i, o, name=name):
            timeline.mark("main:fwd %s done" % name)
            for t in tensors(o):
                if t.requires_grad and t.is_floating_point():
                    t.register_hook(lambda g, name=name: timeline.mark("main:bwd reaches %s" % name))
                    break
        mod.register_forward_hook(hook)


def run(prefetch, steps, depth=1):
    dev = torch.device("cuda:0")
    torch.manual_seed(1234)
    model = SIG3DQFormer(num_answers=bench.NUM_ANSWERS).to(dev).train()
    opt = build_optimizer(model, name="flat_adamw")
    batches = [bench.synthetic_batch(bench.BATCH, bench.N_POINTS, 1234 + i, dev) for i in range(4)]
    tl = timeline.enable(dev)
    instrument(model)
    work = torch.cuda.Stream(dev)
    with torch.cuda.stream(work):
        g = GraphedTrainStep(model, opt, batches[0], prefetch_geometry=prefetch, prefetch_depth=depth)
        up = lambda i: [batches[(i + 1 + k) % 4] for k in range(g.prefetch_depth)]
        for i in range(5):
            g(batches[i % 4], upcoming=up(i))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            g(batches[i % 4], upcoming=up(i))
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
    marks = tl.read()
    timeline.disable()
    return ms, marks


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--no-prefetch", action="store_true")
    ap.add_argument("--depth", type=int, default=1, help="geometry chains in flight")
    a = ap.parse_args()
    ms, marks = run(not a.no_prefetch, a.steps, a.depth)
    print("prefetch=%s  %.3f ms/step (marks included)" % (not a.no_prefetch, ms))
    base = dict(marks).get("main:start", 0.0)
    prev = {"main": base, "geo": base}
    for name, us in marks:
        br = name.split(":")[0]
        print("%9.1f us  (+%7.1f)  %s%s" % (us - base, us - prev.get(br, base), "" if br == "main" else "        ", name))
        prev[br] = us
