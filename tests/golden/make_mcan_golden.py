"""Generate tests/golden/mcan_golden.npz -- run ONLY in the build container (/root/reference).

    python tests/golden/make_mcan_golden.py

situation3d/models/mcan_sqa_module.py (torch only) is imported unmodified from /root/reference; SA, SGA,
AttFlat and MCAN_ED are built with seeded weights in eval mode (dropout off) and run on CPU.  Stored:
state_dicts, inputs, masks, outputs, and gradients of sum(out * G) w.r.t. the inputs and two weights.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference/situation3d")
from models import mcan_sqa_module as ref  # noqa: E402


def put(out, prefix, module):
    for k, v in module.state_dict().items():
        out[prefix + "sd." + k] = v.detach().numpy()


def mask(b, n, lens):
    m = torch.zeros(b, 1, 1, n, dtype=torch.bool)
    for i, l in enumerate(lens):
        m[i, :, :, l:] = True
    return m


def main():
    torch.manual_seed(20240918)
    out = {}
    hidden, heads = 192, 2          # head size 96, as in the 768 / 8 configuration of sqa_module.py
    b, nx, ny = 3, 37, 70
    x = torch.randn(b, nx, hidden, requires_grad=True)
    y = torch.randn(b, ny, hidden, requires_grad=True)
    xm, ym = mask(b, nx, [37, 20, 5]), mask(b, ny, [70, 64, 33])

    def randomise(m):
        for p in m.parameters():   # make a_2 / b_2 / biases non-trivial
            with torch.no_grad():
                p.add_(0.1 * torch.randn_like(p))
        return m.eval()

    sa = randomise(ref.SA(hidden, heads, 0.1))
    g = torch.randn(b, nx, hidden)
    o = sa(x, xm)
    (o * g).sum().backward()
    put(out, "sa.", sa)
    out.update({"sa.x": x.detach().numpy(), "sa.mask": xm.numpy(), "sa.out": o.detach().numpy(), "sa.g": g.numpy(),
                "sa.dx": x.grad.numpy().copy(), "sa.dWq": sa.mhatt.linear_q.weight.grad.numpy().copy(),
                "sa.da2": sa.norm1.a_2.grad.numpy().copy(), "sa.db_merge": sa.mhatt.linear_merge.bias.grad.numpy().copy()})
    x.grad = None

    # SGA / MCAN_ED with one head of 96 (fixture size); the two-head split is pinned by the SA case
    h1 = 96
    x = torch.randn(b, nx, h1, requires_grad=True)
    y = torch.randn(b, ny, h1, requires_grad=True)
    sga = randomise(ref.SGA(h1, 1, 0.1))
    o = sga(x, y, None, ym)     # the scene tokens are never masked (sqa_module.py:351-354)
    g = torch.randn(b, nx, h1)
    (o * g).sum().backward()
    put(out, "sga.", sga)
    out.update({"sga.x": x.detach().numpy(), "sga.y": y.detach().numpy(), "sga.ymask": ym.numpy(),
                "sga.out": o.detach().numpy(), "sga.g": g.numpy(), "sga.dx": x.grad.numpy().copy(),
                "sga.dy": y.grad.numpy().copy(), "sga.dWk2": sga.mhatt2.linear_k.weight.grad.numpy().copy(),
                "sga.dW_ffn2": sga.ffn.mlp.linear.weight.grad.numpy().copy()})
    x.grad = None
    y.grad = None

    flat = randomise(ref.AttFlat(h1, 64, 2, 128, 0.1))
    o, att = flat(y, ym)
    g = torch.randn(b, 128)
    (o * g).sum().backward()
    put(out, "flat.", flat)
    out.update({"flat.x": y.detach().numpy(), "flat.mask": ym.numpy(), "flat.out": o.detach().numpy(),
                "flat.att": att.detach().numpy(), "flat.g": g.numpy(), "flat.dx": y.grad.numpy().copy()})
    y.grad = None

    ed = randomise(ref.MCAN_ED(h1, 1, 2, 0.1))
    ox, oy = ed(x, y, xm, ym)
    put(out, "ed.", ed)
    out.update({"ed.x": x.detach().numpy(), "ed.y": y.detach().numpy(), "ed.xmask": xm.numpy(), "ed.ymask": ym.numpy(),
                "ed.out_x": ox.detach().numpy(), "ed.out_y": oy.detach().numpy()})

    path = os.path.join(HERE, "mcan_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
