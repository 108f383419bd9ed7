"""CPU suite: oracle/mcan_ref.py against vectors produced by the REFERENCE's mcan_sqa_module.py
(tests/golden/make_mcan_golden.py).  Tolerance 1e-5 (same float32 torch-CPU arithmetic, different op order)."""
import os

import numpy as np
import torch

from oracle import mcan_ref as ref

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mcan_golden.npz")
HEADS = 2


def load(prefix):
    g = np.load(GOLD, allow_pickle=False)
    sd = {k[len(prefix) + 3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(prefix + "sd.")}
    rest = {k[len(prefix):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(prefix) and ".sd." not in k[len(prefix) - 1:]}
    return sd, rest


def close(a, b, tol=1e-5):
    assert (a - b).abs().max().item() <= tol * max(1.0, b.abs().max().item()), (a - b).abs().max().item()


def test_sa_sga_flat_ed_match_reference():
    sd, t = load("sa.")
    close(ref.sa(sd, "", t["x"], t["mask"], HEADS), t["out"])
    sd, t = load("sga.")
    close(ref.sga(sd, "", t["x"], t["y"], None, t["ymask"], 1), t["out"])
    sd, t = load("flat.")
    o, att = ref.att_flat(sd, "", t["x"], t["mask"])
    close(o, t["out"])
    close(att, t["att"])
    sd, t = load("ed.")
    ox, oy = ref.mcan_ed(sd, t["x"], t["y"], t["xmask"], t["ymask"], 1, 2)
    close(ox, t["out_x"])
    close(oy, t["out_y"])


def test_oracle_gradients_match_reference():
    sd, t = load("sa.")
    sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    x = t["x"].clone().requires_grad_(True)
    (ref.sa(sd, "", x, t["mask"], HEADS) * t["g"]).sum().backward()
    close(x.grad, t["dx"])
    close(sd["mhatt.linear_q.weight"].grad, t["dWq"])
    close(sd["norm1.a_2"].grad, t["da2"])
    close(sd["mhatt.linear_merge.bias"].grad, t["db_merge"])
