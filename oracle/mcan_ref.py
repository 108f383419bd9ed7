"""CPU restatement (functional torch, float64-capable) of the MCAN blocks of the native SIG3D head
(situation3d/models/mcan_sqa_module.py), driven by a state_dict.  TEST INFRASTRUCTURE ONLY.

Follows  LayerNorm :57-69, AttFlat :74-108, MHAtt :113-178, FFN :184-199, SA :205-226, SGA :232-263,
MCAN_ED :269-286 (eval mode: every dropout is the identity).
Pinned by tests/golden/mcan_golden.npz, produced by importing the reference module itself
(tests/golden/make_mcan_golden.py); the reference has no tests of its own for these blocks.
"""
import math

import torch


def _lin(sd, prefix, x):
    return x @ sd[prefix + ".weight"].T + sd[prefix + ".bias"]


def _gelu(x):
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def mcan_norm(sd, prefix, x, eps=1e-6):
    mean = x.mean(-1, keepdim=True)
    std = ((x - mean) ** 2).sum(-1, keepdim=True).div(x.shape[-1] - 1).sqrt()
    return sd[prefix + ".a_2"] * (x - mean) / (std + eps) + sd[prefix + ".b_2"]


def mlp(sd, prefix, x):
    return _lin(sd, prefix + ".linear", _gelu(_lin(sd, prefix + ".fc.linear", x)))


def mhatt(sd, prefix, v, k, q, mask, heads):
    b, hidden = q.shape[0], q.shape[-1]
    d = hidden // heads
    split = lambda t: t.reshape(b, -1, heads, d).transpose(1, 2)
    vv, kk, qq = (split(_lin(sd, prefix + ".linear_" + n, t)) for n, t in (("v", v), ("k", k), ("q", q)))
    scores = qq @ kk.transpose(-2, -1) / math.sqrt(d)
    if mask is not None:
        scores = scores.masked_fill(mask, -1e9)
    ctx = torch.softmax(scores, -1) @ vv
    return _lin(sd, prefix + ".linear_merge", ctx.transpose(1, 2).reshape(b, -1, hidden))


def sa(sd, prefix, x, x_mask, heads):
    x = mcan_norm(sd, prefix + "norm1", x + mhatt(sd, prefix + "mhatt", x, x, x, x_mask, heads))
    return mcan_norm(sd, prefix + "norm2", x + mlp(sd, prefix + "ffn.mlp", x))


def sga(sd, prefix, x, y, x_mask, y_mask, heads):
    x = mcan_norm(sd, prefix + "norm1", x + mhatt(sd, prefix + "mhatt1", x, x, x, x_mask, heads))
    x = mcan_norm(sd, prefix + "norm2", x + mhatt(sd, prefix + "mhatt2", y, y, x, y_mask, heads))
    return mcan_norm(sd, prefix + "norm3", x + mlp(sd, prefix + "ffn.mlp", x))


def att_flat(sd, prefix, x, x_mask):
    att = mlp(sd, prefix + "mlp", x)
    if x_mask is not None:
        att = att.masked_fill(x_mask.squeeze(1).squeeze(1).unsqueeze(2), -1e9)
    att = torch.softmax(att, dim=1)
    pooled = torch.cat([(att[:, :, i:i + 1] * x).sum(1) for i in range(att.shape[-1])], dim=1)
    return _lin(sd, prefix + "linear_merge", pooled), att


def mcan_ed(sd, x, y, x_mask, y_mask, heads, layers):
    for i in range(layers):
        x = sa(sd, "enc_list.%d." % i, x, x_mask, heads)
    for i in range(layers):
        y = sga(sd, "dec_list.%d." % i, y, x, y_mask, x_mask, heads)
    return x, y
