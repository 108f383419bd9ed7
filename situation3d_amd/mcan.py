"""The MCAN blocks of the native SIG3D head (situation3d/models/mcan_sqa_module.py) over the HIP path:
same class names, constructor arguments and state_dict keys, so a reference checkpoint loads unchanged and
`sqa_module.py:185-188,241-243` can build its `SA` / `SGA` / `AttFlat` from here.

What runs where:
  * MHAtt: three projection GEMMs (library), `sig3d_attention_fwd/bwd` with head size 768/8 = 96 (exact-f32
    MFMA, streaming softmax, dropout on the probabilities inside the kernel; mcan_sqa_module.py:164-178),
    merge GEMM;
  * `x = norm(x + dropout(sublayer(x)))` (mcan_sqa_module.py:216-224, 249-261): the merge / second FFN GEMM
    runs without its bias and `sig3d_dropout_add_mcan_norm_fwd/bwd` does bias + dropout + residual + the
    MCAN LayerNorm (unbiased std, eps added to the std, :57-69) in one kernel each way;
  * AttFlat (:74-108) and the GELU between the two FFN GEMMs are short row-wise torch ops.
Masks are the reference's boolean (B,1,1,N) key masks, True = padded.  masked_fill(mask, -1e9) is realised as
the additive mask -1e9: |score| < 32 is absorbed by fp32 rounding (ulp(1e9) = 64), so the masked scores are
exactly -1e9 as in the reference -- including rows whose keys are ALL masked (uniform attention).
Dropout uses this build's counter-hash stream (same distribution as torch's, different bits).
"""
import math

import torch
import torch.nn as nn

from .qformer import _AttentionFn, _DropoutAddLayerNormFn, _LinearFn, _call_ids, linear


class FC(nn.Module):
    def __init__(self, in_size, out_size, pdrop=0., use_gelu=True):
        super().__init__()
        self.pdrop = pdrop
        self.use_gelu = use_gelu
        self.linear = nn.Linear(in_size, out_size)
        if use_gelu:
            self.gelu = nn.GELU()
        if pdrop > 0:
            self.dropout = nn.Dropout(pdrop)

    def forward(self, x):
        x = linear(self.linear, x)
        if self.use_gelu:
            x = self.gelu(x)
        if self.pdrop > 0:
            x = self.dropout(x)
        return x


class MLP(nn.Module):
    def __init__(self, in_size, mid_size, out_size, pdrop=0., use_gelu=True):
        super().__init__()
        self.fc = FC(in_size, mid_size, pdrop=pdrop, use_gelu=use_gelu)
        self.linear = nn.Linear(mid_size, out_size)

    def forward(self, x):
        return linear(self.linear, self.fc(x))


class LayerNorm(nn.Module):
    """mcan_sqa_module.py:57-69: a_2 * (x - mean) / (std + eps) + b_2, std unbiased."""

    def __init__(self, size, eps=1e-6):
        super().__init__()
        self.eps = eps
        self.a_2 = nn.Parameter(torch.ones(size))
        self.b_2 = nn.Parameter(torch.zeros(size))

    def forward(self, x):
        _require_hip(x, self.a_2.shape[0])
        return _DropoutAddLayerNormFn.apply(x, None, torch.zeros_like(x), self.a_2, self.b_2, 0.0,
                                            float(self.eps), 0, True)


def _require_hip(x, width):
    """No CPU fallback (like the rest of the HIP path): float32 GPU tensors, rows of 2..1024 features."""
    if not x.is_cuda:
        raise RuntimeError("CPU not supported")
    if x.dtype != torch.float32 or not 2 <= width <= 1024:
        raise RuntimeError("the MCAN blocks run in float32 with 2..1024 features per row")


def _residual_norm(norm, dropout, dense, hidden, residual, call_id):
    """norm(residual + dropout(dense(hidden))): GEMM without bias + one fused kernel."""
    _require_hip(hidden, norm.a_2.shape[0])
    y = _LinearFn.apply(hidden, dense.weight, None)
    p = dropout.p if dropout.training else 0.0
    return _DropoutAddLayerNormFn.apply(y, dense.bias, residual, norm.a_2, norm.b_2, float(p), float(norm.eps),
                                        call_id, True)


class AttFlat(nn.Module):
    def __init__(self, hidden_size, flat_mlp_size=512, flat_glimpses=1, flat_out_size=1024, pdrop=0.1):
        super().__init__()
        self.mlp = MLP(in_size=hidden_size, mid_size=flat_mlp_size, out_size=flat_glimpses, pdrop=pdrop,
                       use_gelu=True)
        self.flat_glimpses = flat_glimpses
        self.linear_merge = nn.Linear(hidden_size * flat_glimpses, flat_out_size)

    def forward(self, x, x_mask):
        """x (B,N,C), x_mask (B,1,1,N) bool or None -> (merged (B,flat_out), token weights (B,N,glimpses))."""
        logits = self.mlp(x)                                              # (B, N, G)
        if x_mask is not None:
            logits = logits.masked_fill(x_mask.reshape(x.shape[0], -1, 1), -1e9)
        weights = torch.softmax(logits, dim=1)                            # over the tokens
        pooled = torch.einsum("bng,bnc->bgc", weights, x).flatten(1)      # glimpse-major, like the cat
        return linear(self.linear_merge, pooled), weights


class MHAtt(nn.Module):
    def __init__(self, hidden_size, num_heads=8, pdrop=0.1):
        super().__init__()
        self.linear_v = nn.Linear(hidden_size, hidden_size)
        self.linear_k = nn.Linear(hidden_size, hidden_size)
        self.linear_q = nn.Linear(hidden_size, hidden_size)
        self.linear_merge = nn.Linear(hidden_size, hidden_size)
        self.hidden_size = hidden_size
        self.num_heads = num_heads
        self.head_hidden_size = int(hidden_size / num_heads)
        self.dropout = nn.Dropout(pdrop)
        self._call_id = next(_call_ids)

    def attend(self, v, k, q, mask):
        """Everything of forward() up to (excluding) linear_merge: (B, Nq, hidden) context."""
        _require_hip(q, self.hidden_size)
        if self.head_hidden_size not in (64, 96):
            raise RuntimeError("attention head size %d: the HIP kernels are built for 64 and 96"
                               % self.head_hidden_size)
        vv, kk, qq = linear(self.linear_v, v), linear(self.linear_k, k), linear(self.linear_q, q)
        add_mask = None
        if mask is not None:   # (B,1,1,Nk) bool, True = masked
            add_mask = mask.reshape(mask.shape[0], -1).to(torch.float32) * -1e9
        p = self.dropout.p if self.training else 0.0
        return _AttentionFn.apply(qq, kk, vv, add_mask, self.num_heads, 1.0 / math.sqrt(self.head_hidden_size),
                                  float(p), self._call_id)

    def forward(self, v, k, q, mask):
        return linear(self.linear_merge, self.attend(v, k, q, mask))


class FFN(nn.Module):
    def __init__(self, hidden_size, pdrop=0.1):
        super().__init__()
        self.mlp = MLP(in_size=hidden_size, mid_size=int(hidden_size * 4), out_size=hidden_size, pdrop=pdrop,
                       use_gelu=True)

    def forward(self, x):
        return self.mlp(x)


class SA(nn.Module):
    def __init__(self, hidden_size, num_heads=8, pdrop=0.1):
        super().__init__()
        self.mhatt = MHAtt(hidden_size, num_heads, pdrop)
        self.ffn = FFN(hidden_size, pdrop)
        self.dropout1 = nn.Dropout(pdrop)
        self.norm1 = LayerNorm(hidden_size)
        self.dropout2 = nn.Dropout(pdrop)
        self.norm2 = LayerNorm(hidden_size)
        self._ids = (next(_call_ids), next(_call_ids))

    def forward(self, x, x_mask):
        x = _residual_norm(self.norm1, self.dropout1, self.mhatt.linear_merge,
                           self.mhatt.attend(x, x, x, x_mask), x, self._ids[0])
        return _residual_norm(self.norm2, self.dropout2, self.ffn.mlp.linear, self.ffn.mlp.fc(x), x, self._ids[1])


class SGA(nn.Module):
    def __init__(self, hidden_size, num_heads=8, pdrop=0.1):
        super().__init__()
        self.mhatt1 = MHAtt(hidden_size, num_heads, pdrop)
        self.mhatt2 = MHAtt(hidden_size, num_heads, pdrop)
        self.ffn = FFN(hidden_size, pdrop)
        self.dropout1 = nn.Dropout(pdrop)
        self.norm1 = LayerNorm(hidden_size)
        self.dropout2 = nn.Dropout(pdrop)
        self.norm2 = LayerNorm(hidden_size)
        self.dropout3 = nn.Dropout(pdrop)
        self.norm3 = LayerNorm(hidden_size)
        self._ids = (next(_call_ids), next(_call_ids), next(_call_ids))

    def forward(self, x, y, x_mask, y_mask):
        x = _residual_norm(self.norm1, self.dropout1, self.mhatt1.linear_merge,
                           self.mhatt1.attend(x, x, x, x_mask), x, self._ids[0])
        x = _residual_norm(self.norm2, self.dropout2, self.mhatt2.linear_merge,
                           self.mhatt2.attend(y, y, x, y_mask), x, self._ids[1])
        return _residual_norm(self.norm3, self.dropout3, self.ffn.mlp.linear, self.ffn.mlp.fc(x), x, self._ids[2])


class MCAN_ED(nn.Module):
    def __init__(self, hidden_size, num_heads=8, num_layers=6, pdrop=0.1):
        super().__init__()
        self.enc_list = nn.ModuleList([SA(hidden_size, num_heads, pdrop) for _ in range(num_layers)])
        self.dec_list = nn.ModuleList([SGA(hidden_size, num_heads, pdrop) for _ in range(num_layers)])

    def forward(self, x, y, x_mask, y_mask):
        for enc in self.enc_list:
            x = enc(x, x_mask)
        for dec in self.dec_list:
            y = dec(y, x, y_mask, x_mask)
        return x, y
