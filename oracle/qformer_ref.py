"""oracle/qformer_ref.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Functional (no nn.Module) fp32 restatement on plain torch CPU ops of the reference's Q-Former
forward pass, 3DLLM_BLIP2-base/lavis/models/blip2_models/Qformer.py, driven by a reference
state_dict (name -> tensor).  Differentiable through torch autograd, so it is also the checker
for gradients.  Eval-mode semantics (all dropouts are identity), like the golden vectors.

Pinned by tests/golden/qformer_small.npz, which was produced by importing the reference's own
Qformer.py in the build container (tests/golden/make_golden.py) -- see
tests/test_oracle_golden.py.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import this file.
"""
import math

import torch
import torch.nn.functional as F


def _linear(x, sd, prefix):
    return F.linear(x, sd[prefix + ".weight"], sd[prefix + ".bias"])


def _layer_norm(x, sd, prefix, eps):
    return F.layer_norm(x, (x.shape[-1],), sd[prefix + ".weight"], sd[prefix + ".bias"], eps)


def _heads(x, num_heads):
    """Qformer.py:140-147 transpose_for_scores: (B,N,H*d) -> (B,H,N,d)"""
    b, n, hd = x.shape
    return x.view(b, n, num_heads, hd // num_heads).permute(0, 2, 1, 3)


def self_attention(hidden, sd, prefix, num_heads, additive_mask, kv_source=None):
    """Qformer.py:150-232 (absolute position embeddings, no head mask, eval-mode dropout):
    additive_mask is the (B,1,1,Nk) extended mask."""
    src = hidden if kv_source is None else kv_source
    key = _heads(_linear(src, sd, prefix + ".key"), num_heads)        # :164-176
    value = _heads(_linear(src, sd, prefix + ".value"), num_heads)
    query = _heads(_linear(hidden, sd, prefix + ".query"), num_heads)  # :178-180
    scores = torch.matmul(query, key.transpose(-1, -2))                # :185
    scores = scores / math.sqrt(query.shape[-1])                       # :207
    if additive_mask is not None:
        scores = scores + additive_mask                                # :210
    probs = torch.softmax(scores, dim=-1)                              # :213
    ctx = torch.matmul(probs, value)                                   # :223
    b, h, n, d = ctx.shape
    return ctx.permute(0, 2, 1, 3).contiguous().view(b, n, h * d)      # :225-227


def attention_block(hidden, sd, prefix, num_heads, additive_mask, eps, kv_source=None):
    """BertAttention = BertSelfAttention + BertSelfOutput (Qformer.py:235-299)."""
    ctx = self_attention(hidden, sd, prefix + ".self", num_heads, additive_mask, kv_source)
    dense = _linear(ctx, sd, prefix + ".output.dense")
    return _layer_norm(dense + hidden, sd, prefix + ".output.LayerNorm", eps)


def ffn(x, sd, inter, out, eps):
    """BertIntermediate (erf GELU) + BertOutput (Qformer.py:302-328, 420-428)."""
    h = F.gelu(_linear(x, sd, inter + ".dense"))
    return _layer_norm(_linear(h, sd, out + ".dense") + x, sd, out + ".LayerNorm", eps)


def bert_model(sd, cfg, query_embeds=None, input_ids=None, attention_mask=None,
               encoder_hidden_states=None, encoder_attention_mask=None, prefix="",
               return_all=False):
    """BertModel.forward, encoder mode (Qformer.py:734-869) -> last_hidden_state
    (and the per-layer hidden states when return_all)."""
    eps = cfg["layer_norm_eps"]
    nh = cfg["num_attention_heads"]
    # BertEmbeddings.forward (Qformer.py:70-98)
    if input_ids is not None:
        t = input_ids.shape[1]
        emb = F.embedding(input_ids, sd[prefix + "embeddings.word_embeddings.weight"])
        pos = sd[prefix + "embeddings.position_embeddings.weight"][:t]
        emb = emb + pos.unsqueeze(0)
        if query_embeds is not None:
            emb = torch.cat((query_embeds, emb), dim=1)
    else:
        emb = query_embeds
    hidden = _layer_norm(emb, sd, prefix + "embeddings.LayerNorm", eps)
    query_length = query_embeds.shape[1] if query_embeds is not None else 0
    b, n = hidden.shape[:2]
    if attention_mask is None:
        attention_mask = torch.ones(b, n)
    ext = (1.0 - attention_mask[:, None, None, :].to(hidden.dtype)) * -10000.0  # :729-731
    enc_ext = None
    if encoder_hidden_states is not None:
        if encoder_attention_mask is None:
            encoder_attention_mask = torch.ones(encoder_hidden_states.shape[:2])
        enc_ext = (1.0 - encoder_attention_mask[:, None, None, :].to(hidden.dtype)) * -10000.0

    states = [hidden]
    for i in range(cfg["num_hidden_layers"]):  # BertLayer.forward (Qformer.py:350-418)
        lp = prefix + "encoder.layer.%d" % i
        att = attention_block(hidden, sd, lp + ".attention", nh, ext, eps)
        if query_length > 0:
            q_att = att[:, :query_length, :]
            if cfg["add_cross_attention"] and i % cfg["cross_attention_freq"] == 0:
                q_att = attention_block(q_att, sd, lp + ".crossattention", nh, enc_ext, eps,
                                        kv_source=encoder_hidden_states)
            out = ffn(q_att, sd, lp + ".intermediate_query", lp + ".output_query", eps)
            if att.shape[1] > query_length:
                out_t = ffn(att[:, query_length:, :], sd, lp + ".intermediate", lp + ".output", eps)
                out = torch.cat([out, out_t], dim=1)
        else:
            out = ffn(att, sd, lp + ".intermediate", lp + ".output", eps)
        hidden = out
        states.append(hidden)
    return (hidden, states) if return_all else hidden
