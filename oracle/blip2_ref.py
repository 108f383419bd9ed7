"""oracle/blip2_ref.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement of the position-embedding pre-stage of Blip2T5.forward
(3DLLM_BLIP2-base/lavis/models/blip2_models/blip2_t5.py:106-118), loop for loop:

    pc = samples["pc"].long()
    all_pcs = torch.zeros((pc_embeds.shape))
    for j in range(pc.shape[0]):
        pcs = [pos_embedding[pc[j][:, i]] for i in range(3)]
        all_pcs[j][:, :1407] = torch.cat(pcs, -1)
    pc_embeds = pc_embeds + 0.01 * all_pcs

The table (`pos_embedding`) is an input here: the package that builds it in the reference
(`positional_encodings`) is neither vendored nor pinned, so its layout is "parity unpinned".
Only tests/ may import this file.
"""
import torch


def add_position_embedding(pc_embeds, pc, pos_embedding, scale=0.01):
    pc = pc.long()
    all_pcs = torch.zeros(pc_embeds.shape)
    width = 3 * pos_embedding.shape[1]
    for j in range(pc.shape[0]):
        pcs = []
        for i in range(3):
            pcs.append(pos_embedding[pc[j][:, i]])
        all_pcs[j][:, :width] = torch.cat(pcs, -1)
    return pc_embeds + scale * all_pcs
