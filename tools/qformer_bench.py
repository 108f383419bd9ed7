"""Q-Former fwd+bwd alone at the bench shape (B=8, 32 queries + 20 text tokens, 256 visual tokens):
run under rocprofv3 --kernel-trace --stats to see the kernel mix (developer tool)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from situation3d_amd.qformer import init_Qformer  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
qf, qt = init_Qformer(32, 256)
qf, qt = qf.to(dev).train(), torch.nn.Parameter(qt.detach().to(dev))
tokens = torch.randn(8, 256, 256, device=dev, requires_grad=True)
ids = torch.randint(1000, 30000, (8, 20), device=dev)
att = torch.ones(8, 52, dtype=torch.long, device=dev)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for i in range(iters + 2):
    out = qf.bert(query_embeds=qt.expand(8, -1, -1), input_ids=ids, attention_mask=att,
                  encoder_hidden_states=tokens, return_dict=True).last_hidden_state
    out.sum().backward()
torch.cuda.synchronize()
print("done")
