import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from situation3d_amd.model import SIG3DQFormer
from situation3d_amd.trainer import get_loss
dev = torch.device("cuda:0")
work = torch.cuda.Stream()
with torch.cuda.stream(work):
    torch.manual_seed(1234)
    model = SIG3DQFormer(num_answers=bench.NUM_ANSWERS).to(dev).train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout): m.p = 0.0
    batches = [bench.synthetic_batch(8, 40000, 1234 + i, dev) for i in range(3)]
    params = dict(model.named_parameters())
    def fwdbwd(batch):
        out = model(dict(batch)); loss, _ = get_loss(out); loss.backward(); return loss
    # eager reference grads per batch
    refs = []
    for b in batches:
        model.zero_grad(set_to_none=True); l = fwdbwd(b); torch.cuda.synchronize()
        refs.append(({k: p.grad.clone() for k, p in params.items() if p.grad is not None}, float(l)))
    static = {k: ({kk: vv.clone() for kk, vv in v.items()} if isinstance(v, dict) else v.clone()) for k, v in batches[0].items()}
    side = torch.cuda.current_stream()
    for _ in range(2):
        model.zero_grad(set_to_none=True); fwdbwd(static)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); model.zero_grad(set_to_none=True)
    with torch.cuda.graph(g, stream=side):
        sl = fwdbwd(static)
    torch.cuda.synchronize()
    for it in range(6):
        b = batches[it % 3]
        for k, v in b.items():
            if isinstance(v, dict):
                for kk, vv in v.items(): static[k][kk].copy_(vv)
            else: static[k].copy_(v)
        g.replay(); torch.cuda.synchronize()
        ref, rl = refs[it % 3]
        bad = []
        for k, p in params.items():
            if p.grad is None or k not in ref: continue
            r = ref[k]; d = (p.grad - r).abs().max().item(); s = r.abs().max().item() + 1e-12
            if not (d <= 1e-3 * s + 1e-6): bad.append((k, d, s))
        print(it, "loss graph %.5f eager %.5f" % (float(sl), rl), "mismatching grads:", len(bad), bad[:4], flush=True)
