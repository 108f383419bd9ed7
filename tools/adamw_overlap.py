"""Would the flat AdamW update (HBM-bound, ~0.9 ms) hide under the NEXT step's point-encoder forward (small
latency-bound kernels in compact mode)?  Two-stream microbenchmark."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from situation3d_amd import gemm_tuning
from situation3d_amd.geometry import GeometryPlan
from situation3d_amd.model import SIG3DQFormer
from situation3d_amd.trainer import build_optimizer, train_step
dev = torch.device("cuda", 0)
gemm_tuning.enable(tune_missing=True)
torch.manual_seed(0)
model = SIG3DQFormer(num_answers=bench.NUM_ANSWERS).to(dev).train()
opt = build_optimizer(model, name="flat_adamw")
batch = bench.synthetic_batch(bench.BATCH, bench.N_POINTS, 7, dev)
main, side = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
with torch.cuda.stream(main):
    for _ in range(2):
        train_step(model, opt, dict(batch))
    pc = batch["point_clouds"]
    xyz = pc[..., :3].contiguous(); feats = pc[..., 3:].transpose(1, 2).contiguous()
    plan = GeometryPlan(bench.BATCH, bench.N_POINTS, model.encoder.LEVELS, dev).compute(xyz)
    def enc():
        with torch.no_grad():
            return model.encoder(xyz, feats, plan)
    def adam():
        for p in model.parameters():
            p.grad = torch.zeros_like(p) if p.grad is None else p.grad
        opt.step()
    def timed(fn, n=10):
        for _ in range(2): fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(main)
        for _ in range(n): fn()
        e.record(main); torch.cuda.synchronize()
        return s.elapsed_time(e) / n
    g_enc, g_ad, g_both = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    for p in model.parameters():
        p.grad = torch.zeros_like(p)
    def adam_only():
        opt._upload()
    enc(); torch.cuda.synchronize()
    with gemm_tuning.no_tuning(), torch.cuda.graph(g_enc, stream=main):
        enc()
    t_enc = timed(g_enc.replay)
    import ctypes
    from situation3d_amd import _lib
    def launch_adam():
        g0 = opt.param_groups[0]; b1, b2 = g0["betas"]
        _lib.call("sig3d_adamw_table", len(opt._static), _lib.ptr(opt._table), _lib.ptr(opt._step), ctypes.c_float(g0["lr"]),
                  _lib.ptr(opt._lr_dev), ctypes.c_float(b1), ctypes.c_float(b2), ctypes.c_float(g0["eps"]), ctypes.c_float(opt.clip_value),
                  _lib.stream_ptr(dev))
    opt._upload()
    with torch.cuda.graph(g_ad, stream=main):
        launch_adam()
    t_ad = timed(g_ad.replay)
    with gemm_tuning.no_tuning(), torch.cuda.graph(g_both, stream=main):
        side.wait_stream(main)
        with torch.cuda.stream(side):
            launch_adam()
        enc()
        main.wait_stream(side)
    t_both = timed(g_both.replay)
print("encoder forward %.3f ms, AdamW %.3f ms, forked together %.3f ms (sum %.3f)" % (t_enc, t_ad, t_both, t_enc + t_ad))
