"""Does the flat AdamW update (HBM-bound, ~0.9 ms) hide beside other work of the step on a forked hipGraph branch?
Work A: the point encoder's forward (small latency-bound kernels; the NEXT step's, in a real schedule).
Work B: the Q-Former's forward + backward (416-row GEMMs, small attention / LayerNorm launches).
Update forms: the full grid (one workgroup per 64 Ki-element chunk) and bounded grids (sig3d_adamw_table_bounded),
side stream at normal and at low priority.  lr = 0 so repeated updates leave the parameters alone."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from situation3d_amd import _lib, gemm_tuning
from situation3d_amd.geometry import GeometryPlan
from situation3d_amd.model import SIG3DQFormer
from situation3d_amd.trainer import build_optimizer, train_step
dev = torch.device("cuda", 0)
gemm_tuning.enable(tune_missing=True)
torch.manual_seed(0)
model = SIG3DQFormer(num_answers=bench.NUM_ANSWERS).to(dev).train()
opt = build_optimizer(model, name="flat_adamw")
batch = bench.synthetic_batch(bench.BATCH, bench.N_POINTS, 7, dev)
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
main = torch.cuda.Stream(dev)
sides = {"normal": torch.cuda.Stream(dev), "low": torch.cuda.Stream(dev, priority=0), "high-main": None}
main_hi = torch.cuda.Stream(dev, priority=-1)


def timed(stream, fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(stream)
    for _ in range(n):
        fn()
    e.record(stream); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


with torch.cuda.stream(main):
    for _ in range(2):
        train_step(model, opt, dict(batch))
    pc = batch["point_clouds"]
    xyz = pc[..., :3].contiguous(); feats = pc[..., 3:].transpose(1, 2).contiguous()
    plan = GeometryPlan(bench.BATCH, bench.N_POINTS, model.encoder.LEVELS, dev).compute(xyz)
    tokens = torch.randn(bench.BATCH, 256, 256, device=dev, requires_grad=True)
    q = batch["q_feat"]
    ones = torch.ones(bench.BATCH, 32, dtype=q["attention_mask"].dtype, device=dev)
    mask = torch.cat([ones, q["attention_mask"]], dim=1)

    def enc():
        with torch.no_grad():
            return model.encoder(xyz, feats, plan)

    def qf():
        out = model.Qformer.bert(query_embeds=model.query_tokens.expand(bench.BATCH, -1, -1), input_ids=q["input_ids"],
                                 attention_mask=mask, encoder_hidden_states=tokens, encoder_attention_mask=None,
                                 return_dict=True)
        h = getattr(out, "query_hidden_state", None)
        h = out.last_hidden_state[:, :32] if h is None else h
        h.float().pow(2).mean().backward()

    for p in model.parameters():
        p.grad = torch.zeros_like(p)
    opt._upload()
    g0 = opt.param_groups[0]; b1, b2 = g0["betas"]
    nchunks = len(opt._static)
    lr0 = torch.zeros(1, device=dev)

    def launch_adam(bound):
        args = [nchunks, _lib.ptr(opt._table), _lib.ptr(opt._step), ctypes.c_float(0.0), _lib.ptr(lr0), ctypes.c_float(b1),
                ctypes.c_float(b2), ctypes.c_float(g0["eps"]), ctypes.c_float(opt.clip_value)]
        if bound:
            _lib.call("sig3d_adamw_table_bounded", *args, int(bound), _lib.stream_ptr(dev))
        else:
            _lib.call("sig3d_adamw_table", *args, _lib.stream_ptr(dev))

    print("chunks %d" % nchunks)
    works = {"encoder forward": enc, "Q-Former forward+backward": qf}
    alone = {}
    for name, fn in works.items():
        fn(); fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with gemm_tuning.no_tuning(), torch.cuda.graph(g, stream=main):
            fn()
        alone[name] = timed(main, g.replay)
        print("%-28s alone %.3f ms" % (name, alone[name]))
    for bound in (0, 2048, 1024, 512, 256):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=main):
            launch_adam(bound)
        t_ad = timed(main, g.replay)
        print("AdamW grid %-5s alone %.3f ms" % (bound or "full", t_ad))
        for name, fn in works.items():
            for sname in ("normal", "low"):
                side = sides[sname]
                gb = torch.cuda.CUDAGraph()
                with gemm_tuning.no_tuning(), torch.cuda.graph(gb, stream=main):
                    side.wait_stream(main)
                    with torch.cuda.stream(side):
                        launch_adam(bound)
                    fn()
                    main.wait_stream(side)
                t = timed(main, gb.replay)
                print("   + %-28s side=%-6s forked %.3f ms (sum %.3f, hidden %.3f)" %
                      (name, sname, t, alone[name] + t_ad, alone[name] + t_ad - t))
