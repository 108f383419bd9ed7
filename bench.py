"""bench.py -- headline metric of BASELINE.json on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one training pass of the hot path over one synthetic batch:
point encoder (SA1-4 on 40 000 points / scene) -> situational pose re-encode -> Q-Former fusion
(32 queries + 20 question tokens) -> losses -> backward -> value clip -> AdamW.  B = 8 scenes per
GPU (BASELINE config "SQA3D train step fwd+bwd+Adam, 40k pts, B=8"), weak scaling over ranks
(one process per GPU, bucketed RCCL all-reduce overlapped with backward).  Inputs are generated
before the timed region and are resident in HBM.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from situation3d_amd import _lib, gemm_tuning  # noqa: E402
from situation3d_amd.ddp import GradBucketReducer, init_distributed  # noqa: E402
from situation3d_amd.graph_step import GraphedTrainStep  # noqa: E402
from situation3d_amd.model import SIG3DQFormer  # noqa: E402
from situation3d_amd.trainer import build_optimizer, get_loss, train_step  # noqa: E402

N_POINTS, BATCH, N_QUERY, N_TEXT, NUM_ANSWERS = 40000, 8, 32, 20, 706
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
SA_LEVELS = [(40000, 2048, 64, 3), (2048, 1024, 32, 128), (1024, 512, 16, 256), (512, 256, 16, 256)]


def synthetic_batch(batch, n_points, seed, device):
    """xyz ~ U([0,8]x[0,8]x[0,3]) m, colours U(0,1), pose = U(room) +
    z-rotation quaternion, 20 question token ids U(1000,30000), soft multi-hot answers."""
    g = torch.Generator().manual_seed(seed)
    xyz = torch.rand(batch, n_points, 3, generator=g) * torch.tensor([8.0, 8.0, 3.0])
    rgb = torch.rand(batch, n_points, 3, generator=g)
    t = torch.rand(batch, 3, generator=g) * torch.tensor([8.0, 8.0, 3.0])
    ang = (torch.rand(batch, generator=g) * 2 - 1) * 3.14159265
    quat = torch.stack([torch.zeros(batch), torch.zeros(batch), torch.sin(ang / 2), torch.cos(ang / 2)], 1)
    ids = torch.randint(1000, 30000, (batch, N_TEXT), generator=g)
    ans = torch.zeros(batch, NUM_ANSWERS)
    ans[torch.arange(batch), torch.randint(0, NUM_ANSWERS, (batch,), generator=g)] = 1.0
    d = {
        "point_clouds": torch.cat([xyz, rgb], -1),
        "auxiliary_task": torch.cat([t, quat], 1),
        "q_feat": {"input_ids": ids, "attention_mask": torch.ones(batch, N_TEXT, dtype=torch.long)},
        "answer_cat_scores": ans,
    }
    return to_device(d, device)


def to_device(d, device):
    out = {}
    for k, v in d.items():
        out[k] = to_device(v, device) if isinstance(v, dict) else v.to(device).contiguous()
    return out


CPU_SCENES = 4   # scenes in the bounded cpu_baseline sample


def group_algorithmic_bytes(b, n, m, ns, c):
    """group_points(C) = B*(4CN + 4*M*ns + 4C*M*ns); the fused kernel moves the xyz
    group (C=3) and the feature group (C=c) in one launch and reads idx once."""
    return b * (4 * 3 * n + 4 * c * n + 4 * m * ns + 4 * (3 + c) * m * ns)


def oracle_forward(cpu_model, batch):
    """The composed path of `batch` on the HOST through the oracle: the nine native ops from
    oracle/pointnet2_oracle.c (bound as `pointnet2._ext` under this build's module stack), SharedMLP on
    torch CPU, the Q-Former through oracle/qformer_ref.py (eval-mode dropout) -> data_dict with the model's
    output keys and the loss.  Checker code only -- never on the product path."""
    from oracle import pointnet2_ref, qformer_ref
    from situation3d_amd.pointnet2 import pointnet2_utils
    saved_ext = pointnet2_utils._ext
    pointnet2_utils._ext = pointnet2_ref  # CPU restatement of the nine ops
    try:
        n = batch["point_clouds"].shape[0]
        pc = batch["point_clouds"]
        xyz = pc[..., :3].contiguous()
        feats = pc[..., 3:].transpose(1, 2).contiguous()
        tok_xyz, tok_feat = cpu_model.encoder(xyz, feats)
        tok_feat = tok_feat.transpose(1, 2)
        pose = batch["auxiliary_task"]
        M = pointnet2_ref.pose_to_matrix(pose)
        sit = torch.einsum("bcr,bnc->bnr", M[:, :3, :3], tok_xyz - pose[:, None, :3])
        tokens = tok_feat + cpu_model.pos_embed(sit if getattr(cpu_model, "pos_embed_dim", 3) == 3
                                                else tok_xyz[..., :2])
        sd = dict(cpu_model.Qformer.bert.state_dict())
        sd.update({k: v for k, v in cpu_model.Qformer.bert.named_parameters()})
        c = cpu_model.Qformer.config
        cfg = dict(num_hidden_layers=c.num_hidden_layers, num_attention_heads=c.num_attention_heads,
                   layer_norm_eps=c.layer_norm_eps, add_cross_attention=True,
                   cross_attention_freq=c.cross_attention_freq)
        q = batch["q_feat"]
        nq = cpu_model.query_tokens.shape[1]
        att = torch.cat([torch.ones(n, nq, dtype=torch.long), q["attention_mask"]], 1)
        hidden = qformer_ref.bert_model(sd, cfg, query_embeds=cpu_model.query_tokens.expand(n, -1, -1),
                                        input_ids=q["input_ids"], attention_mask=att,
                                        encoder_hidden_states=tokens)
        fused = hidden[:, :nq]
        pooled = fused.mean(1)
        dd = dict(batch)
        dd["scene_positions"], dd["att_feat_pre"], dd["att_feat_ori"] = tok_xyz, tok_feat, fused
        dd["aux_scores"] = cpu_model.aux_reg(pooled)
        dd["answer_scores"] = cpu_model.answer_cls(pooled)
        get_loss(dd)
    finally:
        pointnet2_utils._ext = saved_ext
    return dd


def _without_dropout(model):
    """The oracle's Q-Former is eval-mode (dropout = identity) while BatchNorm keeps batch statistics:
    the comparable product run is train mode with every dropout probability at zero."""
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    return model


def cpu_baseline(model, seed):
    """The same step on the host cores through the ORACLE (kind "port"): CPU_SCENES scenes of the
    B=8 workload as one batch, forward + backward (~10 s) -- and, for free, an end-to-end parity check at
    full size: the same scenes with the same (trained) weights through the HIP path give `loss_gpu`."""
    from oracle import pointnet2_ref
    import copy
    cpu_model = _without_dropout(copy.deepcopy(model).cpu().train())
    batch = synthetic_batch(CPU_SCENES, N_POINTS, seed, "cpu")
    t0 = time.perf_counter()
    dd = oracle_forward(cpu_model, batch)
    dd["loss"].backward()
    dt = time.perf_counter() - t0
    dev = next(model.parameters()).device
    gpu_model = _without_dropout(copy.deepcopy(model).train())
    with torch.no_grad():
        out = gpu_model(to_device(batch, dev))
        loss_gpu, _ = get_loss(out)
    loss_cpu, loss_gpu = float(dd["loss"].detach()), float(loss_gpu)
    scores_diff = float((out["answer_scores"].cpu() - dd["answer_scores"].detach()).abs().max())
    threads = max(pointnet2_ref.num_threads(), torch.get_num_threads())
    return {"value": round(CPU_SCENES / dt, 4), "unit": "samples/s", "cores": threads,
            "kind": "port",
            "sample": "%d scenes (B=%d of the B=8 step), 40k pts, fwd+bwd, oracle C ops (OpenMP, %d threads) + "
                      "torch CPU fp32 MLP/Q-Former (%d threads), %.1f s of wall time"
                      % (CPU_SCENES, CPU_SCENES, pointnet2_ref.num_threads(), torch.get_num_threads(), dt),
            # full-size end-to-end parity of the timed model (trained weights, dropout off, same scenes)
            "loss_cpu": round(loss_cpu, 5), "loss_gpu": round(loss_gpu, 5),
            "loss_rel_diff": float("%.3g" % (abs(loss_cpu - loss_gpu) / max(abs(loss_cpu), 1e-12))),
            "answer_scores_max_abs_diff": float("%.3g" % scores_diff)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="issue every launch eagerly")
    ap.add_argument("--torch-adamw", action="store_true",
                    help="torch.optim.AdamW + clip_grad_value_ instead of the fused flat optimizer")
    ap.add_argument("--force-reducer", action="store_true",
                    help="use the data-parallel code path (flat gradient buckets, two graphs) at N=1")
    ap.add_argument("--no-gemm-tuning", action="store_true",
                    help="library GEMMs with the default heuristic instead of the tuned solutions")
    ap.add_argument("--no-prefetch", action="store_true",
                    help="compute FPS/ball-query geometry inline instead of one batch ahead")
    args = ap.parse_args()

    rank, local, world = init_distributed()
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    assert torch.cuda.is_available(), "bench.py needs an MI355X; the HIP path has no CPU fallback"
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    if not args.no_gemm_tuning:
        gemm_tuning.enable(tune_missing=True)  # committed winners; unseen shapes tuned in warm-up
    torch.manual_seed(1234)  # identical initial weights on every rank
    model = SIG3DQFormer(num_answers=NUM_ANSWERS).to(device).train()
    # clip_grad_value_(1.0) + AdamW (lr 2e-5, wd 0.05: scripts/train.sh:7) + zero_grad fused over
    # flat storage; under data parallelism the same flat gradient buffers are all-reduced in place
    optimizer = build_optimizer(model, name="adamw" if args.torch_adamw else "flat_adamw")
    reducer = None
    if world > 1 or args.force_reducer:
        reducer = (GradBucketReducer(model.parameters()) if args.torch_adamw
                   else GradBucketReducer.from_flat(optimizer.flat_grad_buffers()))

    n_batches = min(4, args.steps + args.warmup)
    batches = [synthetic_batch(BATCH, N_POINTS, 1234 + 1000 * rank + i, device) for i in range(n_batches)]

    # every forward/backward of this process runs on ONE non-default stream (see graph_step.py)
    work = torch.cuda.Stream(device)
    KSTEPS = 3
    with torch.cuda.stream(work):
        use_graph = not args.no_graph
        if use_graph:
            graphed = GraphedTrainStep(model, optimizer, batches[0],
                                       prefetch_geometry=not args.no_prefetch, reducer=reducer)

            def step(i):
                return graphed(batches[i % n_batches], batches[(i + 1) % n_batches])
        else:
            def step(i):
                return train_step(model, optimizer, dict(batches[i % n_batches]), reducer=reducer)

        for i in range(args.warmup):
            step(i)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            loss = step(args.warmup + i)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        final_loss = float(loss.item())

        # Per-launch durations of the hand-written kernels: HIP events on the launch stream around
        # every call.  Under hipGraph replay there are no per-kernel events, so the SAME launches are
        # issued once more eagerly right after the timed region and bracketed there.
        _lib.enable_timing(["sig3d_query_group_fused", "sig3d_query_group_fused_pm", "sig3d_query_group_compact",
                            "sig3d_transpose_cn", "sig3d_adamw_table", "sig3d_adamw_flat", "sig3d_ball_query", "sig3d_ball_query_grid",
                            "sig3d_furthest_point_sampling"])
        if reducer is not None:
            reducer.hooks_enabled = True
        for i in range(KSTEPS):
            train_step(model, optimizer, dict(batches[i % n_batches]), reducer=reducer)
        torch.cuda.synchronize()
        recs = _lib.timing_records()
        _lib.enable_timing(None)

    t = torch.tensor([dt], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    if rank == 0:
        def kernel_ms(name):
            r = recs[name]
            return [s.elapsed_time(e) for s, e, _ in r]

        # the grouping launches of a step: the narrow SA1 level through query_group_fused_kernel, the wide
        # levels through its point-major twin (plus the small transposes that feed it, reported apart)
        # Levels whose neighbour lists are mostly padding run in compact mode (distinct neighbours only:
        # DESIGN.md 5d) and never form the dense grouped tensor; the roofline is taken over the launches
        # that do, with the algorithmic bytes of exactly those launches (shapes from the recorded arguments).
        grp, grp_bytes = [], 0
        for name, pm in (("sig3d_query_group_fused", False), ("sig3d_query_group_fused_pm", True)):
            for s_ev, e_ev, ints in recs[name]:
                bb, nn, mm, cc = ints[0], ints[1], ints[2], ints[3]
                ns_ = ints[5] if pm else ints[4]
                grp.append(s_ev.elapsed_time(e_ev))
                grp_bytes += group_algorithmic_bytes(bb, nn, mm, ns_, cc)
        cgrp = kernel_ms("sig3d_query_group_compact")
        # the largest HBM-bound kernel of the step by time is the flat AdamW update: per parameter it reads
        # p, g, m, v and writes p, m, v (28 B: clip + update in one pass; the gradients are dropped, not zeroed)
        adam = kernel_ms("sig3d_adamw_table") + kernel_ms("sig3d_adamw_flat")
        n_params = sum(p.numel() for p in model.parameters() if p.requires_grad)
        adam_gbs = 28.0 * n_params * KSTEPS / (sum(adam) * 1e-3) / 1e9 if adam else 0.0
        tr = kernel_ms("sig3d_transpose_cn")
        achieved = grp_bytes / (sum(grp) * 1e-3) / 1e9 if grp else 0.0
        bq = kernel_ms("sig3d_ball_query") + kernel_ms("sig3d_ball_query_grid")
        fps = kernel_ms("sig3d_furthest_point_sampling")
        # HBM traffic of the roofline kernel cannot be read from inside this process: it comes from
        # the committed rocprofv3 --pmc passes (FETCH_SIZE doubled per MI355X_MICROARCH.md, + WRITE_SIZE)
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_query_group_fused.json")
        if os.path.exists(pmc):
            traffic = round(json.load(open(pmc))["traffic_bytes_per_launch"])
        compact_info = {}
        plan = getattr(graphed, "plan_cur", None) if use_graph else None
        if plan is not None:
            for li, cl in enumerate(plan.compact):
                if cl is not None:
                    bsz, mpt, nsm = cl.shape
                    compact_info["SA%d" % (li + 1)] = round(float(cl.n_act.float().mean().item()) / (mpt * nsm), 4)
        out = {
            "metric": "QA samples/sec fwd+bwd (SQA3D, 40k pts, B=8)",
            "value": round(world * BATCH * args.steps / dt, 3),
            "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "SQA3D train step fwd+bwd+AdamW, 40k pts/scene, B=8/GPU, "
                                   "SA1-4 -> 256 tokens -> situational re-encode -> Q-Former "
                                   "(32 queries + 20 question tokens, 12 layers)",
                       "global_batch": world * BATCH, "points_per_scene": N_POINTS,
                       "parallelism": "dp%d" % world},
            "roofline": {"bound": "hbm", "kernel": "query_group_fused_kernel + query_group_fused_pm_kernel",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "algorithmic_bytes_per_launch": round(grp_bytes / max(len(grp), 1)),
                         "launches": len(grp), "avg_launch_us": round(sum(grp) / max(len(grp), 1) * 1e3, 2)},
            "roofline_adamw": {"bound": "hbm", "kernel": "adamw_table_kernel", "achieved": round(adam_gbs, 1),
                               "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(adam_gbs / HBM_PEAK_GBS, 4),
                               "algorithmic_bytes_per_launch": 28 * n_params,
                               "avg_launch_us": round(sum(adam) / max(len(adam), 1) * 1e3, 1)},
            "kernels_ms_per_step": {"query_group_fused": round(sum(grp) / KSTEPS, 4),
                                    "query_group_compact": round(sum(cgrp) / KSTEPS, 4),
                                    "point_major_transposes": round(sum(tr) / KSTEPS, 4),
                                    "ball_query": round(sum(bq) / KSTEPS, 4),
                                    "furthest_point_sampling": round(sum(fps) / KSTEPS, 4)},
            # set-abstraction levels that ran over the distinct neighbours only, with the fraction of their
            # (centre, sample) positions that are distinct on this batch
            "compact_levels": compact_info,
            "launch_mode": "hipGraph replay" if use_graph else "eager",
            "library_gemms": "default heuristic" if args.no_gemm_tuning else "tuned (TunableOp)",
            "final_loss": round(final_loss, 5),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(model, 1234)
        line = json.dumps(out)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # RCCL prints its version banner through C stdio (flushed at exit when stdout is a file): push it
        # out first so that the JSON line is the LAST line of rank 0's stdout
        import ctypes
        ctypes.CDLL(None).fflush(None)
        print(line, flush=True)


if __name__ == "__main__":
    main()
