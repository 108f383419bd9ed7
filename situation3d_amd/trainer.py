"""Training-step semantics of the reference's two harnesses, for the composed hot path.

SQA3D side (lib/solver.py + lib/loss_helper.py + situation3d/train/train.py):
  * loss = QA_W * answer_loss + SITUATION_W * (POS_W * pos + ROT_W * rot), then `loss *= 10`
    (loss_helper.py:195-227, 286-300; weights lib/config.py:72-79);
  * answer loss = BCE-with-logits, reduction 'sum' / batch when soft multi-hot targets
    (`answer_cat_scores`) are present, cross-entropy on `answer_cat` otherwise (:222-227);
  * zero_grad -> backward -> clip_grad_VALUE_(1.0) -> optimizer.step()  (solver.py:618-627);
  * AdamW with no weight decay on names containing "bias" / "LayerNorm.weight"
    (train.py:186-238; scripts/train.sh:7: lr 2e-5, wd 0.05).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .scratch import STEP_ZEROS

LOSS_W = dict(SITUATION_W=0.1, QA_W=0.1, SITUATION_POS_W=1.0, SITUATION_ROT_W=1.0)


def compute_answer_classification_loss(data_dict):
    if "answer_cat_scores" in data_dict:
        return F.binary_cross_entropy_with_logits(
            data_dict["answer_scores"], data_dict["answer_cat_scores"],
            reduction="sum") / data_dict["answer_scores"].shape[0]
    return F.cross_entropy(data_dict["answer_scores"], data_dict["answer_cat"])


def compute_aux_situation_loss(data_dict, tag="__l2__quat__"):
    fn = F.mse_loss if "__l2__" in tag else F.l1_loss
    pos = fn(data_dict["aux_scores"][:, :3], data_dict["auxiliary_task"][:, :3], reduction="mean")
    rot = fn(data_dict["aux_scores"][:, 3:], data_dict["auxiliary_task"][:, 3:], reduction="mean")
    return LOSS_W["SITUATION_POS_W"] * pos + LOSS_W["SITUATION_ROT_W"] * rot, pos, rot


class _FusedSQALossFn(torch.autograd.Function):
    """loss_helper.py:195-227, 286-300 as one HIP launch (csrc/sqa_loss.hip); backward = one scaling launch."""

    @staticmethod
    def forward(ctx, answer_scores, aux_scores, answer_targets, aux_targets, l1):
        import ctypes
        from . import _lib
        a, x = answer_scores.contiguous(), aux_scores.contiguous()
        dev = a.device
        losses = torch.empty(5, dtype=torch.float32, device=dev)
        d_a, d_x = torch.empty_like(a), torch.empty_like(x)
        f = ctypes.c_float
        with torch.cuda.device(dev):
            _lib.call("sig3d_sqa_loss", a.shape[0], a.shape[1], x.shape[1], int(l1), _lib.ptr(a),
                      _lib.ptr(answer_targets.contiguous()), _lib.ptr(x), _lib.ptr(aux_targets.contiguous()),
                      f(LOSS_W["QA_W"]), f(LOSS_W["SITUATION_W"]), f(LOSS_W["SITUATION_POS_W"]),
                      f(LOSS_W["SITUATION_ROT_W"]), f(10.0), _lib.ptr(losses), _lib.ptr(d_a), _lib.ptr(d_x),
                      _lib.stream_ptr(dev))
        ctx.save_for_backward(d_a, d_x)
        ctx.mark_non_differentiable(losses)
        return losses[0], losses

    @staticmethod
    def backward(ctx, g, _):
        from . import _lib
        d_a, d_x = ctx.saved_tensors
        g_a, g_x = torch.empty_like(d_a), torch.empty_like(d_x)
        with torch.cuda.device(d_a.device):
            _lib.call("sig3d_sqa_loss_scale", d_a.numel(), d_x.numel(), _lib.ptr(g.contiguous()), _lib.ptr(d_a),
                      _lib.ptr(d_x), _lib.ptr(g_a), _lib.ptr(g_x), _lib.stream_ptr(d_a.device))
        return g_a, g_x, None, None, None


def get_loss(data_dict, situation_loss_tag="__l2__quat__", use_aux_situation=True, use_answer=True):
    a, x = data_dict.get("answer_scores"), data_dict.get("aux_scores")
    if (use_aux_situation and use_answer and a is not None and a.is_cuda and a.dtype == torch.float32
            and "answer_cat_scores" in data_dict and x is not None and x.dim() == 2 and x.shape[1] >= 4
            and ("__l2__" in situation_loss_tag or "__l1__" in situation_loss_tag)):
        # the GPU hot path: every term of the loss and its gradient in one launch
        loss, parts = _FusedSQALossFn.apply(a, x, data_dict["answer_cat_scores"].to(torch.float32),
                                            data_dict["auxiliary_task"].to(torch.float32),
                                            "__l2__" not in situation_loss_tag)
        data_dict["answer_loss"], data_dict["pos_loss"], data_dict["rot_loss"], data_dict["aux_loss"] = \
            parts[1], parts[2], parts[3], parts[4]
        data_dict["loss"] = loss
        return loss, data_dict
    zero = data_dict["answer_scores"].new_zeros(())
    data_dict["answer_loss"] = compute_answer_classification_loss(data_dict) if use_answer else zero
    if use_aux_situation:
        aux, pos, rot = compute_aux_situation_loss(data_dict, situation_loss_tag)
    else:
        aux = pos = rot = zero
    data_dict["aux_loss"], data_dict["pos_loss"], data_dict["rot_loss"] = aux, pos, rot
    loss = LOSS_W["SITUATION_W"] * aux + LOSS_W["QA_W"] * data_dict["answer_loss"]
    loss = loss * 10  # loss_helper.py:300 "amplify"
    data_dict["loss"] = loss
    return loss, data_dict


def storage_layout(model, param_lists, qf_cut=None):
    """The order in which FlatAdamW lays `param_lists` (one list per parameter group) out in its flat buffers: the lists'
    own (named_parameters) order, except that the head of every adjacency group of the Q-Former
    (qformer.parameter_adjacency_groups) pulls the rest of its group in right behind it -- operands the Q-Former stacks
    into one GEMM, and the layer-batched weight-gradient buffers, are then views of the flat storage.  qf_cut = k: two
    kind-major arenas (layers below k, layers from k on), each one contiguous stretch of every list."""
    from .qformer import parameter_adjacency_groups
    groups = parameter_adjacency_groups(model, cut=qf_cut)
    follow = {}
    for grp in groups:
        for p in grp[1:]:
            follow[id(p)] = grp
    heads = {id(g[0]): g for g in groups}

    def ordered(params):
        present = {id(p) for p in params}
        out, done = [], set()
        for p in params:
            if id(p) in done or (id(p) in follow and id(follow[id(p)][0]) in present):
                continue  # emitted together with the head of its group
            out.append(p)
            done.add(id(p))
        final = []
        for p in out:       # heads pull their followers in right behind them
            final.append(p)
            for q in heads.get(id(p), ())[1:]:
                if id(q) in present:
                    final.append(q)
        return final

    return [ordered(ps) for ps in param_lists]


def build_optimizer(model, lr=2e-5, wd=0.05, betas=(0.9, 0.999), eps=1e-8, name="adamw",
                    clip_value=1.0, qf_cut=None):
    """qf_cut (flat_adamw only): store the Q-Former's layers below / from this layer on as two arenas, for a data-parallel
    step whose backward pass is cut there (graph_step.GraphedTrainStep reads the same number from the encoder's
    `storage_cut`); None / 0: one arena."""
    no_decay_filter = ("bias", "LayerNorm.weight")
    decay, no_decay = [], []
    for n, p in model.named_parameters():
        if not p.requires_grad:
            continue
        (no_decay if any(nd in n for nd in no_decay_filter) else decay).append(p)
    if name == "flat_adamw":
        # the groups keep named_parameters() order (what a torch.optim.AdamW / reference checkpoint indexes by:
        # lib/solver.py:657, train.py:262); only the flat buffers follow the adjacency order
        layout = storage_layout(model, [decay, no_decay], qf_cut=qf_cut)
        enc = getattr(getattr(getattr(model, "Qformer", None), "bert", None), "encoder", None)
        if enc is not None:
            enc.storage_cut = int(qf_cut) if qf_cut else None
    groups = [{"params": decay, "weight_decay": wd}, {"params": no_decay, "weight_decay": 0.0}]
    if name == "flat_adamw":
        # clip_grad_value_(1.0) + AdamW + zero_grad as one streaming kernel per group (optim.py)
        from .optim import FlatAdamW
        return FlatAdamW(groups, lr=lr, betas=betas, eps=eps, clip_value=clip_value, storage_order=layout)
    cls = torch.optim.AdamW if name == "adamw" else torch.optim.Adam
    # fused: one multi-tensor HIP kernel per step instead of ~4 launches per parameter;
    # capturable: step counters live on the device so the step can sit inside a hipGraph
    on_gpu = all(p.is_cuda for g in groups for p in g["params"])
    kw = {"fused": True, "capturable": True} if on_gpu else {}
    return cls(groups, lr=lr, betas=betas, eps=eps, **kw)


def train_step(model, optimizer, data_dict, max_grad_value=1.0, reducer=None):
    """One Solver iteration (solver.py:374-402, 618-627).  With a GradBucketReducer the gradient
    all-reduce is launched bucket by bucket from inside backward and joined before clipping."""
    fused = getattr(optimizer, "flat_grad_buffers", None) is not None  # optim.FlatAdamW
    if fused and reducer is not None and getattr(optimizer, "process_group", None) is None:
        optimizer.process_group = reducer.group   # FlatAdamW agrees on live parameters among the ranks that average them
    if fused:
        pass  # gradients were zeroed by the previous step() (and start at zero)
    elif reducer is not None:
        reducer.zero_grad()
    else:
        optimizer.zero_grad(set_to_none=False)
    pc = data_dict.get("point_clouds")
    # every zero-initialised accumulator of the step comes out of one region cleared by one fill (scratch.py)
    STEP_ZEROS.begin_step(pc.device if torch.is_tensor(pc) else "cpu", grads_ok=fused)
    try:
        data_dict = model(data_dict)
        loss, data_dict = get_loss(data_dict)
        loss.backward()
        enc = getattr(getattr(getattr(model, "Qformer", None), "bert", None), "encoder", None)
        if enc is not None and hasattr(enc, "flush_weight_grads"):
            enc.flush_weight_grads()   # deferred, layer-batched weight gradients (qformer._WeightGradArena)
        if reducer is not None:
            if fused:
                optimizer.gather_grads()  # scattered grads -> flat buffers (reducer.from_flat slices)
            reducer.finish()
        if not fused and max_grad_value is not None and max_grad_value > 0:
            nn.utils.clip_grad_value_(model.parameters(), clip_value=max_grad_value)
        optimizer.step()  # FlatAdamW: clip + update + zero_grad in one kernel per group
    finally:
        STEP_ZEROS.end_step()
    return loss
