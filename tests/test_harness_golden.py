"""the training-step semantics and the live situational math against fixtures
the REFERENCE's own functions produced (tests/golden/make_golden_r2.py compiled `get_loss`,
`compute_*_loss` from lib/loss_helper.py:195-302, `Solver._backward` from lib/solver.py:618-627 and the
statements sqa_module.py:328-338 / :274-278 in the build container).  CPU tests here; the same
trajectory through the fused flat optimizer on the GPU is marked gpu."""
import os

import numpy as np
import pytest
import torch

from util import tiny_head

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return {k: torch.from_numpy(np.array(v, copy=True)) for k, v in np.load(os.path.join(GOLD, name)).items()}


def _trajectory(device, optimizer_name):
    from situation3d_amd.trainer import build_optimizer, get_loss, train_step
    g = _load("harness_trajectory.npz")
    model = tiny_head()
    model.load_state_dict({k[len("state."):]: v for k, v in g.items() if k.startswith("state.")}, strict=True)
    model.to(device).train()
    opt = build_optimizer(model, lr=float(g["lr"]), wd=float(g["wd"]), name=optimizer_name)
    losses, parts = [], []
    for i in range(3):
        dd = {"x": g["x"][i].to(device), "auxiliary_task": g["auxiliary_task"][i].to(device)}
        if i < 2:
            dd["answer_cat_scores"] = g["answer_cat_scores"][i].to(device)
        else:
            dd["answer_cat"] = g["answer_cat"][i].to(device)
        loss = train_step(model, opt, dd)
        losses.append(float(loss.detach()))
        parts.append([float(dd[k].detach()) for k in ("answer_loss", "pos_loss", "rot_loss", "aux_loss")])
    # loss x10, the two answer-loss branches, L2 auxiliary loss, weights of lib/config.py:72-79
    torch.testing.assert_close(torch.tensor(losses, dtype=torch.float64), g["losses"], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(torch.tensor(parts, dtype=torch.float64), g["loss_parts"], rtol=1e-4, atol=1e-4)
    # value clip at 1.0 (exercised: 665 clipped elements in the fixture) + AdamW groups
    for k, v in model.state_dict().items():
        torch.testing.assert_close(v.cpu(), g["state_after." + k], rtol=1e-4, atol=1e-5, msg=lambda m: k + ": " + m)
    # remaining branches of get_loss (evaluated, like the fixture, with the weights after the three steps)
    base = {"x": g["x"][0].to(device), "auxiliary_task": g["auxiliary_task"][0].to(device),
            "answer_cat_scores": g["answer_cat_scores"][0].to(device)}
    l1, _ = get_loss(model(dict(base)), situation_loss_tag="__l1__quat__")
    noaux, _ = get_loss(model(dict(base)), use_aux_situation=False)
    torch.testing.assert_close(l1.detach().cpu().double(), g["loss_l1_tag"], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(noaux.detach().cpu().double(), g["loss_no_aux"], rtol=1e-4, atol=1e-4)


def test_three_step_trajectory_matches_reference_harness_cpu():
    assert int(_load("harness_trajectory.npz")["clipped_elements"]) > 0
    _trajectory("cpu", "adamw")


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["adamw", "flat_adamw"])
def test_three_step_trajectory_matches_reference_harness_gpu(name):
    """flat_adamw: clip + AdamW + zero_grad in ONE HIP kernel (csrc/optim.hip)."""
    _trajectory("cuda:0", name)


def test_gaussian_localisation_target_matches_reference():
    """sqa_module.py:328-338 (2-D token positions, sigma 0.16, one agent exactly on a token)."""
    from situation3d_amd.situational import gaussian_localisation_target
    g = _load("situational_live.npz")
    w = gaussian_localisation_target(g["scene_positions"], g["auxiliary_task"][:, :3])
    torch.testing.assert_close(w, g["auxiliary_task_loc_gt"], rtol=1e-5, atol=1e-7)
    # 3-D token positions (this build's SA4 centres): only x, y enter, as in the reference
    p3 = torch.cat([g["scene_positions"], torch.rand(3, 256, 1)], -1)
    torch.testing.assert_close(gaussian_localisation_target(p3, g["auxiliary_task"][:, :3]),
                               g["auxiliary_task_loc_gt"], rtol=1e-5, atol=1e-7)


def test_two_d_pos_embed_matches_reference():
    """SIG3DQFormer(pos_embed_dim=2) builds the reference's Linear(2,128)-GELU-Linear(128,256)
    (sqa_module.py:274-278): same state_dict keys, same values on the reference's input."""
    from situation3d_amd.model import build_pos_embed
    g = _load("situational_live.npz")
    pe = build_pos_embed(2, 256)
    pe.load_state_dict({k[len("pos_embed."):]: v for k, v in g.items() if k.startswith("pos_embed.")}, strict=True)
    torch.testing.assert_close(pe(g["scene_positions"]), g["pos_embed_out"], rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["__l2__quat__", "__l1__quat__"])
def test_fused_loss_kernel_matches_torch_formulation(tag):
    """csrc/sqa_loss.hip (one launch for every term of loss_helper.py:195-227, 286-300 and its gradient) against
    the torch formulation the CPU path keeps, at the bench's sizes: loss terms and d loss / d scores."""
    from situation3d_amd import trainer
    g = torch.Generator().manual_seed(3)
    B, A = 8, 706
    scores = (torch.randn(B, A, generator=g) * 3).cuda().requires_grad_(True)
    aux = torch.randn(B, 7, generator=g).cuda().requires_grad_(True)
    targets = (torch.rand(B, A, generator=g) > 0.99).float().cuda()
    pose = (torch.randn(B, 7, generator=g) * 2).cuda()
    dd = {"answer_scores": scores, "aux_scores": aux, "answer_cat_scores": targets, "auxiliary_task": pose}
    loss, dd = trainer.get_loss(dd, situation_loss_tag=tag)
    (loss * 0.7).backward()
    s2, a2 = scores.detach().cpu().double().requires_grad_(True), aux.detach().cpu().double().requires_grad_(True)
    ref = {"answer_scores": s2, "aux_scores": a2, "answer_cat_scores": targets.cpu().double(), "auxiliary_task": pose.cpu().double()}
    rl, ref = trainer.get_loss(ref, situation_loss_tag=tag)     # host tensors: the torch formulation
    (rl * 0.7).backward()
    for k in ("loss", "answer_loss", "pos_loss", "rot_loss", "aux_loss"):
        torch.testing.assert_close(dd[k].detach().cpu().double(), ref[k].detach(), rtol=1e-5, atol=1e-6, msg=k)
    torch.testing.assert_close(scores.grad.cpu().double(), s2.grad, rtol=1e-5, atol=1e-8)
    torch.testing.assert_close(aux.grad.cpu().double(), a2.grad, rtol=1e-5, atol=1e-8)


@pytest.mark.gpu
def test_gaussian_target_kernel_matches_reference():
    from situation3d_amd.situational import gaussian_localisation_target
    g = _load("situational_live.npz")
    w = gaussian_localisation_target(g["scene_positions"].cuda(), g["auxiliary_task"].cuda())
    torch.testing.assert_close(w.cpu(), g["auxiliary_task_loc_gt"], rtol=1e-5, atol=1e-7)
    p3 = torch.cat([g["scene_positions"], torch.rand(3, 256, 1)], -1).cuda()
    torch.testing.assert_close(gaussian_localisation_target(p3, g["auxiliary_task"].cuda()).cpu(),
                               g["auxiliary_task_loc_gt"], rtol=1e-5, atol=1e-7)
