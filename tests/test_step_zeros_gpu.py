"""The step's one zero-filled scratch region (situation3d_amd/scratch.py) and the entry points that take pre-zeroed
outputs or read a stored weight transposed instead of a copy: sig3d_attention_bwd_z, sig3d_pos_mlp_bwd_z,
sig3d_query_group_fused_grad_pm_z, sig3d_mlp_layer_dx, sig3d_mlp_layer0_scatter_dx_w (include/sig3d_hip.h) -- each
against the entry point it shadows, bit for bit where the sums have one order, else within f32 rounding."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def test_region_is_sized_from_the_previous_step_and_falls_back_when_it_does_not_fit():
    from situation3d_amd.scratch import StepZeros
    z = StepZeros()
    outside = z.zeros((3, 5), torch.float32, DEV)              # no step open: a plain tensor
    assert outside.shape == (3, 5) and float(outside.abs().sum()) == 0 and z.hits == 0
    z.begin_step(DEV)
    a = z.zeros((1000,), torch.float64, DEV)                   # first step: demand is recorded, plain tensors
    b = z.zeros((7, 3), torch.int32, DEV)
    z.end_step()
    assert z.hits == 0 and z.misses == 2 and not z.owns(a)
    a.fill_(1.0)
    z.begin_step(DEV)
    a2 = z.zeros((1000,), torch.float64, DEV)
    b2 = z.zeros((7, 3), torch.int32, DEV)
    assert z.hits == 2 and z.owns(a2) and z.owns(b2)
    assert a2.data_ptr() % 256 == 0 and b2.data_ptr() % 256 == 0 and b2.data_ptr() >= a2.data_ptr() + 8000
    assert float(a2.abs().sum()) == 0 and int(b2.abs().sum()) == 0
    c2 = z.zeros((1 << 20,), torch.float32, DEV)               # more than last step asked for: falls back, still zero
    assert not z.owns(c2) and float(c2.abs().sum()) == 0
    a2.fill_(3.0)
    b2.fill_(5)
    z.end_step()
    z.begin_step(DEV)                                          # the next step sees zeros again and room for all three
    a3 = z.zeros((1000,), torch.float64, DEV)
    b3 = z.zeros((7, 3), torch.int32, DEV)
    c3 = z.zeros((1 << 20,), torch.float32, DEV)
    assert z.owns(a3) and z.owns(b3) and z.owns(c3)
    assert float(a3.abs().sum()) == 0 and int(b3.abs().sum()) == 0 and float(c3.abs().sum()) == 0
    z.end_step()
    assert not z.owns(a3)                                      # closed: nothing is promised any more


def test_only_the_stream_that_opened_the_step_is_served_from_the_region():
    """A request with another stream current (a geometry chain beside the step) gets its own torch.zeros: the region
    is ordered with the step's stream only and is zeroed again at the next begin_step."""
    from situation3d_amd.scratch import StepZeros
    z = StepZeros()
    for _ in range(2):
        z.begin_step(DEV)
        z.zeros((1024,), torch.float32, DEV)
        z.end_step()
    z.begin_step(DEV)
    mine = z.zeros((256,), torch.float32, DEV)
    side = torch.cuda.Stream(DEV)
    with torch.cuda.stream(side):
        other = z.zeros((256,), torch.float32, DEV)
    again = z.zeros((256,), torch.float32, DEV)
    assert z.owns(mine) and z.owns(again) and not z.owns(other)
    assert float(other.abs().sum()) == 0
    z.end_step()


def test_a_captured_step_keeps_its_region_when_a_later_step_outgrows_it():
    from situation3d_amd.scratch import StepZeros
    z = StepZeros()
    s = torch.cuda.Stream(DEV)
    with torch.cuda.stream(s):
        for _ in range(2):
            z.begin_step(DEV)
            z.zeros((4096,), torch.float32, DEV)
            z.end_step()
        g = torch.cuda.CUDAGraph()
        out = torch.empty(4096, device=DEV)
        with torch.cuda.graph(g, stream=s):
            z.begin_step(DEV)
            t = z.zeros((4096,), torch.float32, DEV)
            assert z.owns(t)
            t.add_(2.0)
            out.copy_(t)
            z.end_step()
        first = t.data_ptr()
        z.begin_step(DEV)                                      # an eager step that needs far more
        z.zeros((1 << 22,), torch.float32, DEV)
        z.end_step()
        z.begin_step(DEV)
        big = z.zeros((1 << 22,), torch.float32, DEV)
        assert z.owns(big)
        big.fill_(7.0)
        z.end_step()
        g.replay()                                             # the graph's slice is still its own memory
        g.replay()
    torch.cuda.synchronize()
    assert float(out.min()) == 2.0 and float(out.max()) == 2.0
    assert not (big.data_ptr() <= first < big.data_ptr() + big.numel() * 4)
    assert float(big.min()) == 7.0


@pytest.mark.parametrize("cin,cout,e,compact", [(64, 64, 4096, False), (64, 128, 1000, False), (128, 128, 2048, True),
                                                (131, 128, 640, True), (128, 256, 516, False), (6, 64, 333, False),
                                                (259, 128, 96, False)])
def test_layer_input_gradient_reads_the_stored_weight(cin, cout, e, compact):
    """dA = W^T dY with W as the forward layer stores it == the forward kernel run on a W^T copy (what rounds 1-3
    did, bit for bit: same tiles, same order of sums) == einsum in float64."""
    from situation3d_amd import _lib as L
    b = 3
    g = torch.Generator().manual_seed(cin * 1000 + cout + e)
    dY = torch.randn(b, cout, e, generator=g).to(DEV)
    w = (torch.randn(cout, cin, generator=g) / cout ** 0.5).to(DEV)
    n_act = torch.tensor([e, e // 2 + 3, 5], dtype=torch.int32, device=DEV) if compact else None
    dA = torch.full((b, cin, e), float("nan"), device=DEV)
    ref_k = torch.full((b, cin, e), float("nan"), device=DEV)
    st = L.stream_ptr(DEV)
    with torch.cuda.device(DEV):
        L.call("sig3d_mlp_layer_dx", b, cin, cout, e, L.ptr(dY), L.ptr(w), L.ptr(dA), L.ptr(n_act), st)
        wt = w.t().contiguous()
        if compact:
            L.call("sig3d_mlp_layer_fwd_compact", b, cout, cin, e, L.ptr(dY), L.ptr(wt), L.ptr(None), L.ptr(None),
                   L.ptr(ref_k), L.ptr(None), L.ptr(None), 0, L.ptr(n_act), L.ptr(None), st)
        else:
            L.call("sig3d_mlp_layer_fwd", b, cout, cin, e, L.ptr(dY), L.ptr(wt), L.ptr(None), L.ptr(None), L.ptr(ref_k),
                   L.ptr(None), L.ptr(None), 0, st)
    ref = torch.einsum("oc,boe->bce", w.double(), dY.double())
    for i in range(b):
        live = int(n_act[i]) if compact else e
        assert torch.equal(dA[i, :, :live], ref_k[i, :, :live])
        err = float((dA[i, :, :live].double() - ref[i, :, :live]).abs().max())
        assert err < 2e-5 * max(1.0, float(ref.abs().max())), err


def test_attention_backward_into_a_zeroed_dq():
    """Cross-attention shape of the step (32 queries, 256 keys: the keys are split over workgroups)."""
    from situation3d_amd import _lib as L
    b, h, nq, nk, d = 8, 12, 32, 256, 64
    hd = h * d
    g = torch.Generator().manual_seed(5)
    q, k, v, go = (torch.randn(b, n, hd, generator=g).to(DEV) for n in (nq, nk, nk, nq))
    out = torch.empty(b, nq, hd, device=DEV)
    lse = torch.empty(b, h, nq, device=DEV)
    st = L.stream_ptr(DEV)
    res = []
    with torch.cuda.device(DEV):
        L.call("sig3d_attention_fwd", b, h, nq, nk, d, nq, nk, 0, 0, 0, 0, hd, hd, hd, ctypes.c_float(0.125), L.ptr(q),
               L.ptr(k), L.ptr(v), L.ptr(None), L.ptr(out), L.ptr(lse), ctypes.c_float(0.0), ctypes.c_uint(0), L.ptr(None),
               0, L.ptr(None), st)
        for entry, fill in (("sig3d_attention_bwd", float("nan")), ("sig3d_attention_bwd_z", 0.0)):
            dq = torch.full((b, nq, hd), fill, device=DEV)
            dk = torch.full((b, nk, hd), float("nan"), device=DEV)
            dv = torch.full((b, nk, hd), float("nan"), device=DEV)
            L.call(entry, b, h, nq, nk, d, nq, nk, 0, 0, 0, 0, hd, hd, hd, ctypes.c_float(0.125), L.ptr(q), L.ptr(k),
                   L.ptr(v), L.ptr(None), L.ptr(out), L.ptr(lse), L.ptr(go), L.ptr(dq), L.ptr(dk), L.ptr(dv),
                   ctypes.c_float(0.0), ctypes.c_uint(0), L.ptr(None), st)
            res.append((dq, dk, dv))
    assert torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])
    # dq: float atomics over the key splits, order not fixed
    torch.testing.assert_close(res[0][0], res[1][0], rtol=1e-4, atol=1e-5 * float(res[0][0].abs().max()))
    assert float(res[1][0].abs().max()) > 0


def test_positional_mlp_backward_into_zeroed_gradients():
    from situation3d_amd import _lib as L
    rows, cin, hid, cout = 2048, 3, 128, 256
    g = torch.Generator().manual_seed(9)
    x, w2 = torch.randn(rows, cin, generator=g).to(DEV), torch.randn(cout, hid, generator=g).to(DEV)
    pre, dy = torch.randn(rows, hid, generator=g).to(DEV), torch.randn(rows, cout, generator=g).to(DEV)
    n = hid * cin + hid + cout * hid + cout
    st = L.stream_ptr(DEV)
    outs = []
    with torch.cuda.device(DEV):
        for entry, fill in (("sig3d_pos_mlp_bwd", float("nan")), ("sig3d_pos_mlp_bwd_z", 0.0)):
            dpre = torch.empty(rows, hid, device=DEV)
            grads = torch.full((n,), fill, device=DEV)
            L.call(entry, rows, cin, hid, cout, L.ptr(x), L.ptr(w2), L.ptr(pre), L.ptr(dy), L.ptr(dpre), L.ptr(grads), st)
            outs.append((dpre, grads))
    assert torch.equal(outs[0][0], outs[1][0])
    torch.testing.assert_close(outs[0][1], outs[1][1], rtol=1e-4, atol=1e-5 * float(outs[0][1].abs().max()))


def test_grouping_gradient_into_a_zeroed_point_major_buffer():
    from situation3d_amd import _lib as L
    b, n, m, ns, c = 2, 700, 128, 16, 256
    g = torch.Generator().manual_seed(2)
    idx = torch.randint(0, n, (b, m, ns), generator=g, dtype=torch.int32).to(DEV)
    go = torch.randn(b, c + 3, m, ns, generator=g).to(DEV)
    st = L.stream_ptr(DEV)
    outs = []
    with torch.cuda.device(DEV):
        for entry, fill in (("sig3d_query_group_fused_grad_pm", float("nan")), ("sig3d_query_group_fused_grad_pm_z", 0.0)):
            gp = torch.full((b, n, c), fill, device=DEV)
            L.call(entry, b, n, m, c, c, ns, c + 3, 3, L.ptr(go), L.ptr(idx), L.ptr(gp), st)
            outs.append(gp)
    torch.testing.assert_close(outs[0], outs[1], rtol=1e-4, atol=1e-5 * float(outs[0].abs().max()))
    ref = torch.zeros(b, n, c, dtype=torch.float64, device=DEV)
    ref.scatter_add_(1, idx.view(b, -1, 1).long().expand(b, m * ns, c), go[:, 3:].double().permute(0, 2, 3, 1).reshape(b, m * ns, c))
    torch.testing.assert_close(outs[1].double(), ref, rtol=1e-4, atol=1e-5 * float(ref.abs().max()))


def test_training_step_uses_the_region_and_matches_the_step_without_it(monkeypatch):
    """Two steps of the small model with and without the region: same losses (the sums that depend on it are float
    atomics either way: tolerance, not bits), and the second step takes every request from the region."""
    import bench
    from situation3d_amd import scratch
    from situation3d_amd.model import SIG3DQFormer
    from situation3d_amd.trainer import build_optimizer, train_step
    losses = {}
    for on in (True, False):
        monkeypatch.setattr(scratch, "ENABLED", on)
        scratch.STEP_ZEROS.hits = scratch.STEP_ZEROS.misses = 0
        torch.manual_seed(0)
        model = SIG3DQFormer(num_answers=bench.NUM_ANSWERS, qformer_overrides=dict(num_hidden_layers=2, hidden_dropout_prob=0.0,
                                                                                   attention_probs_dropout_prob=0.0)).to(DEV).train()
        model.answer_cls[2].p = 0.0        # (the heads' dropout stream advances from run to run)
        opt = build_optimizer(model, name="flat_adamw")
        batch = bench.synthetic_batch(2, 4096, 7, DEV)
        ls = []
        for i in range(3):
            if i == 2:
                scratch.STEP_ZEROS.hits = scratch.STEP_ZEROS.misses = 0
            ls.append(float(train_step(model, opt, dict(batch)).detach()))
        losses[on] = ls
        if on:
            assert scratch.STEP_ZEROS.hits >= 8 and scratch.STEP_ZEROS.misses == 0
        else:
            assert scratch.STEP_ZEROS.hits == 0
    for a, b2 in zip(losses[True], losses[False]):
        assert abs(a - b2) <= 2e-4 * abs(b2), (losses)
