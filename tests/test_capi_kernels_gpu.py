"""Direct C-ABI checks (ctypes -> libsig3d_hip.so) of the situational transform and the
attention kernels.  Floating point rows: tolerance written next to each assert
(north star: fp32 activations within 1e-4).
"""
import ctypes
import math

import pytest
import torch

from util import poses, scene

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _lib():
    from situation3d_amd import _lib
    return _lib


@pytest.mark.parametrize("n", [1, 7, 256, 40000])
@pytest.mark.parametrize("inverse", [0, 1])
def test_situational_transform_vs_oracle(oracle, n, inverse):
    L = _lib()
    b = 3
    pose = poses(b, seed=n)
    pose[1, 3:] = torch.tensor([0.3, -0.2, 0.5, 0.9])  # non-unit quaternion: x2-y2-z2+w2 form
    pts = scene(b, n, seed=n + 1)
    out = torch.empty(b, n, 3, device=DEV)
    pd, xd = pose.to(DEV), pts.to(DEV)
    L.call("sig3d_situational_transform", b, n, L.ptr(pd), L.ptr(xd), L.ptr(out), inverse,
           L.stream_ptr())
    if not inverse:
        ref = oracle.situational_transform(pose, pts)
        assert torch.equal(out.cpu(), ref)  # same unfused evaluation order: bit-exact
    else:
        M = oracle.pose_to_matrix(pose)
        ref = torch.einsum("bcr,bnc->bnr", M[:, :3, :3].double(), (pts - pose[:, None, :3]).double())
        torch.testing.assert_close(out.cpu().double(), ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("inverse", [0, 1])
def test_situational_transform_grad(oracle, inverse):
    L = _lib()
    b, n = 2, 1000
    pose = poses(b, seed=3)
    pose[:, 3:] += 0.1
    pts = scene(b, n, seed=4)
    go = torch.randn(b, n, 3, generator=torch.Generator().manual_seed(5))

    def torch_fwd(pose, pts):
        t, x, y, z, w = pose[:, :3], pose[:, 3], pose[:, 4], pose[:, 5], pose[:, 6]
        R = torch.stack([
            torch.stack([x * x - y * y - z * z + w * w, 2 * (x * y - z * w), 2 * (x * z + y * w)], -1),
            torch.stack([2 * (x * y + z * w), -x * x + y * y - z * z + w * w, 2 * (y * z - x * w)], -1),
            torch.stack([2 * (x * z - y * w), 2 * (y * z + x * w), -x * x - y * y + z * z + w * w], -1)], 1)
        if inverse:
            return torch.einsum("bcr,bnc->bnr", R, pts - t[:, None])
        return torch.einsum("brc,bnc->bnr", R, pts) + t[:, None]

    p64 = pose.double().requires_grad_(True)
    x64 = pts.double().requires_grad_(True)
    (torch_fwd(p64, x64) * go.double()).sum().backward()
    gp = torch.empty(b, n, 3, device=DEV)
    gpose = torch.empty(b, 7, device=DEV)
    pd, xd, gd = pose.to(DEV), pts.to(DEV), go.to(DEV)
    L.call("sig3d_situational_transform_grad", b, n, L.ptr(pd), L.ptr(xd), L.ptr(gd), L.ptr(gp),
           L.ptr(gpose), inverse, L.stream_ptr())
    torch.testing.assert_close(gp.cpu().double(), x64.grad, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(gpose.cpu().double(), p64.grad, rtol=1e-4, atol=1e-3)


def _attn_ref(q, k, v, mask, scale):
    s = torch.matmul(q.double(), k.double().transpose(-1, -2)) * scale
    if mask is not None:
        s = s + mask.double()[:, None, None, :]
    p = torch.softmax(s, -1)
    ctx = torch.matmul(p, v.double())  # (b,h,nq,d)
    b, h, nq, d = ctx.shape
    return ctx.permute(0, 2, 1, 3).reshape(b, nq, h * d), torch.logsumexp(s, -1)


@pytest.mark.parametrize("b,h,nq,nk,use_mask", [(2, 12, 32, 256, True), (1, 3, 52, 52, True),
                                                (2, 2, 32, 5000, False), (1, 1, 5, 33, True),
                                                (1, 2, 32, 1, False), (1, 2, 100, 77, True),
                                                (2, 2, 256, 256, True), (1, 2, 300, 48, False), (1, 1, 129, 129, True),
                                                (1, 2, 20, 700, True), (3, 1, 32, 1000, True), (1, 1, 7, 97, False)])
@pytest.mark.parametrize("D", [64, 96])
@pytest.mark.parametrize("small", ["1", "0"])
def test_attention_fwd_bwd(b, h, nq, nk, use_mask, D, small, monkeypatch):
    """Head sizes 64 (Q-Former) and 96 (MCAN blocks); more than 128 query rows run the backward in query
    chunks with atomic dK / dV.  small: the small-problem backward (eight waves per (batch, head), everything in LDS:
    <= 64 x 64 and 32 x <= 1024, head size 64) or the generic backward for every shape."""
    if small == "0" and (D != 64 or (nq > 64 and nk > 64)):
        pytest.skip("no small-problem kernel for this shape: same launches as small=1")
    monkeypatch.setenv("SIG3D_ATTN_BWD_SMALL", small)
    L = _lib()
    g = torch.Generator().manual_seed(nq * 131 + nk + D)
    q = torch.randn(b, h, nq, D, generator=g)
    k = torch.randn(b, h, nk, D, generator=g)
    v = torch.randn(b, h, nk, D, generator=g)
    mask = None
    if use_mask:
        keep = (torch.rand(b, nk, generator=g) > 0.2).float()
        keep[:, 0] = 1.0
        mask = (1.0 - keep) * -10000.0  # Qformer.py:731
    scale = 1.0 / math.sqrt(D)
    go = torch.randn(b, nq, h * D, generator=g)

    # C ABI takes token-major (b, n, h*d) operands (what nn.Linear produces)
    tm = lambda t: t.permute(0, 2, 1, 3).reshape(t.shape[0], t.shape[2], -1).contiguous()
    untm = lambda t: t.reshape(t.shape[0], t.shape[1], h, D).permute(0, 2, 1, 3)
    qd, kd, vd, god = tm(q).to(DEV), tm(k).to(DEV), tm(v).to(DEV), go.to(DEV)
    md = mask.to(DEV) if mask is not None else None
    out = torch.empty(b, nq, h * D, device=DEV)
    lse = torch.empty(b, h, nq, device=DEV)
    ld = h * D  # dense token-major operands
    L.call("sig3d_attention_fwd", b, h, nq, nk, D, nq, nk, 0, 0, 0, 0, ld, ld, ld, ctypes.c_float(scale), L.ptr(qd), L.ptr(kd),
           L.ptr(vd), L.ptr(md), L.ptr(out), L.ptr(lse), ctypes.c_float(0.0), ctypes.c_uint(0),
           L.ptr(None), 1, L.ptr(None), L.stream_ptr())

    q64 = q.double().requires_grad_(True)
    k64 = k.double().requires_grad_(True)
    v64 = v.double().requires_grad_(True)
    ref, ref_lse = _attn_ref(q64, k64, v64, mask, scale)
    # tolerance: 1e-4 absolute on O(1) activations (north star), observed ~1e-6
    torch.testing.assert_close(out.cpu().double(), ref.detach(), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(lse.cpu().double(), ref_lse.detach(), rtol=1e-4, atol=1e-4)

    (ref * go.double()).sum().backward()
    dq = torch.empty_like(qd)
    dk = torch.empty_like(kd)
    dv = torch.empty_like(vd)
    L.call("sig3d_attention_bwd", b, h, nq, nk, D, nq, nk, 0, 0, 0, 0, ld, ld, ld, ctypes.c_float(scale), L.ptr(qd), L.ptr(kd),
           L.ptr(vd), L.ptr(md), L.ptr(out), L.ptr(lse), L.ptr(god), L.ptr(dq), L.ptr(dk),
           L.ptr(dv), ctypes.c_float(0.0), ctypes.c_uint(0), L.ptr(None), L.stream_ptr())
    torch.testing.assert_close(untm(dq.cpu()).double(), q64.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(untm(dk.cpu()).double(), k64.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(untm(dv.cpu()).double(), v64.grad, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("b,use_mask", [(1, True), (2, False)])
def test_attention_fwd_bwd_config5_nk80000(b, use_mask):
    """BASELINE config 5: 12 heads x 32 queries x 80 000 keys -- the key-split forward and the register-resident
    one-query-tile backward (`attention_bwd_kernel<64, ONEQT>`), against float64 torch (Qformer.py:185-227).
    Sharpened scores (q x 3) so that the softmax over 80 000 keys is not flat; relative max error, since the
    outputs of an 80 000-key average are O(1e-2)."""
    L = _lib()
    h, nq, nk, D = 12, 32, 80000, 64
    g = torch.Generator().manual_seed(8000 + b)
    q = (torch.randn(b, nq, h * D, generator=g) * 3.0).to(DEV)
    k = torch.randn(b, nk, h * D, generator=g).to(DEV)
    v = torch.randn(b, nk, h * D, generator=g).to(DEV)
    go = torch.randn(b, nq, h * D, generator=g).to(DEV)
    mask = None
    if use_mask:
        keep = (torch.rand(b, nk, generator=g) > 0.2).float()
        keep[:, 0] = 1.0
        mask = ((1.0 - keep) * -10000.0).to(DEV)
    scale = 1.0 / math.sqrt(D)
    ld = h * D
    out = torch.empty(b, nq, ld, device=DEV)
    lse = torch.empty(b, h, nq, device=DEV)
    L.call("sig3d_attention_fwd", b, h, nq, nk, D, nq, nk, 0, 0, 0, 0, ld, ld, ld, ctypes.c_float(scale), L.ptr(q), L.ptr(k),
           L.ptr(v), L.ptr(mask), L.ptr(out), L.ptr(lse), ctypes.c_float(0.0), ctypes.c_uint(0),
           L.ptr(None), 1, L.ptr(None), L.stream_ptr())
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    L.call("sig3d_attention_bwd", b, h, nq, nk, D, nq, nk, 0, 0, 0, 0, ld, ld, ld, ctypes.c_float(scale), L.ptr(q), L.ptr(k),
           L.ptr(v), L.ptr(mask), L.ptr(out), L.ptr(lse), L.ptr(go), L.ptr(dq), L.ptr(dk),
           L.ptr(dv), ctypes.c_float(0.0), ctypes.c_uint(0), L.ptr(None), L.stream_ptr())
    # float64 reference on the device (the same statements as _attn_ref; 0.5 GB of scores per batch element)
    heads = lambda t: t.double().reshape(t.shape[0], t.shape[1], h, D).permute(0, 2, 1, 3)
    q64, k64, v64 = (heads(t).detach().requires_grad_(True) for t in (q, k, v))
    ref, ref_lse = _attn_ref(q64, k64, v64, mask, scale)
    (ref * go.double()).sum().backward()
    rel = lambda got, want: float((got.double() - want).abs().max() / want.abs().max())
    untm = lambda t: t.reshape(t.shape[0], t.shape[1], h, D).permute(0, 2, 1, 3)
    assert rel(out, ref.detach()) < 1e-4                       # north-star fp32 bar; observed ~1e-6
    assert float((lse.double() - ref_lse.detach()).abs().max()) < 1e-4
    assert rel(untm(dq), q64.grad) < 1e-4
    assert rel(untm(dk), k64.grad) < 1e-4
    assert rel(untm(dv), v64.grad) < 1e-4


def _to_segments(t, seg, base2=None, rows=None, fill=0.0):
    """(b, n, c) plain token order -> two-segment row matrix (include/sig3d_hip.h); with base2 / rows
    the segments are padded (rows without a token hold `fill`)."""
    b, n, c = t.shape
    base2 = b * seg if base2 is None else base2
    rows = base2 + b * (n - seg) if rows is None else rows
    out = torch.full((rows, c), fill, dtype=t.dtype, device=t.device)
    out[:b * seg] = t[:, :seg].reshape(b * seg, c)
    out[base2:base2 + b * (n - seg)] = t[:, seg:].reshape(b * (n - seg), c)
    return out


@pytest.mark.parametrize("b,h,n,seg,pad", [(8, 12, 52, 32, 0), (3, 2, 40, 1, 0), (2, 4, 33, 32, 0), (2, 1, 7, 7, 0),
                                           (8, 12, 52, 32, 96), (2, 3, 40, 10, 5)])
@pytest.mark.parametrize("small", ["1", "0"])
def test_attention_two_segment_layout_matches_plain(b, h, n, seg, pad, small, monkeypatch):
    """q_seg / k_seg only re-map token -> storage row: results must equal the plain-layout call
    bit for bit after un-permuting the rows (same arithmetic, same order); small-problem and generic backward."""
    monkeypatch.setenv("SIG3D_ATTN_BWD_SMALL", small)
    L = _lib()
    g = torch.Generator().manual_seed(n * 7 + seg)
    ld = h * 64
    q, k, v, go = (torch.randn(b, n, ld, generator=g).to(DEV) for _ in range(4))
    keep = (torch.rand(b, n, generator=g) > 0.2).float()
    keep[:, 0] = 1.0
    mask = ((1.0 - keep) * -10000.0).to(DEV)
    scale = ctypes.c_float(0.125)

    def run(qs, ks, base2, rows, q, k, v, go):
        nan = float("nan")   # outputs start as NaN: every row, padding included, must be written
        out, lse = torch.full_like(q, nan), torch.empty(b, h, n, device=DEV)
        dq, dk, dv = torch.full_like(q, nan), torch.full_like(k, nan), torch.full_like(v, nan)
        L.call("sig3d_attention_fwd", b, h, n, n, 64, qs, ks, base2, base2, rows, rows, ld, ld, ld, scale,
               L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(mask), L.ptr(out), L.ptr(lse), ctypes.c_float(0.0),
               ctypes.c_uint(0), L.ptr(None), 1, L.ptr(None), L.stream_ptr())
        L.call("sig3d_attention_bwd", b, h, n, n, 64, qs, ks, base2, base2, rows, rows, ld, ld, ld, scale,
               L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(mask), L.ptr(out), L.ptr(lse), L.ptr(go), L.ptr(dq), L.ptr(dk),
               L.ptr(dv), ctypes.c_float(0.0), ctypes.c_uint(0), L.ptr(None), L.stream_ptr())
        return out, lse, dq, dk, dv

    plain = run(n, n, 0, 0, q, k, v, go)
    # padded variant: `pad` extra rows after each segment, garbage (1e30) in the inputs' padding rows
    base2 = b * seg + pad
    rows = base2 + b * (n - seg) + pad
    segd = run(seg, seg, base2, rows if pad else 0, *(_to_segments(t, seg, base2, rows, fill=1e30) for t in (q, k, v, go)))
    assert torch.equal(plain[1], segd[1])  # lse is indexed by (batch, head, token) in both
    for name, a, c in zip(("out", "dq", "dk", "dv"), (plain[0],) + plain[2:], (segd[0],) + segd[2:]):
        # token rows identical bit for bit; rows without a token are exact zeros
        assert torch.equal(_to_segments(a.view(b, n, ld), seg, base2, rows, fill=0.0), c.view(rows, ld)), name


@pytest.mark.parametrize("b,h,nq,nk,splits", [(2, 3, 32, 5000, 7), (1, 2, 40, 1000, 32), (1, 1, 5, 77, 3),
                                              (1, 2, 32, 9 * 32, 4), (1, 12, 32, 80000, 64)])
def test_attention_fwd_key_splits_match_single_pass(b, h, nq, nk, splits):
    """key_splits only re-associates the streaming softmax over key ranges: same result as one pass
    within rounding (1e-5), lse included; more splits than 32-key tiles are clamped.  9 tiles / 4 splits and
    2500 tiles / 64 splits (what qformer._fwd_key_splits picks at B = 1, Nk = 80 000) leave the LAST split
    without a tile at ceil(tiles / splits) tiles per split: round 3 found that split writing NaN partials."""
    L = _lib()
    g = torch.Generator().manual_seed(nk)
    ld = h * 64
    q, k, v = (torch.randn(b, n, ld, generator=g).to(DEV) for n in (nq, nk, nk))
    mask = ((torch.rand(b, nk, generator=g) < 0.1).float() * -10000.0).to(DEV)
    mask[:, 0] = 0.0
    outs = []
    for ks in (1, splits):
        out, lse = torch.empty_like(q), torch.empty(b, h, nq, device=DEV)
        work = torch.full((b * h * ((nq + 31) // 32) * 32 * ks * 66,), float("nan"), device=DEV)
        L.call("sig3d_attention_fwd", b, h, nq, nk, 64, nq, nk, 0, 0, 0, 0, ld, ld, ld, ctypes.c_float(0.125),
               L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(mask), L.ptr(out), L.ptr(lse), ctypes.c_float(0.0),
               ctypes.c_uint(0), L.ptr(None), ks, L.ptr(work), L.stream_ptr())
        outs.append((out, lse))
    torch.testing.assert_close(outs[1][0], outs[0][0], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(outs[1][1], outs[0][1], rtol=1e-5, atol=1e-5)


def test_ticket_handshake_orders_two_streams():
    """sig3d_ticket_signal / sig3d_ticket_wait (csrc/capi.hip): stream B's work starts after stream A's signal without
    a barrier packet between them.  B copies a buffer A fills just before each signal; 40 rounds, enqueued B first."""
    from situation3d_amd import _lib
    dev = torch.device(DEV)
    a_s, b_s = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    words = torch.zeros(4, dtype=torch.int32, device=dev)
    src = torch.zeros(1 << 20, device=dev)
    got = torch.zeros(40, device=dev)
    torch.cuda.synchronize()
    for k in range(40):
        with torch.cuda.stream(b_s):                        # the waiter is enqueued FIRST: it must spin, not race
            _lib.call("sig3d_ticket_wait", _lib.ptr(words[0:1]), _lib.ptr(words[1:2]), 5000000, _lib.ptr(words[2:3]),
                      _lib.stream_ptr(dev))
            got[k:k + 1].copy_(src[-1:])
            done = torch.cuda.Event(); done.record(b_s)
        with torch.cuda.stream(a_s):
            src.fill_(float(k + 1))
            _lib.call("sig3d_ticket_signal", _lib.ptr(words[0:1]), _lib.stream_ptr(dev))
            # the NEXT fill must not overtake this round's copy: A waits for B's copy (an ordinary event)
            a_s.wait_event(done)
    torch.cuda.synchronize()
    assert words[:3].tolist() == [40, 40, 0]
    assert got.tolist() == [float(k + 1) for k in range(40)]


def test_ticket_wait_times_out_instead_of_hanging():
    from situation3d_amd import _lib
    dev = torch.device(DEV)
    words = torch.zeros(4, dtype=torch.int32, device=dev)
    _lib.call("sig3d_ticket_wait", _lib.ptr(words[0:1]), _lib.ptr(words[1:2]), 2000, _lib.ptr(words[2:3]),
              _lib.stream_ptr(dev))                        # nobody signals: gives up after 2 ms
    torch.cuda.synchronize()
    assert words[:3].tolist() == [0, 1, 1]


def test_cu_masked_stream_confines_workgroups():
    """sig3d_stream_create_with_cu_mask + sig3d_whereami: mask bit k = XCD k % 8, CU slot k / 8; two disjoint masks
    run on disjoint CUs (situation3d_amd/streams.py)."""
    from situation3d_amd import _lib, streams
    dev = torch.device(DEV)

    def where(stream, blocks=512):
        slots = torch.zeros(2 * blocks, dtype=torch.int32, device=dev)
        with torch.cuda.stream(stream):
            _lib.call("sig3d_whereami", _lib.ptr(slots), blocks, 256, 100, _lib.stream_ptr(dev))
        torch.cuda.synchronize()
        v = slots.cpu().view(blocks, 2)
        return {(int(x) & 0xF, (int(h) >> 8) & 0xFF) for h, x in v.tolist()}

    a = streams.MaskedStream(dev, streams.cu_mask(0, 2))
    b = streams.MaskedStream(dev, streams.cu_mask(2, 32))
    ca, cb = where(a.stream), where(b.stream)
    assert len(ca) == 16 and {x for x, _ in ca} == set(range(8))
    assert len(cb) == 240 and not (ca & cb)
    a.close(); b.close()


def test_stream_beside_finds_a_free_hardware_queue():
    """streams.run_concurrently / stream_beside: of a dozen fresh streams some pairs share a hardware queue (served in
    order); stream_beside returns one that overlaps with every stream it is given."""
    from situation3d_amd import streams
    dev = torch.device(DEV)
    main = torch.cuda.Stream(dev)
    assert not streams.run_concurrently(main, main, dev)          # one stream: in order by definition
    picked = [main]
    for _ in range(3):                                            # 1 + 3 streams: the four queues of one priority
        st, ok = streams.stream_beside(picked, dev)
        assert ok
        for o in picked:
            assert streams.run_concurrently(o, st, dev)
        picked.append(st)
