"""CPU suite, part 2: host logic and the drop-in boundary (no GPU compute).

  * the C-ABI library loads and exports every symbol include/sig3d_hip.h declares;
  * the product `pointnet2._ext` refuses host tensors like the reference does;
  * the host-side mirror of pointnet2_modules.py reproduces the REFERENCE modules' outputs and
    gradients (golden fixtures) when its `_ext` is temporarily bound to the CPU oracle -- this
    checks class names, ctor kwargs, state_dict keys and forward/backward wiring.
"""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def _load(name):
    return {k: torch.from_numpy(np.array(v, copy=True))
            for k, v in np.load(os.path.join(GOLD, name), allow_pickle=False).items()
            if v.dtype.kind in "fiub"}


# ---- boundary ------------------------------------------------------------------------------
def _declared_symbols(headers=("sig3d_hip.h", "sig3d_debug.h")):
    """Every entry point include/*.h declares (the drop-in boundary + the measurement header)."""
    names = set()
    for h in headers:
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(sig3d_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


def test_measurement_entry_points_are_not_in_the_boundary_header():
    boundary = _declared_symbols(("sig3d_hip.h",))
    for name in ("sig3d_hold", "sig3d_whereami", "sig3d_stream_create_with_cu_mask", "sig3d_timestamp"):
        assert name not in boundary and name in _declared_symbols(("sig3d_debug.h",))
    assert sorted(os.listdir(os.path.join(ROOT, "include"))) == ["sig3d_debug.h", "sig3d_hip.h"]


def test_library_exports_every_declared_symbol():
    from situation3d_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 17
    for name in declared:
        assert hasattr(lib, name), "libsig3d_hip.so does not export %s" % name
    # the Python binding table covers the same set (minus the two info getters)
    assert sorted(list(_lib.SIGNATURES) + list(_lib.INFO_SYMBOLS)) == declared
    assert "gfx950" in _lib.version()


def test_library_is_gfx950_code_object():
    from situation3d_amd import _lib
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in blob and b"gfx942" not in blob and b"sm_" not in blob


def test_ext_surface_matches_reference_bindings():
    """bindings.cpp:6-19: nine functions, positional."""
    import pointnet2._ext as ext
    for fn in ("gather_points", "gather_points_grad", "furthest_point_sampling", "three_nn",
               "three_interpolate", "three_interpolate_grad", "ball_query", "group_points",
               "group_points_grad"):
        assert callable(getattr(ext, fn))


def test_ext_rejects_cpu_tensors_like_reference():
    """AT_ASSERT(false, "CPU not supported") in every reference wrapper (e.g. sampling.cpp:33-35)."""
    import pointnet2._ext as ext
    xyz = torch.rand(1, 8, 3)
    f = torch.rand(1, 2, 8)
    i1 = torch.zeros(1, 4, dtype=torch.int32)
    i3 = torch.zeros(1, 4, 3, dtype=torch.int32)
    w = torch.rand(1, 4, 3)
    calls = [lambda: ext.furthest_point_sampling(xyz, 4), lambda: ext.gather_points(f, i1),
             lambda: ext.gather_points_grad(torch.rand(1, 2, 4), i1, 8),
             lambda: ext.ball_query(xyz, xyz, 0.5, 3), lambda: ext.group_points(f, i3),
             lambda: ext.group_points_grad(torch.rand(1, 2, 4, 3), i3, 8),
             lambda: ext.three_nn(xyz, xyz), lambda: ext.three_interpolate(f, i3, w),
             lambda: ext.three_interpolate_grad(torch.rand(1, 2, 4), i3, w, 8)]
    for c in calls:
        with pytest.raises(RuntimeError, match="CPU not supported"):
            c()
    with pytest.raises(RuntimeError, match="must be a float tensor"):
        ext.furthest_point_sampling(xyz.double(), 4)
    with pytest.raises(RuntimeError, match="must be an int tensor"):
        ext.gather_points(f, i1.long())
    with pytest.raises(RuntimeError, match="must be a contiguous tensor"):
        ext.three_nn(xyz.transpose(1, 2), xyz)


def test_product_code_never_imports_the_oracle():
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "situation3d_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b|oracle/_build|liboracle", txt, flags=re.M):
                    bad.append(f)
    assert not bad, bad


# ---- host-side mirror vs the reference modules (goldens) ------------------------------------
@pytest.fixture()
def mirror(monkeypatch, oracle):
    """Mirror modules with `_ext` bound to the CPU oracle (test wiring only)."""
    from situation3d_amd.pointnet2 import pointnet2_modules, pointnet2_utils
    monkeypatch.setattr(pointnet2_utils, "_ext", oracle)
    return pointnet2_modules


def _check_module(mod, g, args, out_index, grads):
    state = {k[len("state."):]: v for k, v in g.items() if k.startswith("state.")}
    mod.load_state_dict(state, strict=True)  # key-for-key parity with the reference module
    mod.train()
    outs = mod(*args)
    outs = outs if isinstance(outs, tuple) else (outs,)
    for i, o in enumerate(outs):
        if o is None:
            continue
        ref = g["out%d" % i]
        if o.dtype in (torch.int32, torch.int64):
            assert torch.equal(o.to(ref.dtype), ref)
        else:
            torch.testing.assert_close(o, ref, rtol=1e-4, atol=1e-4)
    (outs[out_index] * g["G"]).sum().backward()
    # gradients are long f32 reductions (up to 131072 terms) whose order depends on the BLAS
    # thread count: tolerance 1e-4 relative to the tensor's scale
    def close(a, b, name):
        torch.testing.assert_close(a, b, rtol=1e-3, atol=1e-4 * max(1.0, b.abs().max().item()),
                                   msg=lambda m: name + ": " + m)

    for name, t in grads.items():
        close(t.grad, g[name], name)
    for k, p in mod.named_parameters():
        close(p.grad, g["grad." + k], k)
    for k, v in mod.state_dict().items():  # BatchNorm running statistics after one step
        torch.testing.assert_close(v, g["state_after." + k].to(v.dtype), rtol=1e-4, atol=1e-5, msg=k)


def test_mirror_msg_smoke_config(mirror):
    """pointnet2_modules.py:504-523 (the reference's own smoke configuration)."""
    g = _load("pointnet2_modules_msg_smoke.npz")
    xyz = g["xyz"].clone().requires_grad_(True)
    f = g["features"].clone().requires_grad_(True)
    mod = mirror.PointnetSAModuleMSG(npoint=2, radii=[5.0, 10.0], nsamples=[6, 3], mlps=[[6, 3], [6, 6]])
    _check_module(mod, g, (xyz, f), 1, {"grad_xyz": xyz, "grad_features": f})


def test_mirror_sa1_config1(mirror):
    """BASELINE config 1: single synthetic scene, 4096 pts, SA1 forward(+backward) on the CPU path."""
    g = _load("pointnet2_modules_sa1_4096.npz")
    f = g["features"].clone().requires_grad_(True)
    mod = mirror.PointnetSAModuleVotes(npoint=2048, radius=0.2, nsample=64, mlp=[3, 64, 64, 128],
                                       use_xyz=True, normalize_xyz=True)
    _check_module(mod, g, (g["xyz"], f), 1, {"grad_features": f})


@pytest.mark.parametrize("pooling", ["avg", "rbf"])
def test_mirror_votes_pooling(mirror, pooling):
    g = _load("pointnet2_modules_votes_%s.npz" % pooling)
    f = g["features"].clone().requires_grad_(True)
    mod = mirror.PointnetSAModuleVotes(npoint=37, radius=0.9, nsample=12, mlp=[5, 16, 8],
                                       use_xyz=True, pooling=pooling)
    _check_module(mod, g, (g["xyz"], f), 1, {"grad_features": f})


def test_mirror_fp_module(mirror):
    g = _load("pointnet2_modules_fp.npz")
    uf = g["unknow_feats"].clone().requires_grad_(True)
    kf = g["known_feats"].clone().requires_grad_(True)
    mod = mirror.PointnetFPModule(mlp=[18, 16, 16])
    _check_module(mod, g, (g["unknown"], g["known"], uf, kf), 0,
                  {"grad_unknow_feats": uf, "grad_known_feats": kf})


def test_qformer_state_dict_keys_match_reference():
    """Key-for-key parity with the reference's Qformer.bert state_dict (golden was produced by the
    reference's BertModel), and with the Blip2T5 stripping (blip2_t5.py:63-69)."""
    from situation3d_amd.qformer import QFormer, QFormerConfig
    g = _load("qformer_small.npz")
    c = g["config"].tolist()
    cfg = QFormerConfig(vocab_size=c[0], hidden_size=c[1], num_hidden_layers=c[2],
                        num_attention_heads=c[3], intermediate_size=c[4],
                        max_position_embeddings=c[5], encoder_width=c[6], cross_attention_freq=c[7],
                        query_length=c[8])
    model = QFormer(cfg)
    ref_keys = sorted(k[len("state."):] for k in g if k.startswith("state."))
    assert sorted(model.bert.state_dict().keys()) == ref_keys
    model.strip_text_branch()
    kept = [k for k in model.bert.state_dict() if "word_embeddings" in k or ".intermediate." in k
            or ".output." in k and "attention" not in k]
    assert kept == []


def test_ctypes_structs_match_the_c_header(tmp_path):
    """The ctypes mirrors of the C-ABI structs (sig3d_bq_level, sig3d_column_sum_job, sig3d_sum_slabs_job, sig3d_gemm16_problem,
    sig3d_gemmp_problem) must agree with what a C compiler makes of include/sig3d_hip.h: size and every field offset,
    checked by compiling a probe with gcc."""
    import ctypes
    import subprocess
    from situation3d_amd import _lib
    structs = {"sig3d_bq_level": _lib.BqLevel, "sig3d_column_sum_job": _lib.ColumnSumJob,
               "sig3d_sum_slabs_job": _lib.SumSlabsJob,
               "sig3d_gemm16_problem": _lib.Gemm16Problem, "sig3d_gemmp_problem": _lib.GemmpProblem}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "sig3d_hip.h"', 'int main(void) {']
    for cname, cls in structs.items():
        lines.append('printf("%s size %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, _ in cls._fields_:
            lines.append('printf("%s %s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, fname))
    lines += ['return 0;', '}']
    src = tmp_path / "probe.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "probe"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = {}
    for ln in subprocess.check_output([str(exe)], text=True).splitlines():
        cname, fname, val = ln.split()
        got[(cname, fname)] = int(val)
    for cname, cls in structs.items():
        assert got[(cname, "size")] == ctypes.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert got[(cname, fname)] == getattr(cls, fname).offset, (cname, fname)


def test_point_major_twins_and_scan_attachment_are_validated():
    """fused_mlp.point_major_of / first_layer_scan hand out the side-channel tensors only when they really describe
    the features they ride on (shape, layout, device); anything else falls back to the reference behaviour."""
    from situation3d_amd.pointnet2 import fused_mlp
    f = torch.zeros(2, 8, 5)
    assert fused_mlp.point_major_of(f) is None
    f._pm = torch.zeros(2, 5, 8)
    assert fused_mlp.point_major_of(f) is f._pm
    f._pm = torch.zeros(2, 5, 7)                      # wrong width
    assert fused_mlp.point_major_of(f) is None
    f._pm = torch.zeros(2, 8, 5).transpose(1, 2)      # right shape, not contiguous
    assert fused_mlp.point_major_of(f) is None
    g = f * 2                                         # a new tensor carries no twin
    assert fused_mlp.point_major_of(g) is None
    pc = torch.rand(2, 9, 6)
    feats = fused_mlp.attach_scan(pc[..., 3:].transpose(1, 2), pc)
    assert feats._points_pm is pc and not feats.is_contiguous()
    assert fused_mlp.first_layer_scan(None, pc[..., :3], feats, True) is None      # CPU: never the HIP path


def test_gemmp_work_space_and_argument_checks_are_host_arithmetic():
    """sig3d_gemmp_work_floats sizes the partial tiles of a split reduction from the tiling the launcher will choose
    (no GPU call); sig3d_gemmp refuses inconsistent problems before it touches a device pointer."""
    from situation3d_amd import _lib
    assert _lib.gemmp_work_floats(1, 416, 2304, 1) == 0
    assert _lib.gemmp_work_floats(1, 416, 2304, 2, 2) == 2 * 7 * 18 * 64 * 128          # 64 x 128 tiles
    assert _lib.gemmp_work_floats(1, 1536, 1408, 4, 3) == 4 * 12 * 11 * 128 * 128       # 128 x 128 tiles
    lib = _lib.load()
    q = _lib.GemmpProblem()
    q.A, q.B, q.C = 4096, 8192, 16384            # never dereferenced: the checks come first
    q.chunk_a = q.chunk_b = 64 * 96
    q.stride_a = q.stride_b = 64 * 96 * 4
    q.bytes_a = q.bytes_b = 64 * 96 * 4 * 2
    q.ldc, q.batch, q.m, q.n, q.k, q.splits = 64, 1, 64, 64, 100, 1
    import ctypes
    assert lib.sig3d_gemmp(ctypes.byref(q), None) != 0 and b"multiple of 32" in lib.sig3d_last_error()
    q.k, q.modes = 128, 3
    assert lib.sig3d_gemmp(ctypes.byref(q), None) != 0 and b"modes" in lib.sig3d_last_error()
    q.modes, q.splits = 0, 2
    assert lib.sig3d_gemmp(ctypes.byref(q), None) != 0 and b"work space" in lib.sig3d_last_error()
    q.splits, q.bytes_a = 1, 1 << 33
    assert lib.sig3d_gemmp(ctypes.byref(q), None) != 0 and b"4 GB" in lib.sig3d_last_error()


def test_ball_query_levels_workspace_is_host_arithmetic():
    """sig3d_ball_query_levels_workspace_bytes plans a multi-level launch on the host (no GPU call): scenes of at most
    4096 points take the in-LDS ordered scan and need no scratch; larger ones a counter + 256 list slots per centre;
    more than 16 blocks of 4096 centres do not fit one launch (-1)."""
    from situation3d_amd import _lib
    lib = _lib.load()

    def levels(specs):
        arr = (_lib.BqLevel * len(specs))()
        for q, (n, m, ns, r) in zip(arr, specs):
            q.n, q.m, q.nsample, q.radius = n, m, ns, r
        return arr

    stack = levels([(40000, 2048, 64, 0.2), (2048, 1024, 32, 0.4), (1024, 512, 16, 0.8), (512, 256, 16, 1.2)])
    assert lib.sig3d_ball_query_levels_workspace_bytes(8, 4, stack) == 8 * 2048 * 4 * 257   # SA1 only
    assert lib.sig3d_ball_query_levels_workspace_bytes(8, 3, levels([(2048, 1024, 32, 0.4)] * 3)) == 0
    assert lib.sig3d_ball_query_levels_workspace_bytes(2, 1, levels([(50000, 5000, 8, 0.3)])) == \
        2 * 5000 * 4 * 257        # two blocks of centres
    assert lib.sig3d_ball_query_levels_workspace_bytes(1, 1, levels([(50000, 17 * 4096, 8, 0.3)])) == -1
    assert lib.sig3d_ball_query_levels_workspace_bytes(0, 1, stack) == 0


def test_block_list_fps_and_level_grouping_check_their_arguments_before_any_launch():
    """The two entry points of round 6 refuse an inconsistent call on the host, before a device pointer is touched
    (the reference raises through AT_ASSERT / TORCH_CHECK before its launch, sampling.cpp:9-12, group_points.cpp:9-12);
    the FPS work space is host arithmetic: rows {x, y, z, key} + one distance per padded point."""
    import ctypes
    from situation3d_amd import _lib
    lib = _lib.load()
    ws = lib.sig3d_fps_blocks_workspace_bytes
    assert ws(8, 40000) == 8 * 40000 * 20 and ws(1, 40001) == 40064 * 20        # padded to whole 64-point blocks
    assert ws(0, 40000) == 0 and ws(-1, 5) == -1
    fps = lib.sig3d_furthest_point_sampling_blocks
    P = ctypes.c_void_p
    assert fps(0, 40000, 2048, P(64), P(0), 0, 0, P(64), None) == 0              # empty batch: nothing to do
    assert fps(1, 40000, 0, P(64), P(0), 0, 0, P(64), None) == 0                 # m = 0 (sampling_gpu.cu:73)
    assert fps(1, 40000, 2048, P(64), P(0), 0, 0, P(64), None) != 0 and b"workspace" in lib.sig3d_last_error()
    assert fps(1, 40000, 2048, P(64), P(4096), ws(1, 40000) - 1, 0, P(64), None) != 0
    assert fps(1, 40000, 2048, P(64), P(4100), ws(1, 40000), 0, P(64), None) != 0 and b"aligned" in lib.sig3d_last_error()
    assert fps(1, 40000, 2048, P(64), P(4096), ws(1, 40000), 5, P(64), None) != 0 and b"waves" in lib.sig3d_last_error()
    assert fps(1, -1, 2048, P(64), P(4096), 1 << 20, 0, P(64), None) != 0

    grp = lib.sig3d_query_group_levels
    lv = (_lib.GroupLevel * 5)()
    for q in lv:
        q.n, q.m, q.c, q.ld, q.nsample, q.point_major, q.use_xyz, q.normalize_xyz = 2048, 1024, 128, 128, 32, 1, 1, 1
        q.radius = 0.4
        q.xyz = q.new_xyz = q.features = q.idx = q.out = 4096
    assert grp(0, 4, lv, None) == 0                                              # empty batch
    assert grp(8, 5, lv, None) != 0 and b"1 to 4 levels" in lib.sig3d_last_error()
    assert grp(8, 0, lv, None) != 0
    lv[1].c, lv[1].use_xyz = 0, 0
    assert grp(8, 2, lv, None) != 0 and b"Cannot have not features" in lib.sig3d_last_error()   # pointnet2_utils.py:368
    lv[1].c, lv[1].use_xyz, lv[1].ld = 128, 1, 126
    assert grp(8, 2, lv, None) != 0 and b"point-major" in lib.sig3d_last_error()
    lv[1].ld, lv[1].point_major, lv[1].nsample = 128, 0, 30
    assert grp(8, 2, lv, None) != 0 and b"nsample % 4" in lib.sig3d_last_error()
    lv[1].nsample, lv[1].m = 64, 1 << 26
    assert grp(8, 2, lv, None) != 0 and b"too large" in lib.sig3d_last_error()


def test_storage_layout_keeps_two_arenas_when_the_backward_pass_is_cut_inside_the_qformer():
    """trainer.storage_layout: the order FlatAdamW lays the parameters out in.  One arena: every kind runs over all
    layers (the layer-batched weight-gradient buffers are slices of the flat gradients).  qf_cut = k: the layers below
    and from k on are two arenas, each ONE stretch of every parameter group -- a backward pass cut there (graph_step)
    then owns one stretch per piece instead of a slice per kind -- and inside an arena the kinds still run over its
    layers, [Wq; Wk; Wv] back to back."""
    import torch.nn as nn
    from situation3d_amd.qformer import init_Qformer
    from situation3d_amd.trainer import storage_layout

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.pre = nn.Linear(8, 8)
            self.Qformer, self.query_tokens = init_Qformer(32, 64, hidden_size=64, num_hidden_layers=6,
                                                           num_attention_heads=2, intermediate_size=128,
                                                           max_position_embeddings=64)
            self.head = nn.Linear(64, 4)

    net = Net()
    names = {id(p): n for n, p in net.named_parameters()}
    decay = [p for n, p in net.named_parameters() if "bias" not in n and "LayerNorm.weight" not in n]
    no_decay = [p for n, p in net.named_parameters() if "bias" in n or "LayerNorm.weight" in n]

    def layer_of(p):
        n = names[id(p)]
        return int(n.split("encoder.layer.")[1].split(".")[0]) if "encoder.layer." in n else None

    for cut in (None, 4):
        for lst, laid in zip((decay, no_decay), storage_layout(net, [decay, no_decay], qf_cut=cut)):
            assert sorted(map(id, laid)) == sorted(map(id, lst))                  # a permutation
            side = [None if layer_of(p) is None else int(layer_of(p) >= (cut or 0)) for p in laid]
            layered = [x for x in side if x is not None]
            if cut:
                # lower arena, then upper arena: one switch, and no non-layer parameter in between
                assert layered == sorted(layered) and 0 < sum(layered) < len(layered)
                first, last = side.index(0), len(side) - 1 - side[::-1].index(1)
                assert None not in side[first:last + 1]
            order = [names[id(p)] for p in laid]
            for l in range(6):       # [Wq; Wk; Wv] (or their biases) of a layer are adjacent in either layout
                pre = "Qformer.bert.encoder.layer.%d.attention.self." % l
                kind = "weight" if lst is decay else "bias"
                i = order.index(pre + "query." + kind)
                assert order[i + 1:i + 3] == [pre + "key." + kind, pre + "value." + kind]
        dec = [names[id(p)] for p in storage_layout(net, [decay, no_decay], qf_cut=cut)[0]]
        i = dec.index("Qformer.bert.encoder.layer.0.attention.self.value.weight")
        nxt = "Qformer.bert.encoder.layer.1.attention.self.query.weight"
        assert dec[i + 1] == nxt                                                   # kind-major inside an arena
        i = dec.index("Qformer.bert.encoder.layer.3.attention.self.value.weight")
        after = dec[i + 1]
        assert (after == "Qformer.bert.encoder.layer.4.attention.self.query.weight") == (cut is None)


def test_runs_index_like_the_tensor_they_replace_and_refuse_a_range_across_the_cut():
    """qformer._Runs: a layer-batched buffer stored as one run per parameter arena.  Indexing by a layer, by a
    (layer, ...) tuple and by a range inside one run must return what the single tensor would; a range that straddles the
    storage cut is an error (flush() cuts its ranges at the boundary and never issues one)."""
    import pytest
    from situation3d_amd.qformer import _Runs
    whole = torch.arange(6 * 2 * 3, dtype=torch.float32).view(6, 2, 3)
    runs = _Runs([(0, 4, whole[:4].clone()), (4, 6, whole[4:].clone())])
    for l in range(6):
        assert torch.equal(runs[l], whole[l]) and torch.equal(runs[l, 1], whole[l, 1])
    assert torch.equal(runs[0:4], whole[0:4]) and torch.equal(runs[4:6], whole[4:6]) and torch.equal(runs[1:3, 0], whole[1:3, 0])
    assert torch.equal(runs[:4], whole[:4]) and torch.equal(runs[4:], whole[4:])
    with pytest.raises(IndexError):
        runs[3:5]
    # row-indexed form (the stacked key / value weight gradients: 2H rows per cross layer)
    rows = torch.arange(12 * 5, dtype=torch.float32).view(12, 5)
    kv = _Runs([(0, 8, rows[:8].clone()), (8, 12, rows[8:].clone())])
    assert torch.equal(kv[4:8], rows[4:8]) and torch.equal(kv[8:10], rows[8:10])
    like = kv.empty_like()
    assert [t.shape for t in like.tensors()] == [t.shape for t in kv.tensors()] and like.tensors()[0] is not kv.tensors()[0]
    kv[8:12].fill_(7.0)                                  # views: a write through an index lands in the run's storage
    assert float(kv.tensors()[1].min()) == 7.0


def test_capture_tables_hand_out_every_set_once():
    from situation3d_amd.optim import CaptureTables
    import pytest
    t = CaptureTables(["a", "b"])
    assert t.take() == "a" and t.take() == "b"
    with pytest.raises(RuntimeError, match="more gradient tables than were reserved"):
        t.take()


def test_comm_model_states_both_forms_and_the_cut_never_costs_more():
    """bench.comm_model: the prediction a multi-GPU line carries -- both data-parallel forms at N = 2 / 4 / 8, the second
    cut putting half of the bytes on the wire earlier, so its predicted step is never the longer one; the line's own
    form follows --qf-cut."""
    import bench
    nbytes = 615_044_192
    m = bench.comm_model(8, nbytes, 7.4, qf_cut=6)
    assert set(m["forms"]) == {"N=2", "N=4", "N=8", "N=8 at rccl 180 GB/s"} and m["form"] == "cut 6"
    for n, f in m["forms"].items():
        assert f["cut"]["predicted_ms_per_step"] <= f["uncut"]["predicted_ms_per_step"], n
        assert f["cut"]["exposed_ms"] <= f["uncut"]["exposed_ms"] and f["cut"]["wire_ms"] == f["uncut"]["wire_ms"]
    assert abs(m["forms"]["N=2"]["uncut"]["wire_ms"] - nbytes / 76.8e9 * 1e3) < 1e-2      # one link between two GPUs
    assert m["predicted_ms_per_step"] == m["forms"]["N=8"]["cut"]["predicted_ms_per_step"]
    assert bench.comm_model(8, nbytes, 7.4, qf_cut=0)["form"] == "uncut"
    assert bench.comm_model(1, nbytes)["predicted_ms_per_step"] is None and "forms" in bench.comm_model(1, nbytes)


def test_ctypes_signatures_have_the_arity_and_argument_classes_of_the_header_prototypes():
    """Every entry point bound in situation3d_amd._lib.SIGNATURES takes as many arguments as its prototype in
    include/sig3d_hip.h / sig3d_debug.h declares, each of the prototype's class (int / long / float / double / pointer): a
    drifted ctypes table only shows on a GPU box, as a wrong result."""
    from situation3d_amd import _lib
    protos = {}
    for h in ("sig3d_hip.h", "sig3d_debug.h"):
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        text = re.sub(r"//[^\n]*", "", text)
        for m in re.finditer(r"\b(?:int|long|const char \*)\s*(sig3d_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
            args = m.group(2).strip()
            protos[m.group(1)] = [] if args in ("", "void") else [a.strip() for a in args.split(",")]
    assert len(protos) >= 100

    def kind(decl):            # C parameter -> the class of ctypes type that may carry it
        if "*" in decl or "[" in decl:
            return "pointer"
        base = decl.rsplit(" ", 1)[0].replace("const", "").replace("unsigned", "").strip() if " " in decl else decl
        return {"int": "int", "long": "long", "float": "float", "double": "double", "": "int",
                "long long": "long", "size_t": "long"}.get(base, base)

    def ctype_kind(t):
        if t in (ctypes.c_void_p,) or hasattr(t, "contents") or getattr(t, "_type_", None) not in ("i", "l", "f", "d", "I", "L", "q", "Q"):
            return "pointer"
        return {"i": "int", "I": "int", "l": "long", "L": "long", "q": "long", "Q": "long", "f": "float", "d": "double"}[t._type_]

    checked = 0
    for name, argtypes in _lib.SIGNATURES.items():
        assert name in protos, "%s is bound but not declared" % name
        assert len(argtypes) == len(protos[name]), "%s: %d ctypes arguments, %d in the header" % (name, len(argtypes), len(protos[name]))
        for i, (t, decl) in enumerate(zip(argtypes, protos[name])):
            assert ctype_kind(t) == kind(decl), "%s argument %d: ctypes %s for `%s`" % (name, i, t, decl)
        checked += 1
    assert checked == len(_lib.SIGNATURES) >= 90


def test_weight_gradient_arena_follows_a_storage_cut():
    """qformer._WeightGradArena over two parameter arenas (storage_cut = k: the layout trainer.build_optimizer(qf_cut=k)
    gives the Q-Former), on the host: every layer parameter still gets exactly one gradient view of its own shape, the
    layer-batched buffers and the stacked key / value weights have a run per arena, a flush over all layers is issued as one
    batch per arena (upper arena first), and the scene-token gradient of the two runs equals the single-run product."""
    from situation3d_amd.qformer import _Runs, _WeightGradArena, init_Qformer
    torch.manual_seed(0)
    qf, _ = init_Qformer(32, 64, hidden_size=64, num_hidden_layers=4, num_attention_heads=2, intermediate_size=128,
                         max_position_embeddings=64, vocab_size=50)
    layers = list(qf.bert.encoder.layer)
    enc2 = torch.randn(2 * 40, 64)

    def arena(cut):
        return _WeightGradArena(layers, 2, 32, 20, 64, enc2, 2, [0], grad_store=None, storage_cut=cut)

    one, two = arena(None), arena(2)
    assert one.runs == [(0, 4)] and two.runs == [(0, 2), (2, 4)] and two.cross_runs == [(0, 1), (1, 2)]
    assert isinstance(two.gwqkv, _Runs) and isinstance(two.gwkv, _Runs) and not isinstance(one.gwqkv, _Runs)
    layer_params = {id(p) for l in layers for p in l.parameters()}
    for a in (one, two):
        views = a.param_views(0, 4)
        assert {id(p) for p, _ in views} == layer_params and len(views) == len(layer_params)
        assert all(tuple(v.shape) == tuple(p.shape) for p, v in views)
        spans = sorted((v.data_ptr(), v.data_ptr() + 4 * v.numel()) for _, v in views)
        assert all(a1 <= b0 for (_, a1), (b0, _) in zip(spans, spans[1:]))          # no two views overlap
    # the stacked key / value weights: one zero-copy-or-cat view per arena, same rows as the single stack
    assert len(two.wkv_runs) == 2 and len(one.wkv_runs) == 1
    assert torch.equal(torch.cat([w for _, _, w in two.wkv_runs]), one.wkv_runs[0][2])
    assert torch.allclose(two.kv, one.kv, atol=1e-5)                                  # the projection of the scene tokens
    # a flush over every layer: one batch of products per arena, the upper one first
    calls = []
    two._products = lambda lo, hi: calls.append((lo, hi))
    two._products_cut(0, 4)
    two._products_cut(1, 2)
    two._products_cut(1, 3)
    assert calls == [(2, 4), (0, 2), (1, 2), (2, 3), (1, 2)]
    # the gradient of the scene tokens over both cross layers
    dkv = torch.randn_like(one.dkv)
    one.dkv.copy_(dkv); two.dkv.copy_(dkv)
    torch.testing.assert_close(two.g_enc(0), one.g_enc(0), rtol=1e-5, atol=1e-5)
