"""Parity of the nine HIP PointNet++ ops (through pointnet2._ext -> C ABI) against the CPU
oracle on identical seeded inputs.  Index outputs must be BIT-EXACT; gathers are bit-exact;
scatter-add gradients (float atomics, order-dependent in the reference too:
group_points_gpu.cu:59-60) within 1e-5 relative.
"""
import pytest
import torch

from util import feats, scene

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _fps_case(hip_ext, oracle, b, n, m, **kw):
    xyz = scene(b, n, **kw)
    ref = oracle.furthest_point_sampling(xyz, m)
    got = hip_ext.furthest_point_sampling(xyz.to(DEV), m).cpu()
    assert got.dtype == torch.int32 and got.shape == (b, m)
    assert torch.equal(got, ref), "FPS indices differ at %s" % (got != ref).nonzero()[:4].tolist()


@pytest.mark.parametrize("n,m", [(1, 1), (3, 2), (9, 2), (64, 16), (100, 37), (255, 64), (256, 64),
                                 (300, 100), (511, 128), (512, 128), (513, 128), (1000, 256),
                                 (1024, 512), (2048, 1024), (4096, 512), (5000, 300), (9000, 256),
                                 (20000, 128)])
def test_fps_sizes(hip_ext, oracle, n, m):
    _fps_case(hip_ext, oracle, 2, n, m, seed=n)


@pytest.fixture(params=["blocks-16", "blocks-8", "blocks-4", "coop"])
def fps_path(request, monkeypatch):
    """Scenes above 8192 points: the block-list kernel behind sig3d_furthest_point_sampling_blocks (one workgroup of 16 /
    8 / 4 waves per scene over a Morton-ordered copy in L2: a stand-alone call takes 16, a chain beside the training step
    4) or the cooperative register-resident kernel behind the reference's own argument list."""
    from situation3d_amd.pointnet2 import _ext as amd_ext
    monkeypatch.setattr(amd_ext, "FPS_BLOCKS", request.param != "coop")
    if request.param != "coop":
        monkeypatch.setattr(amd_ext, "FPS_WAVES", int(request.param.split("-")[1]))
    return request.param


@pytest.mark.parametrize("b,n,m", [(3, 8193, 200), (9, 10000, 150), (2, 16384, 200), (2, 16385, 100), (2, 24576, 150),
                                   (1, 24577, 100), (2, 40960, 100), (1, 40961, 80), (2, 65536, 80), (1, 65537, 60),
                                   (1, 98304, 60), (1, 98305, 50), (1, 131072, 40), (1, 131073, 40), (1, 150000, 40)])
def test_fps_cooperative_instances(hip_ext, oracle, b, n, m, fps_path):
    """Every instance of the cooperative kernel (points per thread) and of the block-list kernel (points per lane and
    block) at and just past its size limit, batches that are not a multiple of the 8 scenes of one launch, ties and a
    zero tail included."""
    _fps_case(hip_ext, oracle, b, n, m, seed=n + b, dup=n // 20, zero_tail=n // 50)


def test_fps_ties_and_skips(hip_ext, oracle):
    # duplicated points (exact ties in min-distance) and an all-zero padded tail
    # (skipped by the mag <= 1e-3 rule, sampling_gpu.cu:100-101)
    _fps_case(hip_ext, oracle, 3, 4096, 1024, seed=5, dup=1500, zero_tail=700)
    _fps_case(hip_ext, oracle, 2, 700, 300, seed=6, dup=400, zero_tail=50)
    # grid-aligned points: massive exact ties
    g = torch.stack(torch.meshgrid(torch.arange(16.), torch.arange(16.), torch.arange(8.), indexing="ij"), -1)
    xyz = (g.reshape(1, -1, 3) * 0.25 + 0.5).contiguous()
    ref = oracle.furthest_point_sampling(xyz, 512)
    got = hip_ext.furthest_point_sampling(xyz.to(DEV), 512).cpu()
    assert torch.equal(got, ref)


def test_fps_all_skipped_and_m_gt_n(hip_ext, oracle):
    xyz = torch.zeros(2, 300, 3)
    assert torch.equal(hip_ext.furthest_point_sampling(xyz.to(DEV), 8).cpu(),
                       oracle.furthest_point_sampling(xyz, 8))
    xyz = scene(1, 40, seed=3)
    assert torch.equal(hip_ext.furthest_point_sampling(xyz.to(DEV), 60).cpu(),
                       oracle.furthest_point_sampling(xyz, 60))


def test_fps_large_scenes_keep_the_reference_tie_order(hip_ext, oracle, fps_path):
    """24 577-40 960 points: waves that own a compact block of the scene's Morton order and sit rounds out (the
    cooperative kernel: a thread keeps its points sorted by tie key; the block-list kernel: the key travels with every
    row).  Exact ties (duplicated points, a grid) and the zero tail resolve as in the reference."""
    _fps_case(hip_ext, oracle, 2, 40000, 600, seed=21, dup=4000, zero_tail=900)
    _fps_case(hip_ext, oracle, 1, 24577, 300, seed=22, dup=2000, zero_tail=100)
    g = torch.stack(torch.meshgrid(torch.arange(40.), torch.arange(40.), torch.arange(20.), indexing="ij"), -1)
    xyz = (g.reshape(1, -1, 3) * 0.25 + 0.5).contiguous()          # 32 000 grid points: massive exact ties
    ref = oracle.furthest_point_sampling(xyz, 400)
    assert torch.equal(hip_ext.furthest_point_sampling(xyz.to(DEV), 400).cpu(), ref)
    # every point inside the skip rule (index 0 for every round), and five points that take part among 9000 that do not
    none = torch.zeros(2, 9000, 3)
    assert torch.equal(hip_ext.furthest_point_sampling(none.to(DEV), 8).cpu(), oracle.furthest_point_sampling(none, 8))
    few = torch.zeros(1, 9000, 3)
    few[0, [7, 4000, 4001, 8999, 123]] = torch.tensor([[1.0, 2, 3], [4, 1, 0.5], [4, 1, 0.5], [0.1, 0.1, 0.1], [7, 7, 2]])
    assert torch.equal(hip_ext.furthest_point_sampling(few.to(DEV), 20).cpu(), oracle.furthest_point_sampling(few, 20))
    # NaN / infinite coordinates: never nearer than anything (min and the strict compare ignore them as the reference's do)
    odd = scene(1, 12000, seed=31)
    odd[0, 100] = float("nan"); odd[0, 5000, 1] = float("inf"); odd[0, 11999, 2] = float("-inf")
    assert torch.equal(hip_ext.furthest_point_sampling(odd.to(DEV), 200).cpu(), oracle.furthest_point_sampling(odd, 200))


def test_fps_blocked_degenerate_boxes(hip_ext, oracle, fps_path):
    """The blocked kernel's Morton order and its sit-out test on boxes that break a grid: a flat scene (one extent 0),
    all points in one spot plus a far outlier, two tight clusters (most waves sit out from round 3), and a scene whose
    first rounds tie everywhere."""
    g = torch.Generator().manual_seed(5)
    flat = torch.rand(1, 30000, 3, generator=g) * 6 + 0.5
    flat[..., 2] = 1.25
    spot = torch.full((1, 26000, 3), 2.0)
    spot[0, 17] = torch.tensor([9.0, 2.0, 2.0])
    spot[0, 5000:5100] += torch.rand(100, 3, generator=g) * 1e-3
    two = torch.cat([torch.rand(1, 20000, 3, generator=g) * 0.2 + 1.0, torch.rand(1, 20000, 3, generator=g) * 0.2 + 7.0], 1)
    two = two[:, torch.randperm(40000, generator=g)]
    line = torch.zeros(1, 32768, 3)
    line[0, :, 0] = (torch.arange(32768) % 4096) * 0.01 + 1.0           # 8 copies of each of 4096 collinear points
    for cloud, m in ((flat, 500), (spot, 300), (two, 700), (line, 300)):
        cloud = cloud.contiguous()
        ref = oracle.furthest_point_sampling(cloud, m)
        assert torch.equal(hip_ext.furthest_point_sampling(cloud.to(DEV), m).cpu(), ref)


def test_fps_scene_40k(hip_ext, oracle, fps_path):
    # BASELINE shape (one scene, 40k points, SA1 npoint=2048) incl. the overflow-tail path
    _fps_case(hip_ext, oracle, 1, 40000, 2048, seed=11, dup=500, zero_tail=100)


def _nested(points, m, **kw):
    # an addition of this library, not one of the nine reference bindings the `pointnet2._ext` shim exports
    from situation3d_amd.pointnet2 import _ext as amd_ext
    return amd_ext.furthest_point_sampling_nested(points, m, **kw)


def _fps_ordered(hip_ext, xyz, m):
    idx = hip_ext.furthest_point_sampling(xyz.to(DEV), m).long()
    return torch.gather(xyz.to(DEV), 1, idx[..., None].expand(-1, -1, 3)).contiguous()


@pytest.mark.parametrize("n0,n,m", [(40000, 2048, 1024), (5000, 1024, 512), (3000, 512, 256), (900, 300, 300),
                                    (20000, 4096, 1000), (20000, 8192, 700), (600, 257, 100)])
def test_nested_fps_proves_the_prefix_on_fps_ordered_clouds(hip_ext, oracle, n0, n, m):
    """SA level l+1 samples the FPS-ordered centres of level l: the nested entry point must return what
    the plain one returns (== the oracle), and on tie-free scenes it must have PROVEN it (flag 1, no
    dependent rounds)."""
    cloud = _fps_ordered(hip_ext, scene(3, n0, seed=n0 + m), n)
    got, proven = _nested(cloud, m, return_proven=True)
    assert torch.equal(got.cpu(), oracle.furthest_point_sampling(cloud.cpu(), m))
    assert torch.equal(got, hip_ext.furthest_point_sampling(cloud, m))
    assert proven.tolist() == [1, 1, 1]
    assert torch.equal(got.cpu(), torch.arange(m, dtype=torch.int32).expand(3, m))


def test_nested_fps_falls_back_per_scene_on_ties_skips_and_unordered_input(hip_ext, oracle):
    # scene 0: FPS-ordered (provable); scene 1: random order; scene 2: FPS-ordered with duplicated points
    # and a zero tail (ties + skipped points inside the prefix) -> those two run the ordinary rounds
    a = _fps_ordered(hip_ext, scene(1, 6000, seed=1), 2048)
    b = scene(1, 2048, seed=2).to(DEV)
    c = _fps_ordered(hip_ext, scene(1, 6000, seed=3, dup=2500, zero_tail=800), 2048)
    cloud = torch.cat([a, b, c]).contiguous()
    got, proven = _nested(cloud, 1024, return_proven=True)
    ref = oracle.furthest_point_sampling(cloud.cpu(), 1024)
    assert torch.equal(got.cpu(), ref)
    assert proven[0].item() == 1 and proven[1].item() == 0
    assert (proven[2].item() == 1) == torch.equal(ref[2], torch.arange(1024, dtype=torch.int32))
    # lattice: every round ties; all-zero cloud: every point skipped; m > n and n > 8192: plain kernel
    g = torch.stack(torch.meshgrid(torch.arange(16.), torch.arange(16.), torch.arange(8.), indexing="ij"), -1)
    lattice = (g.reshape(1, -1, 3) * 0.25 + 0.5).contiguous()
    for cloud, m in ((lattice, 512), (_fps_ordered(hip_ext, lattice, 1024).cpu(), 512), (torch.zeros(2, 300, 3), 8),
                     (scene(1, 40, seed=3), 60), (scene(2, 9000, seed=4), 100), (scene(1, 1, seed=5), 1)):
        got = _nested(cloud.to(DEV), m).cpu()
        assert torch.equal(got, oracle.furthest_point_sampling(cloud, m)), (tuple(cloud.shape), m)


@pytest.mark.parametrize("n0,ms", [(2048, [1024, 512, 256]), (600, [300, 128]), (700, [700]), (513, [512, 255, 100, 9])])
def test_nested_fps_chain_equals_level_by_level(hip_ext, oracle, n0, ms):
    """sig3d_fps_nested_chain (one proof for all levels) against sig3d_furthest_point_sampling_nested + gather, level
    by level, and the oracle: FPS-ordered scenes (every level proven), a random order (nothing proven), an ordered
    scene with duplicates / a zero tail inside the prefix, and one swapped pair late in the order (level 0 fails, the
    levels below must then run on their TRUE input)."""
    from situation3d_amd.pointnet2 import _ext as amd_ext
    a = _fps_ordered(hip_ext, scene(1, 3 * n0, seed=n0), n0)
    b_ = scene(1, n0, seed=n0 + 1).to(DEV)
    c = _fps_ordered(hip_ext, scene(1, 3 * n0, seed=n0 + 2, dup=n0, zero_tail=n0 // 3), n0)
    d = a.clone()
    i, j = n0 - 3, n0 - 2
    d[0, [i, j]] = d[0, [j, i]]
    cloud = torch.cat([a, b_, c, d]).contiguous()
    idxs, cent, proven = amd_ext.furthest_point_sampling_nested_chain(cloud, ms)
    cur = cloud
    for l, m in enumerate(ms):
        ref = oracle.furthest_point_sampling(cur.cpu(), m)
        assert torch.equal(idxs[l].cpu(), ref), (l, (idxs[l].cpu() != ref).nonzero()[:3].tolist())
        assert torch.equal(idxs[l], _nested(cur, m))
        nxt = torch.gather(cur, 1, ref.to(DEV).long()[..., None].expand(-1, -1, 3)).contiguous()
        assert torch.equal(cent[l], nxt)
        for sc in range(4):
            if proven[l, sc].item():
                assert torch.equal(idxs[l][sc].cpu(), torch.arange(m, dtype=torch.int32))
                assert all(proven[k, sc].item() for k in range(l))       # a proof holds only below proven levels
        cur = nxt
    assert proven[:, 0].tolist() == [1] * len(ms) and proven[:, 1].sum().item() == 0
    if ms[0] > n0 - 3:
        assert proven[0, 3].item() == 0


def test_nested_fps_rejects_a_prefix_that_is_one_swap_away(hip_ext, oracle):
    """Adversarial: swap two late points of an FPS-ordered cloud -- a single round is wrong, every other
    condition holds -- and perturb one point by one ulp-scale step so it wins its round early."""
    base = _fps_ordered(hip_ext, scene(1, 9000, seed=8), 1024)
    for i, j in ((1, 2), (500, 501), (1022, 1023), (3, 900)):
        cloud = base.clone()
        cloud[0, [i, j]] = cloud[0, [j, i]]
        got, proven = _nested(cloud, 1024, return_proven=True)
        assert torch.equal(got.cpu(), oracle.furthest_point_sampling(cloud.cpu(), 1024)), (i, j)
        assert proven.item() == 0
    cloud = base.clone()
    got, proven = _nested(cloud[:, :700].contiguous(), 512, return_proven=True)
    assert proven.item() == 1 and torch.equal(got.cpu()[0], torch.arange(512, dtype=torch.int32))


@pytest.mark.parametrize("n,m,r,ns", [(9, 2, 5.0, 6), (9, 2, 10.0, 3), (4096, 512, 0.2, 64),
                                      (4096, 512, 0.8, 16), (1000, 77, 0.5, 7), (100, 100, 0.3, 1),
                                      (63, 5, 1.0, 130), (20000, 256, 0.4, 32)])
def test_ball_query(hip_ext, oracle, n, m, r, ns):
    xyz = scene(2, n, seed=n + ns, dup=n // 8)
    new_xyz = xyz[:, torch.randperm(n, generator=torch.Generator().manual_seed(1))[:m]].contiguous()
    new_xyz[:, -1] = 100.0  # a centre with no neighbour: row must stay all-zero
    ref = oracle.ball_query(new_xyz, xyz, r, ns)
    got = hip_ext.ball_query(new_xyz.to(DEV), xyz.to(DEV), r, ns).cpu()
    assert torch.equal(got, ref)
    assert (got[:, -1] == 0).all()


@pytest.mark.parametrize("n,m,r,ns,dup,zero_tail", [(40000, 2048, 0.2, 64, 0, 0), (40000, 2048, 0.2, 64, 5000, 3000),
                                                    (9000, 300, 0.4, 32, 1000, 0), (20000, 256, 1.5, 16, 0, 500),
                                                    (8192, 64, 0.05, 8, 0, 0)])
def test_ball_query_grid_path_bit_exact(hip_ext, oracle, n, m, r, ns, dup, zero_tail):
    """n >= 256 takes the cell-binned kernels (centres hashed into cells in LDS, points streamed once, hit lists
    ranked back into index order): index order, padding, all-zero rows and dense neighbourhoods (duplicates, a
    zero-padded tail with thousands of coincident points -> the ordered-scan fallback) must equal the
    reference's serial scan bit for bit."""
    from situation3d_amd.pointnet2 import _ext
    assert n >= _ext.GRID_MIN_POINTS
    xyz = scene(2, n, seed=n + ns, dup=dup)
    if zero_tail:
        xyz[:, -zero_tail:] = 0.0
    sel = torch.randperm(n, generator=torch.Generator().manual_seed(2))[:m]
    new_xyz = xyz[:, sel].contiguous()
    new_xyz[:, 0] = 0.0            # a centre inside the zero-padded cluster (if any)
    new_xyz[:, -1] = 100.0         # a centre with no neighbour: row must stay all-zero
    new_xyz[:, 1] = -3.7           # negative cell coordinates
    ref = oracle.ball_query(new_xyz, xyz, r, ns)
    got = hip_ext.ball_query(new_xyz.to(DEV), xyz.to(DEV), r, ns).cpu()
    assert torch.equal(got, ref)


def test_ball_query_levels_one_launch_pair_for_a_whole_stack(hip_ext, oracle):
    """sig3d_ball_query_levels: the four SA levels of BASELINE config 3 (each level queries the centres of the
    level above) in ONE scatter + ONE rank launch -- every list equal to the reference's serial scan."""
    from situation3d_amd.pointnet2 import _ext
    b = 2
    xyz = scene(b, 40000, seed=77, dup=3000)
    g = torch.Generator().manual_seed(3)
    chain, cur = [], xyz
    for m, r, ns in ((2048, 0.2, 64), (1024, 0.4, 32), (512, 0.8, 16), (256, 1.2, 16)):
        sel = torch.randperm(cur.shape[1], generator=g)[:m]
        nxt = cur[:, sel].contiguous()
        chain.append((nxt, cur, r, ns))
        cur = nxt
    got = _ext.ball_query_levels([(a.to(DEV), c.to(DEV), r, ns) for a, c, r, ns in chain])
    for (a, c, r, ns), idx in zip(chain, got):
        assert torch.equal(idx.cpu(), oracle.ball_query(a, c, r, ns)), (a.shape, r, ns)


@pytest.mark.parametrize("n,m,r,ns", [(2049, 5000, 0.3, 8), (300, 4097, 0.5, 4), (6000, 9000, 0.25, 16)])
def test_ball_query_more_centres_than_one_table(hip_ext, oracle, n, m, r, ns):
    """More than 4096 centres: blocks of 4096 share the launch (one LDS table each)."""
    xyz = scene(2, n, seed=n + m)
    g = torch.Generator().manual_seed(5)
    new_xyz = (xyz[:, torch.randint(0, n, (m,), generator=g)] + torch.randn(2, m, 3, generator=g) * 0.05).contiguous()
    ref = oracle.ball_query(new_xyz, xyz, r, ns)
    got = hip_ext.ball_query(new_xyz.to(DEV), xyz.to(DEV), r, ns).cpu()
    assert torch.equal(got, ref)


def test_ball_query_cells_survive_far_away_negative_and_colliding_coordinates(hip_ext, oracle):
    """Hash collisions (a scene hundreds of cells wide hashed into 2m buckets), negative coordinates, points on
    cell boundaries (multiples of the cell edge 2.02 r) and a cluster of 600 coincident points around ONE centre
    (more hits than the 256 list slots: ordered-scan fallback next to ranked neighbours in the same wave)."""
    g = torch.Generator().manual_seed(11)
    n, m, r, ns = 20000, 700, 0.25, 32
    xyz = (torch.rand(2, n, 3, generator=g) - 0.5) * torch.tensor([300.0, 300.0, 40.0])
    e = 2.02 * r
    xyz[:, :2000] = torch.round(xyz[:, :2000] / e) * e           # exactly on cell boundaries (as far as f32 goes)
    xyz[:, 5000:5600] = xyz[:, 4999:5000]                         # 600 coincident points
    sel = torch.randperm(n, generator=g)[:m]
    sel[3] = 5003
    new_xyz = xyz[:, sel].contiguous()
    new_xyz[:, 10:200] += torch.randn(2, 190, 3, generator=g) * 0.1
    ref = oracle.ball_query(new_xyz, xyz, r, ns)
    got = hip_ext.ball_query(new_xyz.to(DEV), xyz.to(DEV), r, ns).cpu()
    assert torch.equal(got, ref)
    assert int((ref[:, 3] != ref[:, 3, :1]).sum()) > 0            # the cluster's centre really has a full list


@pytest.mark.parametrize("c,n,p,s", [(3, 4096, 512, 64), (6, 9, 2, 3), (131, 2048, 256, 32),
                                     (5, 100, 7, 5), (1, 50, 1, 1)])
def test_group_points_and_grad(hip_ext, oracle, c, n, p, s):
    g = torch.Generator().manual_seed(c + n)
    pts = feats(2, c, n)
    idx = torch.randint(0, n, (2, p, s), generator=g, dtype=torch.int32)
    ref = oracle.group_points(pts, idx)
    got = hip_ext.group_points(pts.to(DEV), idx.to(DEV)).cpu()
    assert torch.equal(got, ref)
    go = torch.rand(2, c, p, s, generator=g)
    refg = oracle.group_points_grad(go, idx, n)
    gotg = hip_ext.group_points_grad(go.to(DEV), idx.to(DEV), n).cpu()
    torch.testing.assert_close(gotg, refg, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("c,n,m", [(3, 4096, 512), (6, 9, 2), (256, 2048, 1024)])
def test_gather_points_and_grad(hip_ext, oracle, c, n, m):
    g = torch.Generator().manual_seed(c + m)
    pts = feats(2, c, n)
    idx = torch.randint(0, n, (2, m), generator=g, dtype=torch.int32)
    assert torch.equal(hip_ext.gather_points(pts.to(DEV), idx.to(DEV)).cpu(), oracle.gather_points(pts, idx))
    go = torch.rand(2, c, m, generator=g)
    torch.testing.assert_close(hip_ext.gather_points_grad(go.to(DEV), idx.to(DEV), n).cpu(),
                               oracle.gather_points_grad(go, idx, n), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("n,m", [(512, 256), (1024, 512), (100, 2), (7, 1), (3000, 1500), (50, 3)])
def test_three_nn(hip_ext, oracle, n, m):
    unknown = scene(2, n, seed=n)
    known = scene(2, m, seed=m + 1, dup=m // 4)
    rd, ri = oracle.three_nn(unknown, known)
    gd, gi = hip_ext.three_nn(unknown.to(DEV), known.to(DEV))
    assert torch.equal(gi.cpu(), ri)
    assert torch.equal(gd.cpu(), rd)  # includes +inf for unfilled slots when m < 3


def test_three_interpolate_and_grad(hip_ext, oracle):
    g = torch.Generator().manual_seed(0)
    b, c, m, n = 2, 256, 256, 512
    pts = feats(b, c, m)
    idx = torch.randint(0, m, (b, n, 3), generator=g, dtype=torch.int32)
    w = torch.rand(b, n, 3, generator=g)
    w = (w / w.sum(-1, keepdim=True)).contiguous()
    assert torch.equal(hip_ext.three_interpolate(pts.to(DEV), idx.to(DEV), w.to(DEV)).cpu(),
                       oracle.three_interpolate(pts, idx, w))
    go = torch.rand(b, c, n, generator=g)
    torch.testing.assert_close(
        hip_ext.three_interpolate_grad(go.to(DEV), idx.to(DEV), w.to(DEV), m).cpu(),
        oracle.three_interpolate_grad(go, idx, w, m), rtol=1e-5, atol=1e-5)


def test_reference_error_behaviour(hip_ext):
    # "CPU not supported" (ball_query.cpp:27-29) and the CHECK_* macros (utils.h:5-25)
    xyz = scene(1, 16)
    with pytest.raises(RuntimeError, match="CPU not supported"):
        hip_ext.ball_query(xyz, xyz, 0.5, 4)
    with pytest.raises(RuntimeError, match="contiguous"):
        hip_ext.furthest_point_sampling(scene(1, 16).to(DEV).transpose(1, 2), 4)
    with pytest.raises(RuntimeError, match="int tensor"):
        hip_ext.gather_points(feats(1, 3, 16).to(DEV), torch.zeros(1, 4, dtype=torch.int64, device=DEV))
    with pytest.raises(RuntimeError, match="float tensor"):
        hip_ext.three_nn(xyz.double().to(DEV), xyz.to(DEV))


@pytest.mark.parametrize("b,n,m,ns,c,use_xyz,norm", [(2, 2048, 1024, 32, 128, 1, 1), (2, 1024, 512, 16, 256, 1, 0),
                                                     (1, 300, 37, 5, 36, 1, 1), (2, 512, 100, 16, 260, 0, 0),
                                                     (1, 64, 3, 7, 4, 1, 1)])
def test_query_group_fused_point_major_is_bit_identical(b, n, m, ns, c, use_xyz, norm):
    """sig3d_query_group_fused_pm (+ sig3d_transpose_cn) against sig3d_query_group_fused: gathers and the
    individually rounded centre subtraction / division are exact, so every bit must agree -- ragged tiles
    (m*ns % 64 != 0), channel counts that are not multiples of 128, no-xyz mode."""
    import ctypes
    from situation3d_amd import _lib as L
    g = torch.Generator().manual_seed(b * 1000 + n + c)
    xyz = torch.rand(b, n, 3, generator=g).to(DEV)
    new_xyz = xyz[:, :m].contiguous()
    feat = torch.randn(b, c, n, generator=g).to(DEV)
    idx = torch.randint(0, n, (b, m, ns), generator=g, dtype=torch.int32).to(DEV)
    ct = (3 if use_xyz else 0) + c
    ref = torch.full((b, ct, m, ns), float("nan"), device=DEV)
    out = torch.full((b, ct, m, ns), float("nan"), device=DEV)
    L.call("sig3d_query_group_fused", b, n, m, c, ns, use_xyz, norm, ctypes.c_float(0.37), L.ptr(xyz), L.ptr(new_xyz),
           L.ptr(feat), L.ptr(idx), L.ptr(ref), L.stream_ptr())
    pm = torch.empty(b, n, c, device=DEV)
    L.call("sig3d_transpose_cn", b, c, n, L.ptr(feat), L.ptr(pm), L.stream_ptr())
    assert torch.equal(pm, feat.transpose(1, 2).contiguous())
    L.call("sig3d_query_group_fused_pm", b, n, m, c, c, ns, use_xyz, norm, ctypes.c_float(0.37), L.ptr(xyz),
           L.ptr(new_xyz), L.ptr(pm), L.ptr(idx), L.ptr(out), L.stream_ptr())
    assert not torch.isnan(ref).any() and torch.equal(out, ref)


@pytest.mark.parametrize("b", [8, 2, 3, 1])
def test_query_group_levels_in_one_launch_equals_the_level_by_level_launches(b):
    """sig3d_query_group_levels: the grouping of a whole set-abstraction stack as one grid (largest level first, every
    level's range starting at a multiple of 8 workgroups) against sig3d_query_group_fused / _pm level by level: every
    bit, at batch sizes that take each branch of the XCD-local scene dealing; ragged tiles and a no-feature level."""
    import ctypes
    from situation3d_amd import _lib as L
    g = torch.Generator().manual_seed(77 + b)
    #        n     m    ns   c   point-major
    shape = [(3000, 500, 16, 3, False), (500, 260, 32, 128, True), (260, 100, 12, 36, True), (100, 37, 8, 260, True)]
    probs, refs, outs = [], [], []
    for n, m, ns, c, pm in shape:
        xyz = torch.rand(b, n, 3, generator=g).to(DEV)
        new_xyz = xyz[:, :m].contiguous()
        feat = torch.randn(b, c, n, generator=g).to(DEV)
        idx = torch.randint(0, n, (b, m, ns), generator=g, dtype=torch.int32).to(DEV)
        ref = torch.full((b, 3 + c, m, ns), float("nan"), device=DEV)
        L.call("sig3d_query_group_fused", b, n, m, c, ns, 1, 1, ctypes.c_float(0.4), L.ptr(xyz), L.ptr(new_xyz),
               L.ptr(feat), L.ptr(idx), L.ptr(ref), L.stream_ptr())
        out = torch.full((b, 3 + c, m, ns), float("nan"), device=DEV)
        src = feat.transpose(1, 2).contiguous() if pm else feat
        probs.append((xyz, new_xyz, 0.4, idx, src, pm, out))
        refs.append(ref); outs.append(out)
    arr = L.group_levels(probs)
    L.call("sig3d_query_group_levels", b, len(arr), arr, L.stream_ptr())
    for ref, out in zip(refs, outs):
        assert not torch.isnan(ref).any() and torch.equal(out, ref)
    sub = L.group_levels(probs[1:3])           # fewer levels, another order of sizes
    for o in outs:
        o.fill_(float("nan"))
    L.call("sig3d_query_group_levels", b, len(sub), sub, L.stream_ptr())
    assert torch.equal(outs[1], refs[1]) and torch.equal(outs[2], refs[2]) and torch.isnan(outs[0]).all()


@pytest.mark.parametrize("b,n,m,ns,c,c_off", [(2, 2048, 1024, 32, 128, 3), (2, 1024, 512, 16, 256, 3),
                                              (1, 300, 37, 5, 36, 0), (2, 512, 100, 16, 260, 3), (1, 64, 3, 7, 4, 3)])
@pytest.mark.parametrize("padded", [False, True])
def test_query_group_fused_grad_point_major_vs_channel_major(b, n, m, ns, c, c_off, padded):
    """Scatter-add into the point-major gradient (+ transpose back) against the channel-major kernel and
    against an exact float64 index_add; float atomics: 1e-5 relative to the largest magnitude.  `padded`
    lists repeat their first index like ball query's padding (the run-merging path)."""
    import ctypes
    from situation3d_amd import _lib as L
    g = torch.Generator().manual_seed(b * 1000 + n + c + int(padded))
    idx = torch.randint(0, n, (b, m, ns), generator=g, dtype=torch.int32)
    if padded:
        keep = torch.randint(1, ns + 1, (b, m, 1), generator=g)
        idx = torch.where(torch.arange(ns).view(1, 1, ns) < keep, idx, idx[:, :, :1])
    idx = idx.to(DEV)
    ct = c_off + c
    go = torch.randn(b, ct, m, ns, generator=g).to(DEV)
    ref = torch.empty(b, c, n, device=DEV)
    L.call("sig3d_query_group_fused_grad", b, n, m, c, ns, ct, c_off, L.ptr(go), L.ptr(idx), L.ptr(ref), L.stream_ptr())
    pm = torch.full((b, n, c), float("nan"), device=DEV)
    L.call("sig3d_query_group_fused_grad_pm", b, n, m, c, c, ns, ct, c_off, L.ptr(go), L.ptr(idx), L.ptr(pm),
           L.stream_ptr())
    back = torch.empty(b, c, n, device=DEV)
    L.call("sig3d_transpose_cn", b, n, c, L.ptr(pm), L.ptr(back), L.stream_ptr())
    exact = torch.zeros(b, c, n, dtype=torch.float64, device=DEV)
    exact.scatter_add_(2, idx.long().view(b, 1, m * ns).expand(b, c, m * ns),
                       go[:, c_off:].reshape(b, c, m * ns).double())
    scale = exact.abs().max().item()
    assert (back.double() - exact).abs().max().item() <= 1e-5 * scale
    assert (ref.double() - exact).abs().max().item() <= 1e-5 * scale
