"""Independent evidence for the C restatement of the native ops (the reference CUDA cannot run anywhere in
this project): the oracle against
  * a thread-level NumPy simulator of the FPS kernel written from the .cu (tests/fps_thread_sim.py) on
    tie-heavy scenes at every block size the launcher can pick (4 ... 512) and n up to 40 000;
  * exhaustive vectorised re-derivations of ball_query and three_nn over EVERY row at N = 40 000, in the
    same individually rounded float32 arithmetic (exact index equality, not a tolerance).
"""
import numpy as np
import pytest
import torch

import fps_thread_sim
from util import scene


def _tie_heavy(n, seed, lattice, dup, zero_tail):
    """Points on a coarse lattice (many exactly equal distances), duplicated points, an all-zero tail."""
    g = torch.Generator().manual_seed(seed)
    xyz = torch.randint(0, lattice, (n, 3), generator=g).float() * 0.25 + 0.5
    if dup:
        src = torch.randint(0, n, (dup,), generator=g)
        dst = torch.randint(0, n, (dup,), generator=g)
        xyz[dst] = xyz[src]
    if zero_tail:
        xyz[n - zero_tail:] = 0.0
    return xyz.contiguous()


# n -> block size by cuda_utils.h:13-19: 5->4, 9->8, 20->16, 33->32, 70->64, 130->128, 300->256, then 512
@pytest.mark.parametrize("n,m,lattice", [(5, 5, 2), (9, 9, 2), (20, 16, 3), (33, 20, 3), (70, 40, 3), (130, 64, 4),
                                         (300, 64, 4), (700, 64, 5), (5000, 64, 8), (40000, 56, 12)])
def test_fps_oracle_equals_thread_level_simulation_on_ties(oracle, n, m, lattice):
    rounds_with_ties = 0
    for seed in range(3 if n >= 5000 else 6):
        xyz = _tie_heavy(n, 100 * n + seed, lattice, dup=n // 10, zero_tail=n // 20)
        got = oracle.furthest_point_sampling(xyz[None].contiguous(), m)[0].numpy()
        ref = fps_thread_sim.furthest_point_sampling(xyz.numpy(), m)
        assert (got == ref).all(), (n, seed, np.nonzero(got != ref)[0][:5])
        # count the rounds whose winner was decided by the tie rule: another point had the same distance
        sel = xyz[torch.from_numpy(ref).long()]
        d = torch.cdist(xyz.double(), sel.double())
        for j in range(1, m):
            dmin = d[:, :j].min(1).values
            dmin[(xyz * xyz).sum(1) <= 1e-3] = -1
            if (dmin == dmin.max()).sum() > 1:
                rounds_with_ties += 1
    assert rounds_with_ties >= (50 if n >= 700 else 1), rounds_with_ties


def test_fps_block_size_rule(oracle):
    for w in list(range(1, 70)) + [127, 128, 255, 256, 511, 512, 513, 1023, 40000]:
        assert oracle.opt_n_threads(w) == fps_thread_sim.opt_n_threads(w)


def test_fps_oracle_equals_simulation_on_random_scene(oracle):
    xyz = scene(1, 40000, seed=77, dup=500, zero_tail=300)
    got = oracle.furthest_point_sampling(xyz, 128)[0].numpy()
    assert (got == fps_thread_sim.furthest_point_sampling(xyz[0].numpy(), 128)).all()


def _d2_f32(a, b):
    """(m,3) x (n,3) -> (m,n) squared distances, every operation a separate float32 rounding, summed
    left to right as ball_query_gpu.cu:31-32 / interpolate_gpu.cu:33 write them."""
    dx = a[:, None, 0] - b[None, :, 0]
    dy = a[:, None, 1] - b[None, :, 1]
    dz = a[:, None, 2] - b[None, :, 2]
    return (dx * dx + dy * dy) + dz * dz


def test_ball_query_every_row_at_40000_points(oracle):
    """ball_query_gpu.cu:9-44 at the SA1 shape: first `nsample` indices with d2 < r*r in index order, the row
    padded with its first hit, all-zero when there is none -- all 2 x 2048 rows, exact."""
    n, m, r, ns = 40000, 2048, 0.2, 64
    xyz = scene(2, n, seed=31, dup=400, zero_tail=100)
    inds = torch.stack([torch.randperm(n, generator=torch.Generator().manual_seed(5 + i))[:m] for i in range(2)])
    new_xyz = torch.stack([xyz[i, inds[i]] for i in range(2)])
    new_xyz[:, -3:] += 50.0                                  # three centres with no neighbour at all
    idx = oracle.ball_query(new_xyz, xyz, r, ns)
    r2 = torch.tensor(r, dtype=torch.float32) * torch.tensor(r, dtype=torch.float32)
    for b in range(2):
        for lo in range(0, m, 256):
            inside = _d2_f32(new_xyz[b, lo:lo + 256], xyz[b]) < r2          # (256, n)
            rank = inside.cumsum(1)
            cnt = rank[:, -1].clamp(max=ns)
            exp = torch.zeros(256, ns, dtype=torch.int32)
            rows, cols = (inside & (rank <= ns)).nonzero(as_tuple=True)
            exp[rows, (rank[rows, cols] - 1)] = cols.int()
            first = exp[:, :1].expand(-1, ns)
            fill = torch.arange(ns)[None, :] >= cnt[:, None]
            exp = torch.where(fill & (cnt[:, None] > 0), first, exp)
            assert torch.equal(idx[b, lo:lo + 256], exp), (b, lo)
    assert (idx[:, -3:] == 0).all()


def test_three_nn_every_row_at_40000_points(oracle):
    """interpolate_gpu.cu:9-59 with 40 000 unknown points against 2048 known ones: the three smallest
    float32 distances in ascending order, lowest index first among equals (strict `<`), every row."""
    n, m = 40000, 2048
    unknown = scene(1, n, seed=41, dup=300)
    known = unknown[:, torch.randperm(n, generator=torch.Generator().manual_seed(9))[:m]].contiguous()
    known[0, 100:140] = known[0, 50:90]                      # duplicated known points: exact ties
    d2, i3 = oracle.three_nn(unknown, known)
    for lo in range(0, n, 4096):
        d = _d2_f32(unknown[0, lo:lo + 4096], known[0]).clone()
        for t in range(3):
            val, arg = d.min(1)                              # first minimum = lowest index among equals
            assert torch.equal(i3[0, lo:lo + 4096, t].long(), arg), (lo, t)
            assert torch.equal(d2[0, lo:lo + 4096, t], val)
            d[torch.arange(d.shape[0]), arg] = float("inf")


def test_three_nn_fewer_than_three_known_points(oracle):
    """m < 3: unfilled slots keep the double 1e40 initial value -> float inf, index 0 (interpolate_gpu.cu:27)."""
    unknown, known = scene(1, 10, seed=1), scene(1, 2, seed=2)
    d2, i3 = oracle.three_nn(unknown, known)
    assert torch.isinf(d2[..., 2]).all() and (i3[..., 2] == 0).all()
    assert torch.isfinite(d2[..., :2]).all()


@pytest.mark.parametrize("n,m,dup,zero_tail", [(9000, 160, 0, 0), (10000, 140, 900, 300), (8500, 120, 4000, 0)])
def test_block_list_fps_algorithm_equals_the_reference_sampling(oracle, n, m, dup, zero_tail):
    """tests/fps_blocks_sim.py -- the ALGORITHM of the block-list kernel (csrc/sampling.hip: fps_blocks_kernel) in
    NumPy float32: Morton blocks of 64 rows, a block sits a round out when the sample is farther from its box (x 0.99998)
    than its largest running distance.  Same indices as the oracle (= the reference's sampling) on random scenes,
    scenes with thousands of duplicated points (exact ties decided by the key) and a zero tail (skipped points), and
    whichever order the points of a Morton cell arrive in; most block-rounds are sat out."""
    import numpy as np
    import fps_blocks_sim
    from util import scene
    xyz = scene(1, n, seed=n + m, dup=dup, zero_tail=zero_tail)
    ref = oracle.furthest_point_sampling(xyz, m)[0].numpy()
    stats = {}
    got = fps_blocks_sim.furthest_point_sampling(xyz[0].numpy(), m, stats=stats)
    assert np.array_equal(got, ref), np.nonzero(got != ref)[0][:5]
    again = fps_blocks_sim.furthest_point_sampling(xyz[0].numpy(), m, rng=np.random.default_rng(3))
    assert np.array_equal(again, ref)
    swept = np.array(stats["swept"])
    assert swept[0] >= stats["blocks"] - 8 and swept[len(swept) // 2:].mean() < 0.35 * stats["blocks"]


def test_block_list_fps_algorithm_on_degenerate_scenes(oracle):
    """A flat scene (one box extent 0), a lattice (every round ties), everything inside the skip rule, NaN / infinite
    coordinates: the sit-out rule and the candidate order of the block-list algorithm against the oracle."""
    import numpy as np
    import torch
    import fps_blocks_sim
    g = torch.Generator().manual_seed(5)
    flat = torch.rand(1, 9000, 3, generator=g) * 6 + 0.5
    flat[..., 2] = 1.25
    lat = torch.stack(torch.meshgrid(torch.arange(24.), torch.arange(24.), torch.arange(16.), indexing="ij"), -1)
    lat = (lat.reshape(1, -1, 3) * 0.25 + 0.5).contiguous()
    none = torch.zeros(1, 8300, 3)
    odd = torch.rand(1, 9000, 3, generator=g) * torch.tensor([8.0, 8.0, 3.0])
    odd[0, 100] = float("nan"); odd[0, 5000, 1] = float("inf"); odd[0, 8999, 2] = float("-inf")
    for cloud, m in ((flat, 120), (lat, 150), (none, 8), (odd, 100)):
        ref = oracle.furthest_point_sampling(cloud.contiguous(), m)[0].numpy()
        got = fps_blocks_sim.furthest_point_sampling(cloud[0].numpy(), m)
        assert np.array_equal(got, ref), (tuple(cloud.shape), np.nonzero(got != ref)[0][:5])
