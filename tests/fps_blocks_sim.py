"""A NumPy simulator of the block-list furthest-point sampling of csrc/sampling.hip (fps_blocks_sort_kernel +
fps_blocks_kernel), written from the kernel's ALGORITHM: Morton-ordered rows, blocks of 64 rows with a bounding box, the
largest running distance of a block, its tie key and its point; a round sweeps only the blocks the new sample can reach.
Every float operation is a separate float32 operation in the kernel's order (the library is built with
-ffp-contract=off and spells distances with _rn intrinsics), so the simulator decides on the CPU what the GPU tests
decide on the device: that sitting a block out (box distance x 0.99998 > the block's largest running distance) never
changes an index of the reference's sampling (lib/pointnet2/_ext_src/src/sampling_gpu.cu:69-173), ties, skipped points
and degenerate boxes included.  Test infrastructure only.
"""
import numpy as np

F = np.float32
NOKEY = np.uint32(0xFFFFFFFF)


def _bitrev(v, bits):
    out = np.zeros_like(v)
    for i in range(bits):
        out |= ((v >> i) & 1) << (bits - 1 - i)
    return out


def tie_keys(n):
    """fps_key of sampling.hip: (bitrev_L(k mod 2^L) << 22) | (k >> L), L = log2 of the reference's block size."""
    L = min(int(np.log(float(n)) / np.log(2.0)), 9)
    k = np.arange(n, dtype=np.uint32)
    hi = _bitrev(k & np.uint32((1 << L) - 1), L) if L else np.zeros(n, np.uint32)
    return ((hi << np.uint32(22)) | (k >> np.uint32(L))).astype(np.uint32), L


def unkey(key, L):
    hi, lo = int(key) >> 22, int(key) & 0x3FFFFF
    low = int(_bitrev(np.array([hi], np.uint32), L)[0]) if L else 0
    return (lo << L) | low


def morton_order(xyz):
    """The 32 x 32 x 8 grid of fps_blocks_sort_kernel (which point of a cell comes first is free: the result may not
    depend on it -- `rng` callers shuffle inside cells to check that)."""
    lo, hi = xyz.min(0), xyz.max(0)
    with np.errstate(divide="ignore", invalid="ignore"):
        s = np.array([32.0, 32.0, 8.0], F) / (hi - lo)
    lim = np.array([31.0, 31.0, 7.0], F)
    with np.errstate(invalid="ignore"):
        c = np.minimum(np.maximum((xyz - lo) * s, F(0)), lim)      # fmaxf(NaN, 0) = 0
    c = np.nan_to_num(c, nan=0.0).astype(np.uint32)

    def spread3(v):
        v = (v | (v << 8)) & np.uint32(0x0300F00F)
        v = (v | (v << 4)) & np.uint32(0x030C30C3)
        v = (v | (v << 2)) & np.uint32(0x09249249)
        return v
    ix, iy, iz = c[:, 0], c[:, 1], c[:, 2]
    hx, hy = ix >> 3, iy >> 3
    top = (hx & 1) | ((hy & 1) << 1) | ((hx & 2) << 1) | ((hy & 2) << 2)
    return spread3(ix & 7) | (spread3(iy & 7) << 1) | (spread3(iz) << 2) | (top << 9)


def furthest_point_sampling(xyz, m, block=64, rng=None, stats=None):
    """xyz (n, 3) float32 -> (m,) int32, by the block-list algorithm.  stats (a dict): swept blocks per round."""
    xyz = np.ascontiguousarray(xyz, dtype=F)
    n = xyz.shape[0]
    keys, L = tie_keys(n)
    cell = morton_order(xyz)
    order = np.argsort(cell, kind="stable")
    if rng is not None:                                   # another order inside the cells: same result expected
        jitter = rng.random(n)
        order = np.lexsort((jitter, cell))
    nb = (n + block - 1) // block
    npad = nb * block
    P = np.zeros((npad, 3), F)
    K = np.full(npad, NOKEY, np.uint32)
    P[:n], K[:n] = xyz[order], keys[order]
    x, y, z = P[:, 0], P[:, 1], P[:, 2]
    mag = (x * x + y * y) + z * z                         # three products, two sums, each rounded (sampling_gpu.cu:100)
    T = np.where((K == NOKEY) | (mag.astype(np.float64) <= 1e-3), F(-1), F(1e10)).astype(F)
    Pb, Kb = P.reshape(nb, block, 3), K.reshape(nb, block)
    Tb = T.reshape(nb, block)
    live = Tb >= 0
    big = F(3.0e38)
    with np.errstate(invalid="ignore"):                   # boxes over the points that take part; fminf / fmaxf drop NaN
        lo = np.fmin(np.where(live[..., None], Pb, big), big).min(1)
        hi = np.fmax(np.where(live[..., None], Pb, -big), -big).max(1)

    def block_candidate(b):                               # largest running distance, lowest key among its holders
        v = Tb[b].max()
        hold = np.nonzero(Tb[b] == v)[0]
        i = hold[np.argmin(Kb[b][hold])]
        return v, Kb[b][i], Pb[b][i]

    bval = np.empty(nb, F)
    bkey = np.empty(nb, np.uint32)
    bpt = np.empty((nb, 3), F)
    for b in range(nb):
        bval[b], bkey[b], bpt[b] = block_candidate(b)
    idxs = np.zeros(m, np.int32)
    s = xyz[0].copy()
    swept = []
    for j in range(1, m):
        with np.errstate(invalid="ignore", over="ignore"):
            e = np.fmax(np.fmax(lo - s, s - hi), F(0))    # fmaxf drops a NaN operand, as the kernel's does
            d_box = ((e[:, 0] * e[:, 0] + e[:, 1] * e[:, 1]) + e[:, 2] * e[:, 2]) * F(0.99998)
            sit_out = (bval < 0) | (d_box > bval)         # NaN compares false: the block sweeps
        todo = np.nonzero(~sit_out)[0]
        swept.append(len(todo))
        for b in todo:
            with np.errstate(invalid="ignore", over="ignore"):
                dx, dy, dz = Pb[b][:, 0] - s[0], Pb[b][:, 1] - s[1], Pb[b][:, 2] - s[2]
                d = (dx * dx + dy * dy) + dz * dz
                Tb[b] = np.fmin(d, Tb[b])                 # fminf: a NaN distance leaves the running distance alone
            bval[b], bkey[b], bpt[b] = block_candidate(b)
        gv = bval.max()
        if gv < 0:                                        # nothing takes part: index 0 for every remaining round
            break
        hold = np.nonzero(bval == gv)[0]
        w = hold[np.argmin(bkey[hold])]
        idxs[j] = unkey(bkey[w], L)
        s = bpt[w].copy()
    if stats is not None:
        stats["swept"] = swept
        stats["blocks"] = nb
    return idxs
