"""An INDEPENDENT thread-level simulator of the reference's FPS kernel, written from
lib/pointnet2/_ext_src/src/sampling_gpu.cu:59-173 (and the block-size rule cuda_utils.h:13-19),
NOT from oracle/pointnet2_oracle.c: NumPy arrays over the thread index stand for the block's threads.

  * thread `tid` walks k = tid, tid + block, ... in order, keeps (best, besti) with the strict `>`
    of :108-109, skips points with x^2+y^2+z^2 <= 1e-3 (float vs double literal, :100-101);
  * `dists` / `dists_i` are the two shared arrays; the halving tree applies `__update` (:59-65,
    `v2 > v1 ? i2 : i1`) to slots (tid, tid + half) for tid < half, half = block/2 ... 1.

Every float operation is a separate NumPy float32 operation (one rounding each, no contraction),
which is the arithmetic contract stated in DESIGN.md section 2.  Test infrastructure only.
"""
import numpy as np


def opt_n_threads(work_size):
    """cuda_utils.h:13-19: 2^floor(log2(work_size)) clamped to [1, 512]."""
    p = 1
    while p * 2 <= work_size:
        p *= 2
    return max(min(p, 512), 1)


def furthest_point_sampling(xyz, m):
    """xyz (n, 3) float32 -> (m,) int32 indices of one batch element."""
    xyz = np.ascontiguousarray(xyz, dtype=np.float32)
    n = xyz.shape[0]
    block = opt_n_threads(n)
    tid = np.arange(block)
    chunks = (n + block - 1) // block
    temp = np.full(n, 1e10, dtype=np.float32)          # sampling.cpp:74-76
    x2, y2, z2 = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    mag = (x2 * x2) + (y2 * y2) + (z2 * z2)            # float32, three roundings + two
    skip = mag.astype(np.float64) <= 1e-3
    idxs = np.zeros(m, dtype=np.int32)
    old = 0
    for j in range(1, m):
        x1, y1, z1 = xyz[old]
        dx, dy, dz = x2 - x1, y2 - y1, z2 - z1
        d = (dx * dx + dy * dy) + dz * dz              # left-to-right as written at :104-105
        best = np.full(block, -1.0, dtype=np.float32)
        besti = np.zeros(block, dtype=np.int64)
        for c in range(chunks):                        # iteration c of every thread's k-loop
            k = c * block + tid
            live = k < n
            kk = np.where(live, k, 0)
            act = live & ~skip[kk]
            d2 = np.minimum(d[kk], temp[kk])
            temp[kk[act]] = d2[act]
            better = act & (d2 > best)
            besti = np.where(better, kk, besti)
            best = np.where(better, d2, best)
        dists, dists_i = best.copy(), besti.copy()
        half = block // 2
        while half >= 1:
            v1, v2 = dists[:half].copy(), dists[half:2 * half].copy()
            i1, i2 = dists_i[:half].copy(), dists_i[half:2 * half].copy()
            dists[:half] = np.maximum(v1, v2)
            dists_i[:half] = np.where(v2 > v1, i2, i1)
            half //= 2
        old = int(dists_i[0])
        idxs[j] = old
    return idxs
