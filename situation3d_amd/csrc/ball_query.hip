// ball_query.hip -- radius neighbour search for gfx950.
//
// Replaces lib/pointnet2/_ext_src/src/ball_query_gpu.cu of the reference: for every centre,
// the first `nsample` points IN INDEX ORDER with d2 < radius*radius (strict, f32), the row
// padded with the first hit, all-zero when there is none (ball_query_gpu.cu:27-42,
// ball_query.cpp:19-21).
//
// The reference gives one THREAD a centre and lets it walk all n points serially.  Here a
// WAVEFRONT walks the scene 64 points at a time (one point per lane, centre coordinates in
// SGPRs), a 64-bit ballot marks the hits of the chunk, and popcount of the lower lanes gives
// each hit its slot -- which preserves index order by construction.  Each wave serves CPW
// centres from the same 64-point register tile, so the scene is read from L2 once per CPW
// centres instead of once per centre.  The kernel writes every element of idx itself
// (padding and the no-hit zero row included): no memset pass is needed.
#include "sig3d_common.h"

namespace {

constexpr int BQ_WAVES = 4;  // waves per workgroup
constexpr int BQ_CPW = 8;    // centres per wave

template <int CPW>
__global__ __launch_bounds__(BQ_WAVES * 64) void ball_query_scan_kernel(
    int n, int m, float radius2, int nsample, const float *__restrict__ new_xyz_all,
    const float *__restrict__ xyz_all, int *__restrict__ idx_all) {
  const int lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int bi = blockIdx.y;
  const int c0 = (blockIdx.x * BQ_WAVES + wave) * CPW;  // first centre of this wave
  if (c0 >= m) return;
  const float *xyz = xyz_all + (size_t)bi * n * 3;
  const float *new_xyz = new_xyz_all + (size_t)bi * m * 3;
  int *idx = idx_all + (size_t)bi * m * nsample;

  float cx[CPW], cy[CPW], cz[CPW];
  int cnt[CPW], first[CPW];
#pragma unroll
  for (int c = 0; c < CPW; ++c) {
    const int j = min(c0 + c, m - 1);  // clamp: out-of-range slots repeat the last centre
    cx[c] = new_xyz[3 * j + 0];
    cy[c] = new_xyz[3 * j + 1];
    cz[c] = new_xyz[3 * j + 2];
    cnt[c] = (c0 + c < m) ? 0 : nsample;  // out-of-range slots start "full"
    first[c] = 0;
  }

  for (int k0 = 0; k0 < n; k0 += 64) {
    const int k = k0 + lane;
    const bool inb = k < n;
    const int kk = inb ? k : n - 1;
    const float x = xyz[3 * kk + 0], y = xyz[3 * kk + 1], z = xyz[3 * kk + 2];
    int open = 0;
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
      // ball_query_gpu.cu:31-32: (new_x - x)^2 + (new_y - y)^2 + (new_z - z)^2, unfused
      const float d2 = sq_dist3(cx[c], cy[c], cz[c], x, y, z);
      const bool hit = inb && (d2 < radius2);
      const unsigned long long mask = __ballot(hit);
      if (mask != 0ull && cnt[c] < nsample) {  // wave-uniform branch
        const int before = __builtin_popcountll(mask & ((1ull << lane) - 1ull));
        const int pos = cnt[c] + before;
        if (hit && pos < nsample) idx[(size_t)(c0 + c) * nsample + pos] = k;
        if (cnt[c] == 0) first[c] = k0 + __builtin_ctzll(mask);
        cnt[c] += __builtin_popcountll(mask);
      }
      open |= (cnt[c] < nsample);
    }
    if (!open) break;  // every centre of this wave is full (ball_query_gpu.cu:27 cnt < nsample)
  }

  // padding with the first hit (ball_query_gpu.cu:34-38) / zero row when no hit
#pragma unroll
  for (int c = 0; c < CPW; ++c) {
    if (c0 + c >= m) break;
    const int have = min(cnt[c], nsample);
    const int fill = (cnt[c] == 0) ? 0 : first[c];
    for (int l = have + lane; l < nsample; l += 64) idx[(size_t)(c0 + c) * nsample + l] = fill;
  }
}


// ---- hashed uniform grid (large scenes) ------------------------------------------------------
// Brute force costs n distance tests per centre (655 M at SA1, 0.5 ms).  With cells of edge
// c = 1.01 * radius every point with d2 < radius^2 lies in the 3x3x3 cell block around the centre's
// cell (|dx| < c  =>  cell coordinates differ by at most one; the 1 % margin absorbs the rounding of
// x * (1/c) for |x| up to ~1e5 radii).  Cells are hashed into H buckets (no scene bounds needed;
// colliding cells only add candidates, which the exact distance test removes):
//   count   : bucket histogram of the scene                         (one atomic per point)
//   scan    : exclusive prefix -> bucket starts                      (one workgroup per scene)
//   scatter : points copied into bucket order as float4 {x, y, z, index}
//   query   : one wave per centre, lane l < 27 walks neighbour bucket l; hits are compacted into an
//             LDS list (ballot + popcount) and ranked by counting smaller indices, which restores the
//             reference's INDEX ORDER exactly; a centre with more than BQG_CAP hits (dense clusters,
//             e.g. a zero-padded tail) falls back to the ordered 64-point scan of the brute-force
//             kernel.  Same strict f32 test, same padding, same all-zero row: bit-identical output.
constexpr int BQG_CAP = 256;   // hits kept per centre before falling back to the ordered scan

__device__ __forceinline__ int bqg_cell(float v, float inv_c) {
  const float f = floorf(v * inv_c);
  return (int)fminf(fmaxf(f, -2097152.f), 2097152.f);
}
__device__ __forceinline__ unsigned bqg_hash(int ix, int iy, int iz, unsigned hmask) {
  return ((unsigned)ix * 73856093u ^ (unsigned)iy * 19349663u ^ (unsigned)iz * 83492791u) & hmask;
}

__global__ __launch_bounds__(256) void bqg_count_kernel(int n, float inv_c, unsigned hmask,
                                                        const float *__restrict__ xyz_all,
                                                        int *__restrict__ counts_all) {
  const int bi = blockIdx.y, k = blockIdx.x * 256 + threadIdx.x;
  if (k >= n) return;
  const float *p = xyz_all + ((size_t)bi * n + k) * 3;
  const unsigned hsh = bqg_hash(bqg_cell(p[0], inv_c), bqg_cell(p[1], inv_c), bqg_cell(p[2], inv_c), hmask);
  atomicAdd(counts_all + (size_t)bi * (hmask + 1) + hsh, 1);
}

// counts (H) -> starts (H + 1, exclusive prefix) and cursor (H, copy of starts for the scatter), two
// coalesced passes over 2048-entry blocks (a one-workgroup-per-scene scan took 280 us at H = 131072):
//   pass 1: block-local exclusive prefix into starts[], block total into block_sums[]
//   pass 2: every block adds the sum of the totals before it and writes the cursor copy
constexpr int BQG_SCAN_BLOCK = 2048;  // entries per workgroup: 256 threads x 8

__global__ __launch_bounds__(256) void bqg_scan_local_kernel(int hsize, const int *__restrict__ counts_all,
                                                             int *__restrict__ starts_all,
                                                             int *__restrict__ block_sums_all) {
  __shared__ int s_wave[4];
  const int bi = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nblk = hsize / BQG_SCAN_BLOCK;
  const int4 *src = reinterpret_cast<const int4 *>(counts_all + (size_t)bi * hsize + (size_t)blk * BQG_SCAN_BLOCK);
  const int4 a = src[2 * tid], c = src[2 * tid + 1];
  const int v[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
  int run[8], tot = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) { run[i] = tot; tot += v[i]; }
  int incl = tot;  // inclusive scan of the thread totals across the wave
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int o = __shfl_up(incl, off);
    if (lane >= off) incl += o;
  }
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  int base = incl - tot;
  for (int w = 0; w < wave; ++w) base += s_wave[w];
  int *dst = starts_all + (size_t)bi * (hsize + 1) + (size_t)blk * BQG_SCAN_BLOCK + 8 * tid;
#pragma unroll
  for (int i = 0; i < 8; ++i) dst[i] = base + run[i];
  if (tid == 255) block_sums_all[(size_t)bi * nblk + blk] = base + tot;
}

__global__ __launch_bounds__(256) void bqg_scan_offset_kernel(int hsize, const int *__restrict__ block_sums_all,
                                                              int *__restrict__ starts_all,
                                                              int *__restrict__ cursor_all) {
  const int bi = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x;
  const int nblk = hsize / BQG_SCAN_BLOCK;
  const int *bs = block_sums_all + (size_t)bi * nblk;
  int off = 0;
  for (int i = 0; i < blk; ++i) off += bs[i];  // <= 512 uniform (scalar) loads
  int *st = starts_all + (size_t)bi * (hsize + 1) + (size_t)blk * BQG_SCAN_BLOCK + 8 * tid;
  int *cu = cursor_all + (size_t)bi * hsize + (size_t)blk * BQG_SCAN_BLOCK + 8 * tid;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int v = st[i] + off;
    st[i] = v;
    cu[i] = v;
  }
  if (blk == nblk - 1 && tid == 255) starts_all[(size_t)bi * (hsize + 1) + hsize] = off + bs[blk];
}

__global__ __launch_bounds__(256) void bqg_scatter_kernel(int n, float inv_c, unsigned hmask,
                                                          const float *__restrict__ xyz_all,
                                                          int *__restrict__ cursor_all,
                                                          float4 *__restrict__ sorted_all) {
  const int bi = blockIdx.y, k = blockIdx.x * 256 + threadIdx.x;
  if (k >= n) return;
  const float *p = xyz_all + ((size_t)bi * n + k) * 3;
  const float x = p[0], y = p[1], z = p[2];
  const unsigned hsh = bqg_hash(bqg_cell(x, inv_c), bqg_cell(y, inv_c), bqg_cell(z, inv_c), hmask);
  const int pos = atomicAdd(cursor_all + (size_t)bi * (hmask + 1) + hsh, 1);
  sorted_all[(size_t)bi * n + pos] = make_float4(x, y, z, __builtin_bit_cast(float, k));
}

__global__ __launch_bounds__(BQ_WAVES * 64) void bqg_query_kernel(
    int n, int m, float radius2, float inv_c, unsigned hmask, int nsample,
    const float *__restrict__ new_xyz_all, const float *__restrict__ xyz_all,
    const int *__restrict__ starts_all, const float4 *__restrict__ sorted_all, int *__restrict__ idx_all) {
  __shared__ int s_list[BQ_WAVES][BQG_CAP];
  const int lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int bi = blockIdx.y;
  const int j = blockIdx.x * BQ_WAVES + wave;  // this wave's centre
  if (j >= m) return;
  const float *ctr = new_xyz_all + ((size_t)bi * m + j) * 3;
  const float cx = ctr[0], cy = ctr[1], cz = ctr[2];
  const int *starts = starts_all + (size_t)bi * (hmask + 2);
  const float4 *sorted = sorted_all + (size_t)bi * n;
  int *row = idx_all + ((size_t)bi * m + j) * nsample;
  int *list = s_list[wave];

  // lane l < 27: neighbour cell l; a bucket shared with a lower lane (hash collision) is walked once
  const int l27 = lane < 27 ? lane : 0;
  const unsigned hsh = bqg_hash(bqg_cell(cx, inv_c) + (l27 % 3) - 1, bqg_cell(cy, inv_c) + (l27 / 3) % 3 - 1,
                                bqg_cell(cz, inv_c) + l27 / 9 - 1, hmask);
  bool mine = lane < 27;
  for (int o = 0; o < 26; ++o) {
    const unsigned other = (unsigned)__builtin_amdgcn_readlane((int)hsh, o);
    mine = mine && !(o < lane && other == hsh);
  }
  int pos = mine ? starts[hsh] : 0;
  const int end = mine ? starts[hsh + 1] : 0;

  int cnt = 0;
  while (__ballot(pos < end) != 0ull) {
    const bool act = pos < end;
    const float4 P = sorted[act ? pos : 0];
    const bool hit = act && (sq_dist3(cx, cy, cz, P.x, P.y, P.z) < radius2);  // ball_query_gpu.cu:31-33
    const unsigned long long mask = __ballot(hit);
    const int slot = cnt + __builtin_popcountll(mask & ((1ull << lane) - 1ull));
    if (hit && slot < BQG_CAP) list[slot] = __builtin_bit_cast(int, P.w);
    cnt += __builtin_popcountll(mask);
    ++pos;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

  if (cnt == 0) {  // ball_query.cpp:19-21: the row stays zero
    for (int l = lane; l < nsample; l += 64) row[l] = 0;
    return;
  }
  if (cnt <= BQG_CAP) {
    // rank of a hit = number of hits with a smaller index (indices are unique): index order restored
    int lo = 0x7FFFFFFF;
    for (int i = lane; i < cnt; i += 64) {
      const int v = list[i];
      int rank = 0;
      for (int q = 0; q < cnt; ++q) rank += list[q] < v;
      if (rank < nsample) row[rank] = v;
      lo = min(lo, v);
    }
    lo = (int)wave_allreduce_min_u32((unsigned)lo);  // first hit in index order: the padding value
    for (int l = cnt + lane; l < nsample; l += 64) row[l] = lo;
    return;
  }
  // dense neighbourhood: ordered scan of the whole scene for this centre (same code path as the
  // brute-force kernel with one centre)
  const float *xyz = xyz_all + (size_t)bi * n * 3;
  int have = 0, first = 0;
  for (int k0 = 0; k0 < n && have < nsample; k0 += 64) {
    const int k = k0 + lane;
    const bool inb = k < n;
    const int kk = inb ? k : n - 1;
    const bool hit = inb && (sq_dist3(cx, cy, cz, xyz[3 * kk + 0], xyz[3 * kk + 1], xyz[3 * kk + 2]) < radius2);
    const unsigned long long mask = __ballot(hit);
    if (mask != 0ull) {
      const int slot = have + __builtin_popcountll(mask & ((1ull << lane) - 1ull));
      if (hit && slot < nsample) row[slot] = k;
      if (have == 0) first = k0 + __builtin_ctzll(mask);
      have += __builtin_popcountll(mask);
    }
  }
  for (int l = min(have, nsample) + lane; l < nsample; l += 64) row[l] = first;
}

}  // namespace

extern "C" int sig3d_ball_query(int b, int n, int m, float radius, int nsample,
                                const float *new_xyz, const float *xyz, int *idx,
                                void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SIG3D_REQUIRE(b >= 0 && n >= 0 && m >= 0 && nsample >= 0, "negative size");
  if (b == 0 || m == 0 || nsample == 0) return 0;
  if (n == 0) {  // no points: all rows stay zero
    SIG3D_HIP_TRY(hipMemsetAsync(idx, 0, sizeof(int) * (size_t)b * m * nsample, stream));
    return 0;
  }
  const float radius2 = radius * radius;  // ball_query_gpu.cu:22, f32 product on the host
  // Centres per wave: 8 amortise the point loads best, but the scan is a chain of n / 64 dependent steps per
  // wave, and the small levels (SA2-4: 8192 / 4096 / 2048 centres in all) then leave most of the chip idle
  // behind 1024 / 512 / 256 long-running waves (20 us each for 2 MB of work).  Fewer centres per wave until
  // there are ~4096 waves: the scene (24 KB at SA2) is cache-resident anyway.
  const long centres = (long)b * m;
  const int cpw = centres >= 8L * 4096 ? 8 : centres >= 4L * 4096 ? 4 : centres >= 2L * 4096 ? 2 : 1;
  dim3 grid(sig3d_ceil_div(m, BQ_WAVES * cpw), b);
#define SIG3D_BQ_SCAN(CPW)                                                                                  \
  hipLaunchKernelGGL((ball_query_scan_kernel<CPW>), grid, dim3(BQ_WAVES * 64), 0, stream, n, m, radius2,    \
                     nsample, new_xyz, xyz, idx)
  if (cpw == 8) SIG3D_BQ_SCAN(8);
  else if (cpw == 4) SIG3D_BQ_SCAN(4);
  else if (cpw == 2) SIG3D_BQ_SCAN(2);
  else SIG3D_BQ_SCAN(1);
#undef SIG3D_BQ_SCAN
  SIG3D_LAUNCH_CHECK("ball_query_scan_kernel");
  return 0;
}

extern "C" int sig3d_ball_query_grid(int b, int n, int m, float radius, int nsample,
                                     const float *new_xyz, const float *xyz, int *idx, void *workspace,
                                     long workspace_bytes, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SIG3D_REQUIRE(b >= 0 && n >= 0 && m >= 0 && nsample >= 0, "negative size");
  if (b == 0 || m == 0 || nsample == 0) return 0;
  if (n < 1024 || !(radius > 0.f))  // tiny scenes / degenerate radius: the ordered scan is the right tool
    return sig3d_ball_query(b, n, m, radius, nsample, new_xyz, xyz, idx, stream_);
  int hsize = BQG_SCAN_BLOCK;
  while (hsize < 2 * n && hsize < (1 << 20)) hsize <<= 1;
  // workspace: counts[b][H] | starts[b][H+1] | cursor[b][H] | block_sums[b][H/2048] (ints), then
  // sorted[b][n] (float4, 16-byte aligned)
  const size_t ints = (size_t)b * (3 * (size_t)hsize + 1 + (size_t)hsize / BQG_SCAN_BLOCK);
  const size_t off_sorted = (ints * sizeof(int) + 15) / 16 * 16;
  const size_t need = off_sorted + (size_t)b * n * sizeof(float4);
  SIG3D_REQUIRE(workspace != nullptr && (size_t)workspace_bytes >= need,
                "workspace too small: b*(3*H+1+H/2048)*4 rounded up to 16 + b*n*16 bytes, H = pow2 >= max(2048, 2n) (<= 2^20)");
  int *counts = (int *)workspace;
  int *starts = counts + (size_t)b * hsize;
  int *cursor = starts + (size_t)b * (hsize + 1);
  int *block_sums = cursor + (size_t)b * hsize;
  float4 *sorted = (float4 *)((char *)workspace + off_sorted);
  const float inv_c = 1.f / (radius * 1.01f);
  const float radius2 = radius * radius;  // ball_query_gpu.cu:22, f32 product on the host
  const unsigned hmask = (unsigned)hsize - 1u;
  SIG3D_HIP_TRY(hipMemsetAsync(counts, 0, sizeof(int) * (size_t)b * hsize, stream));
  dim3 pgrid(sig3d_ceil_div(n, 256), b);
  hipLaunchKernelGGL(bqg_count_kernel, pgrid, dim3(256), 0, stream, n, inv_c, hmask, xyz, counts);
  dim3 sgrid(hsize / BQG_SCAN_BLOCK, b);
  hipLaunchKernelGGL(bqg_scan_local_kernel, sgrid, dim3(256), 0, stream, hsize, counts, starts, block_sums);
  hipLaunchKernelGGL(bqg_scan_offset_kernel, sgrid, dim3(256), 0, stream, hsize, block_sums, starts, cursor);
  hipLaunchKernelGGL(bqg_scatter_kernel, pgrid, dim3(256), 0, stream, n, inv_c, hmask, xyz, cursor, sorted);
  hipLaunchKernelGGL(bqg_query_kernel, dim3(sig3d_ceil_div(m, BQ_WAVES), b), dim3(BQ_WAVES * 64), 0, stream, n, m,
                     radius2, inv_c, hmask, nsample, new_xyz, xyz, starts, sorted, idx);
  SIG3D_LAUNCH_CHECK("ball_query grid kernels");
  return 0;
}
