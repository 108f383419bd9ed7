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
  dim3 grid(sig3d_ceil_div(m, BQ_WAVES * BQ_CPW), b);
  hipLaunchKernelGGL((ball_query_scan_kernel<BQ_CPW>), grid, dim3(BQ_WAVES * 64), 0, stream, n, m,
                     radius2, nsample, new_xyz, xyz, idx);
  SIG3D_LAUNCH_CHECK("ball_query_scan_kernel");
  return 0;
}
