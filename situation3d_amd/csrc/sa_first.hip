// sa_first.hip -- the FIRST SharedMLP layer of the first set-abstraction level, gathered from the raw scan.
//
// SA1 groups (xyz - centre) / radius and the 3 colour channels of every neighbour and feeds the 6-channel column
// to Conv2d(6, 64, 1) (lib/pointnet2/pointnet2_utils.py:348-359 + pytorch_utils.py:11-36).  With 6 input channels
// that layer is no matrix-core work at all (6 FMAs per output); what it costs is the grouped tensor in between:
// written by one launch, read by the next -- and gathered as 4-byte pieces out of a channel-major (B, 3, N) copy
// of the colours that a transpose launch had to make first (36 MB fetched for 11 MB of columns, by the PMC).
// Here the layer reads the scan as it arrives -- point-major rows [x y z r g b ...] of 12 + 4c bytes, one row
// per neighbour -- forms the column in registers, multiplies by W (scalar loads: the weights are wave-uniform) and
// writes the layer's pre-activation rows + BatchNorm batch statistics.  No grouped tensor, no colour transpose;
// the weight gradient gathers the rows once more (recompute in backward).  Dense lists or compact lists
// (compact.hip: distinct neighbours, statistics weighted by the multiplicity).
#include "sig3d_common.h"

namespace {

constexpr int SF_THREADS = 256;
constexpr int SF_MAXCIN = 8;

// column of position u: (xyz[k] - centre[j]) (/ radius), then the point's feature channels
template <int CIN>
__device__ __forceinline__ void sf_column(float (&x)[CIN], const float *__restrict__ row,
                                          const float *__restrict__ ctr, bool normalize, float radius) {
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float d = __fsub_rn(row[a], ctr[a]);          // pointnet2_utils.py:352, individually rounded like the
    x[a] = normalize ? __fdiv_rn(d, radius) : d;        // grouping kernels (:354 divides, it does not multiply)
  }
#pragma unroll
  for (int a = 3; a < CIN; ++a) x[a] = row[a];
}

// sum of 32 per-lane values over the 64 lanes of a wave through a wave-private LDS tile: lane l returns
// sum_lanes v[l & 31].  (A register butterfly of selects + cross-lane moves compiled to ~5000 instructions and 40 us.)
__device__ __forceinline__ float sf_wave_column_sums(const float (&v)[32], float (*tile)[33], int lane) {
#pragma unroll
  for (int i = 0; i < 32; ++i) tile[lane][i] = v[i];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  const int col = lane & 31, r0 = (lane >> 5) * 32;
  float t = 0.f;
#pragma unroll
  for (int r = 0; r < 32; ++r) t += tile[r0 + r][col];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();   // the tile is free again
  return t + __shfl_xor(t, 32);
}

template <int CIN>
__global__ __launch_bounds__(SF_THREADS) void sa_first_layer_fwd_kernel(
    int n, int m, int ns, int cpt, int cout, int normalize, float radius, const float *__restrict__ points,
    const float *__restrict__ new_xyz, const int *__restrict__ idx, const int *__restrict__ centre_of,
    const int *__restrict__ n_act, const float *__restrict__ mult_all, const float *__restrict__ w,
    float *__restrict__ y, double *__restrict__ stat_sum, double *__restrict__ stat_sq) {
  __shared__ float s_red[2][SF_THREADS / 64][32];
  __shared__ float s_tile[SF_THREADS / 64][64][33];
  const int bi = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long e = (long)m * ns;
  const long En = n_act ? (long)n_act[bi] : e;
  if ((long)blockIdx.x * SF_THREADS >= En) return;   // compact lists: nothing for this workgroup
  const int *ib = idx + (size_t)bi * e;
  const float *pb = points + (size_t)bi * n * cpt;
  const float *cb = new_xyz + (size_t)bi * m * 3;
  // persistent: the workgroups of a scene stride over its 256-position chunks, statistics stay in registers;
  // 32 output channels per pass (their weights are scalar loads hoisted out of the chunk loop; gathering the
  // column once for all passes instead was measured slower: 35 vs 32 us, the weight loads land inside the loop)
  for (int g0 = 0; g0 < cout; g0 += 32) {
    float s1 = 0.f, s2 = 0.f;   // channel g0 + (lane & 31), summed over the chunks of this workgroup
    for (long u0 = (long)blockIdx.x * SF_THREADS; u0 < En; u0 += (long)gridDim.x * SF_THREADS) {
      const long u = u0 + tid;
      const bool live = u < En;
      const long uu = live ? u : En - 1;
      const int k = ib[uu];
      const int j = centre_of ? centre_of[(size_t)bi * e + uu] : (int)(uu / ns);
      float x[CIN];
      sf_column<CIN>(x, pb + (size_t)k * cpt, cb + 3 * j, normalize != 0, radius);
      const float mu = live ? (mult_all ? mult_all[(size_t)bi * e + uu] : 1.f) : 0.f;
      float v1[32], v2[32];
#pragma unroll
      for (int co = 0; co < 32; ++co) {
        const float *wr = w + (size_t)(g0 + co) * CIN;   // wave-uniform: scalar loads
        float acc = 0.f;
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) acc += wr[ci] * x[ci];
        if (live) y[((size_t)bi * cout + g0 + co) * e + u] = acc;
        v1[co] = mu * acc;
        v2[co] = mu * acc * acc;
      }
      if (stat_sum) {
        s1 += sf_wave_column_sums(v1, s_tile[wave], lane);
        s2 += sf_wave_column_sums(v2, s_tile[wave], lane);
      }
    }
    if (stat_sum) {
      if (lane < 32) {
        s_red[0][wave][lane] = s1;
        s_red[1][wave][lane] = s2;
      }
      __syncthreads();
      if (tid < 64) {
        const int q = tid >> 5, ch = tid & 31;
        float t = 0.f;
#pragma unroll
        for (int wv = 0; wv < SF_THREADS / 64; ++wv) t += s_red[q][wv][ch];
        unsafeAtomicAdd((q ? stat_sq : stat_sum) + g0 + ch, (double)t);
      }
      __syncthreads();
    }
  }
}

// dW (cout, CIN) += sum_u dY[:, u] x[u]^T over the live positions: a workgroup turns a 256-position chunk of
// columns through LDS, thread (channel, quarter) walks 64 positions of its dY row
template <int CIN>
__global__ __launch_bounds__(SF_THREADS) void sa_first_layer_dw_kernel(
    int n, int m, int ns, int cpt, int cout, int normalize, float radius, const float *__restrict__ points,
    const float *__restrict__ new_xyz, const int *__restrict__ idx, const int *__restrict__ centre_of,
    const int *__restrict__ n_act, const float *__restrict__ dY, float *__restrict__ dW) {
  __shared__ float xs[CIN][SF_THREADS];
  __shared__ float s_acc[4][64][CIN];
  const int bi = blockIdx.y, tid = threadIdx.x;
  const long e = (long)m * ns;
  const long En = n_act ? (long)n_act[bi] : e;
  if ((long)blockIdx.x * SF_THREADS >= En) return;
  const int *ib = idx + (size_t)bi * e;
  const float *pb = points + (size_t)bi * n * cpt;
  const float *cb = new_xyz + (size_t)bi * m * 3;
  const int co_l = tid & 63, quarter = tid >> 6;
  for (int g0 = 0; g0 < cout; g0 += 64) {
    float acc[CIN];
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci) acc[ci] = 0.f;
    const float *drow = dY + ((size_t)bi * cout + g0 + co_l) * e;
    for (long u0 = (long)blockIdx.x * SF_THREADS; u0 < En; u0 += (long)gridDim.x * SF_THREADS) {
      const long u = u0 + tid;
      const bool live = u < En;
      const long uu = live ? u : En - 1;
      const int k = ib[uu];
      const int j = centre_of ? centre_of[(size_t)bi * e + uu] : (int)(uu / ns);
      float x[CIN];
      sf_column<CIN>(x, pb + (size_t)k * cpt, cb + 3 * j, normalize != 0, radius);
      __syncthreads();   // the previous chunk's columns have been consumed
#pragma unroll
      for (int ci = 0; ci < CIN; ++ci) xs[ci][tid] = live ? x[ci] : 0.f;   // dead positions add exact zeros
      __syncthreads();
      const long q0 = u0 + 64 * quarter;
      float d[64];
#pragma unroll
      for (int i = 0; i < 64; ++i) d[i] = (q0 + i < En) ? drow[q0 + i] : 0.f;
#pragma unroll
      for (int i = 0; i < 64; ++i)
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) acc[ci] += d[i] * xs[ci][64 * quarter + i];
    }
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci) s_acc[quarter][co_l][ci] = acc[ci];
    __syncthreads();
    for (int i = tid; i < 64 * CIN; i += SF_THREADS) {
      const int co = i / CIN, ci = i % CIN;
      const float t = (s_acc[0][co][ci] + s_acc[1][co][ci]) + (s_acc[2][co][ci] + s_acc[3][co][ci]);
      unsafeAtomicAdd(dW + (size_t)(g0 + co) * CIN + ci, t);
    }
    __syncthreads();
  }
}

int sf_check(int b, int n, int m, int ns, int cpt, int cin, int cout) {
  if (!(b >= 0 && n >= 1 && m >= 0 && ns >= 1)) return 1;
  if (!(cin >= 3 && cin <= SF_MAXCIN && cpt >= cin)) return 2;
  if (!(cout >= 64 && cout % 64 == 0)) return 3;   // the weight gradient walks 64 channels per pass
  if (!((long)m * ns < (1L << 31) && (long)b * n * cpt < (1L << 40))) return 4;
  return 0;
}

}  // namespace

#define SF_DISPATCH(KERNEL, ...)                                                                       \
  switch (cin) {                                                                                       \
    case 3: hipLaunchKernelGGL((KERNEL<3>), grid, dim3(SF_THREADS), 0, stream, __VA_ARGS__); break;    \
    case 4: hipLaunchKernelGGL((KERNEL<4>), grid, dim3(SF_THREADS), 0, stream, __VA_ARGS__); break;    \
    case 5: hipLaunchKernelGGL((KERNEL<5>), grid, dim3(SF_THREADS), 0, stream, __VA_ARGS__); break;    \
    case 6: hipLaunchKernelGGL((KERNEL<6>), grid, dim3(SF_THREADS), 0, stream, __VA_ARGS__); break;    \
    case 7: hipLaunchKernelGGL((KERNEL<7>), grid, dim3(SF_THREADS), 0, stream, __VA_ARGS__); break;    \
    default: hipLaunchKernelGGL((KERNEL<8>), grid, dim3(SF_THREADS), 0, stream, __VA_ARGS__); break;   \
  }

extern "C" int sig3d_sa_first_layer_fwd(int b, int n, int m, int nsample, int cpt, int cin, int cout, int normalize_xyz,
                                        float radius, const float *points_pm, const float *new_xyz, const int *idx,
                                        const int *centre_of, const int *n_act, const float *mult, const float *w,
                                        float *y, double *stat_sum, double *stat_sq, int accumulate, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SIG3D_REQUIRE(sf_check(b, n, m, nsample, cpt, cin, cout) == 0,
                "sa_first_layer: 3 <= cin <= 8 <= ... cpt >= cin, cout a multiple of 64");
  SIG3D_REQUIRE(points_pm && new_xyz && idx && w && y, "null argument");
  SIG3D_REQUIRE((centre_of == nullptr) == (n_act == nullptr), "compact lists need centre_of and n_act");
  SIG3D_REQUIRE((stat_sum == nullptr) == (stat_sq == nullptr), "statistics come in pairs");
  if (stat_sum && !accumulate) {
    SIG3D_HIP_TRY(hipMemsetAsync(stat_sum, 0, sizeof(double) * cout, stream));
    SIG3D_HIP_TRY(hipMemsetAsync(stat_sq, 0, sizeof(double) * cout, stream));
  }
  if (b == 0 || m == 0) return 0;
  const long e = (long)m * nsample;
  int per_scene = sig3d_ceil_div(e, SF_THREADS);
  const int cap = sig3d_ceil_div(512, b);   // ~512 workgroups in all: one resident round, few statistic atomics
  if (per_scene > cap) per_scene = cap;
  dim3 grid(per_scene, b);
  SF_DISPATCH(sa_first_layer_fwd_kernel, n, m, nsample, cpt, cout, normalize_xyz, radius, points_pm, new_xyz, idx,
              centre_of, n_act, mult, w, y, stat_sum, stat_sq);
  SIG3D_LAUNCH_CHECK("sa_first_layer_fwd_kernel");
  return 0;
}

extern "C" int sig3d_sa_first_layer_dw(int b, int n, int m, int nsample, int cpt, int cin, int cout, int normalize_xyz,
                                       float radius, const float *points_pm, const float *new_xyz, const int *idx,
                                       const int *centre_of, const int *n_act, const float *dY, float *dW,
                                       int accumulate, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SIG3D_REQUIRE(sf_check(b, n, m, nsample, cpt, cin, cout) == 0,
                "sa_first_layer: 3 <= cin <= 8 <= ... cpt >= cin, cout a multiple of 64");
  SIG3D_REQUIRE(points_pm && new_xyz && idx && dY && dW, "null argument");
  SIG3D_REQUIRE((centre_of == nullptr) == (n_act == nullptr), "compact lists need centre_of and n_act");
  if (!accumulate) SIG3D_HIP_TRY(hipMemsetAsync(dW, 0, sizeof(float) * (size_t)cout * cin, stream));
  if (b == 0 || m == 0) return 0;
  const long e = (long)m * nsample;
  int per_scene = sig3d_ceil_div(e, SF_THREADS);
  const int cap = sig3d_ceil_div(512, b);
  if (per_scene > cap) per_scene = cap;
  dim3 grid(per_scene, b);
  SF_DISPATCH(sa_first_layer_dw_kernel, n, m, nsample, cpt, cout, normalize_xyz, radius, points_pm, new_xyz, idx,
              centre_of, n_act, dY, dW);
  SIG3D_LAUNCH_CHECK("sa_first_layer_dw_kernel");
  return 0;
}
