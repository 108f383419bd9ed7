// interpolate.hip -- three_nn / three_interpolate (+grad) for gfx950.
//
// Replaces lib/pointnet2/_ext_src/src/interpolate_gpu.cu of the reference.
//
// three_nn: one lane per unknown point; the known set is streamed through LDS in tiles of
// TN_TILE points that the whole workgroup loads coalesced (the reference lets every thread
// re-read all of `known` from global with 12-byte strides).  Selection semantics are the
// reference's: strict '<' insertion into three running bests kept in DOUBLE and initialised
// to 1e40 (interpolate_gpu.cu:27-50), so the lowest index wins ties and slots that never get
// filled (m < 3) come out as (+inf, index 0).
#include "sig3d_common.h"

namespace {

constexpr int TN_THREADS = 256;
constexpr int TN_TILE = 1024;  // known points per LDS tile (12 KiB)

__global__ __launch_bounds__(TN_THREADS) void three_nn_kernel(int n, int m,
                                                              const float *__restrict__ unknown,
                                                              const float *__restrict__ known,
                                                              float *__restrict__ dist2,
                                                              int *__restrict__ idx) {
  __shared__ float s_known[TN_TILE * 3];
  const int bi = blockIdx.y;
  const int j = blockIdx.x * TN_THREADS + threadIdx.x;
  const bool active = j < n;
  const float *kn = known + (size_t)bi * m * 3;
  float ux = 0.f, uy = 0.f, uz = 0.f;
  if (active) {
    const float *u = unknown + ((size_t)bi * n + j) * 3;
    ux = u[0]; uy = u[1]; uz = u[2];
  }
  double best1 = 1e40, best2 = 1e40, best3 = 1e40;
  int besti1 = 0, besti2 = 0, besti3 = 0;
  for (int t0 = 0; t0 < m; t0 += TN_TILE) {
    const int tn = min(TN_TILE, m - t0);
    __syncthreads();
    for (int f = threadIdx.x; f < tn * 3; f += TN_THREADS) s_known[f] = kn[(size_t)t0 * 3 + f];
    __syncthreads();
    if (active) {
      for (int kk = 0; kk < tn; ++kk) {
        const float d = sq_dist3(ux, uy, uz, s_known[3 * kk + 0], s_known[3 * kk + 1],
                                 s_known[3 * kk + 2]);
        const int k = t0 + kk;
        const double dd = (double)d;
        if (dd < best1) {
          best3 = best2; besti3 = besti2;
          best2 = best1; besti2 = besti1;
          best1 = dd; besti1 = k;
        } else if (dd < best2) {
          best3 = best2; besti3 = besti2;
          best2 = dd; besti2 = k;
        } else if (dd < best3) {
          best3 = dd; besti3 = k;
        }
      }
    }
  }
  if (active) {
    float *d2 = dist2 + ((size_t)bi * n + j) * 3;
    int *ix = idx + ((size_t)bi * n + j) * 3;
    d2[0] = (float)best1; d2[1] = (float)best2; d2[2] = (float)best3;  // 1e40 -> +inf
    ix[0] = besti1; ix[1] = besti2; ix[2] = besti3;
  }
}

// out[b,l,j] = p[i1]*w1 + p[i2]*w2 + p[i3]*w3, left to right, unfused (interpolate_gpu.cu:99-100)
__global__ __launch_bounds__(256) void three_interpolate_kernel(int c, int m, int n,
                                                                const float *__restrict__ points,
                                                                const int *__restrict__ idx,
                                                                const float *__restrict__ weight,
                                                                float *__restrict__ out) {
  const int bi = blockIdx.z;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const float *w = weight + ((size_t)bi * n + j) * 3;
  const int *ix = idx + ((size_t)bi * n + j) * 3;
  const float w1 = w[0], w2 = w[1], w3 = w[2];
  const int i1 = ix[0], i2 = ix[1], i3 = ix[2];
  for (int l = blockIdx.y; l < c; l += gridDim.y) {
    const float *row = points + ((size_t)bi * c + l) * m;
    out[((size_t)bi * c + l) * n + j] =
        __fadd_rn(__fadd_rn(__fmul_rn(row[i1], w1), __fmul_rn(row[i2], w2)), __fmul_rn(row[i3], w3));
  }
}

__global__ __launch_bounds__(256) void three_interpolate_grad_kernel(
    int c, int n, int m, const float *__restrict__ grad_out, const int *__restrict__ idx,
    const float *__restrict__ weight, float *__restrict__ grad_points) {
  const int bi = blockIdx.z;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const float *w = weight + ((size_t)bi * n + j) * 3;
  const int *ix = idx + ((size_t)bi * n + j) * 3;
  const float w1 = w[0], w2 = w[1], w3 = w[2];
  const int i1 = ix[0], i2 = ix[1], i3 = ix[2];
  for (int l = blockIdx.y; l < c; l += gridDim.y) {
    const float g = grad_out[((size_t)bi * c + l) * n + j];
    float *row = grad_points + ((size_t)bi * c + l) * m;
    unsafeAtomicAdd(row + i1, __fmul_rn(g, w1));
    unsafeAtomicAdd(row + i2, __fmul_rn(g, w2));
    unsafeAtomicAdd(row + i3, __fmul_rn(g, w3));
  }
}

}  // namespace

extern "C" int sig3d_three_nn(int b, int n, int m, const float *unknown, const float *known,
                              float *dist2, int *idx, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SIG3D_REQUIRE(b >= 0 && n >= 0 && m >= 0, "negative size");
  if (b == 0 || n == 0) return 0;
  dim3 grid(sig3d_ceil_div(n, TN_THREADS), b);
  hipLaunchKernelGGL(three_nn_kernel, grid, dim3(TN_THREADS), 0, stream, n, m, unknown, known,
                     dist2, idx);
  SIG3D_LAUNCH_CHECK("three_nn_kernel");
  return 0;
}

extern "C" int sig3d_three_interpolate(int b, int c, int m, int n, const float *points,
                                       const int *idx, const float *weight, float *out,
                                       void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SIG3D_REQUIRE(b >= 0 && c >= 0 && n >= 0 && m >= 0, "negative size");
  if (b == 0 || c == 0 || n == 0) return 0;
  dim3 grid(sig3d_ceil_div(n, 256), c < 128 ? c : 128, b);
  hipLaunchKernelGGL(three_interpolate_kernel, grid, dim3(256), 0, stream, c, m, n, points, idx,
                     weight, out);
  SIG3D_LAUNCH_CHECK("three_interpolate_kernel");
  return 0;
}

extern "C" int sig3d_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out,
                                            const int *idx, const float *weight,
                                            float *grad_points, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SIG3D_REQUIRE(b >= 0 && c >= 0 && n >= 0 && m >= 0, "negative size");
  if (b == 0 || c == 0 || m == 0) return 0;
  SIG3D_HIP_TRY(hipMemsetAsync(grad_points, 0, sizeof(float) * (size_t)b * c * m, stream));
  if (n == 0) return 0;
  dim3 grid(sig3d_ceil_div(n, 256), c < 128 ? c : 128, b);
  hipLaunchKernelGGL(three_interpolate_grad_kernel, grid, dim3(256), 0, stream, c, n, m, grad_out,
                     idx, weight, grad_points);
  SIG3D_LAUNCH_CHECK("three_interpolate_grad_kernel");
  return 0;
}
