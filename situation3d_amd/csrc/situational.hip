// situational.hip -- situational pose re-encode of point / token coordinates for gfx950.
//
// Replaces situation3d/utils/temp.py:42-97 of the reference (batch_matrix_function followed
// by the homogeneous bmm): pose (b,7) = [t, q_xyzw]  ->  p' = R(q) p + t, with R spelled
// exactly as temp.py:63-73 (x2-y2-z2+w2 diagonal; equal to the unit-quaternion form of
// sqa_module.py:12-30 only when |q| = 1).  `inverse` selects the agent-frame map
// R^T (p - t), which the reference never writes down but SIG3D's situational encoding needs.
//
// HBM-bound (24 B / point): the 3x4 matrix is built once per workgroup in registers, points
// are streamed with 12-byte-per-lane accesses.  The backward pass produces grad_points in the
// same sweep and folds the pose gradient per workgroup (12 partial sums -> 7 atomics), so
// there is no (b,n,12) intermediate.
#include "sig3d_common.h"
#include "situational_pose.h"

namespace {

constexpr int ST_THREADS = 256;

template <bool INVERSE>
__global__ __launch_bounds__(ST_THREADS) void situational_fwd_kernel(int n,
                                                                     const float *__restrict__ pose,
                                                                     const float *__restrict__ points,
                                                                     float *__restrict__ out) {
  const int bi = blockIdx.y;
  const Pose P = load_pose(pose + (size_t)bi * 7);
  const float *pin = points + (size_t)bi * n * 3;
  float *pout = out + (size_t)bi * n * 3;
  for (int j = blockIdx.x * ST_THREADS + threadIdx.x; j < n; j += gridDim.x * ST_THREADS) {
    float o[3];
    apply_pose<INVERSE>(P, pin[3 * j + 0], pin[3 * j + 1], pin[3 * j + 2], o);
    pout[3 * j + 0] = o[0];
    pout[3 * j + 1] = o[1];
    pout[3 * j + 2] = o[2];
  }
}

// dL/dq from G[a][c] = dL/dR[a][c] (R as in load_pose); linear in G, so per-block partials of
// G may be converted before they are accumulated.
__device__ __forceinline__ void rot_grad_to_quat(const float G[3][3], float x, float y, float z,
                                                 float w, float *gq) {
  gq[0] = 2.f * (x * G[0][0] + y * G[1][0] + z * G[2][0] + y * G[0][1] - x * G[1][1] + w * G[2][1] +
                 z * G[0][2] - w * G[1][2] - x * G[2][2]);
  gq[1] = 2.f * (-y * G[0][0] + x * G[1][0] - w * G[2][0] + x * G[0][1] + y * G[1][1] + z * G[2][1] +
                 w * G[0][2] + z * G[1][2] - y * G[2][2]);
  gq[2] = 2.f * (-z * G[0][0] + w * G[1][0] + x * G[2][0] - w * G[0][1] - z * G[1][1] + y * G[2][1] +
                 x * G[0][2] + y * G[1][2] + z * G[2][2]);
  gq[3] = 2.f * (w * G[0][0] + z * G[1][0] - y * G[2][0] - z * G[0][1] + w * G[1][1] + x * G[2][1] +
                 y * G[0][2] - x * G[1][2] + w * G[2][2]);
}

template <bool INVERSE>
__global__ __launch_bounds__(ST_THREADS) void situational_bwd_kernel(
    int n, const float *__restrict__ pose, const float *__restrict__ points,
    const float *__restrict__ grad_out, float *__restrict__ grad_points,
    float *__restrict__ grad_pose) {
  __shared__ float s_part[ST_THREADS / 64][12];
  const int bi = blockIdx.y;
  const float *q7 = pose + (size_t)bi * 7;
  const Pose P = load_pose(q7);
  const float *pin = points + (size_t)bi * n * 3;
  const float *gin = grad_out + (size_t)bi * n * 3;
  float *gp = grad_points + (size_t)bi * n * 3;
  float acc[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) acc[i] = 0.f;
  for (int j = blockIdx.x * ST_THREADS + threadIdx.x; j < n; j += gridDim.x * ST_THREADS) {
    const float g0 = gin[3 * j + 0], g1 = gin[3 * j + 1], g2 = gin[3 * j + 2];
    const float p0 = pin[3 * j + 0], p1 = pin[3 * j + 1], p2 = pin[3 * j + 2];
    const float g[3] = {g0, g1, g2};
    if (!INVERSE) {
      // out = R p + t : dp = R^T g, dt = g, dR[a][c] = g_a p_c
      const float p[3] = {p0, p1, p2};
#pragma unroll
      for (int c = 0; c < 3; ++c)
        gp[3 * j + c] = g0 * P.r[0][c] + g1 * P.r[1][c] + g2 * P.r[2][c];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[3 * a + c] += g[a] * p[c];
        acc[9 + a] += g[a];
      }
    } else {
      // out_r = sum_c R[c][r] (p_c - t_c) : dp = R g, dt = -R g, dR[c][r] = (p_c - t_c) g_r
      const float qv[3] = {p0 - P.t[0], p1 - P.t[1], p2 - P.t[2]};
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float d = P.r[c][0] * g0 + P.r[c][1] * g1 + P.r[c][2] * g2;
        gp[3 * j + c] = d;
        acc[9 + c] -= d;
#pragma unroll
        for (int r = 0; r < 3; ++r) acc[3 * c + r] += qv[c] * g[r];
      }
    }
  }
  const int lane = lane_id(), wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    const float s = wave_allreduce_sum_f32(acc[i]);
    if (lane == 0) s_part[wave][i] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float G[3][3], gt[3];
    for (int i = 0; i < 12; ++i) {
      float s = 0.f;
      for (int wv = 0; wv < ST_THREADS / 64; ++wv) s += s_part[wv][i];
      if (i < 9) G[i / 3][i % 3] = s; else gt[i - 9] = s;
    }
    float gq[4];
    rot_grad_to_quat(G, q7[3], q7[4], q7[5], q7[6], gq);
    float *o = grad_pose + (size_t)bi * 7;
    unsafeAtomicAdd(o + 0, gt[0]);
    unsafeAtomicAdd(o + 1, gt[1]);
    unsafeAtomicAdd(o + 2, gt[2]);
    unsafeAtomicAdd(o + 3, gq[0]);
    unsafeAtomicAdd(o + 4, gq[1]);
    unsafeAtomicAdd(o + 5, gq[2]);
    unsafeAtomicAdd(o + 6, gq[3]);
  }
}

inline int st_blocks(int n) {
  int g = sig3d_ceil_div(n, ST_THREADS * 4);
  return g < 1 ? 1 : (g > 256 ? 256 : g);
}

}  // namespace

extern "C" int sig3d_situational_transform(int b, int n, const float *pose, const float *points,
                                           float *out, int inverse, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SIG3D_REQUIRE(b >= 0 && n >= 0, "negative size");
  if (b == 0 || n == 0) return 0;
  dim3 grid(st_blocks(n), b);
  if (inverse)
    hipLaunchKernelGGL((situational_fwd_kernel<true>), grid, dim3(ST_THREADS), 0, stream, n, pose,
                       points, out);
  else
    hipLaunchKernelGGL((situational_fwd_kernel<false>), grid, dim3(ST_THREADS), 0, stream, n, pose,
                       points, out);
  SIG3D_LAUNCH_CHECK("situational_fwd_kernel");
  return 0;
}

extern "C" int sig3d_situational_transform_grad(int b, int n, const float *pose,
                                                const float *points, const float *grad_out,
                                                float *grad_points, float *grad_pose,
                                                int inverse, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SIG3D_REQUIRE(b >= 0 && n >= 0, "negative size");
  if (b == 0) return 0;
  SIG3D_HIP_TRY(hipMemsetAsync(grad_pose, 0, sizeof(float) * (size_t)b * 7, stream));
  if (n == 0) return 0;
  dim3 grid(st_blocks(n), b);
  if (inverse)
    hipLaunchKernelGGL((situational_bwd_kernel<true>), grid, dim3(ST_THREADS), 0, stream, n, pose,
                       points, grad_out, grad_points, grad_pose);
  else
    hipLaunchKernelGGL((situational_bwd_kernel<false>), grid, dim3(ST_THREADS), 0, stream, n, pose,
                       points, grad_out, grad_points, grad_pose);
  SIG3D_LAUNCH_CHECK("situational_bwd_kernel");
  return 0;
}
