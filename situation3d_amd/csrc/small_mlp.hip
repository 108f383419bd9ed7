// small_mlp.hip -- the small dense layers around the Q-Former as single launches.
//
// (1) The positional MLP of the scene tokens, sqa_module.py:274-278 / :319-321:
//         tokens = tok_feat + Linear(hid, cout)(GELU(Linear(cin, hid)(pos)))        cin = 2 or 3, hid = 128, cout = 256
//     torch runs it as addmm + gelu + addmm + add (4 launches) and ~12 in the backward pass (85 us for 2048 rows).
//     Here: one forward launch (a workgroup owns 16 token rows: hidden activations in LDS, a thread per output
//     channel), and two backward launches (input-side: dH = dY W2, dPre, dW1, db1; weight-side: dW2 = dY^T H, db2
//     with H recomputed from the stored pre-activation).
#include "sig3d_common.h"

namespace {

__device__ __forceinline__ float sm_gelu(float u) { return 0.5f * u * (1.f + erff(u * 0.70710678118654752440f)); }
__device__ __forceinline__ float sm_gelu_grad(float u) {
  const float cdf = 0.5f * (1.f + erff(u * 0.70710678118654752440f));
  return cdf + u * 0.39894228040143267794f * __expf(-0.5f * u * u);
}

constexpr int PE_ROWS = 16;     // token rows per workgroup
constexpr int PE_MAXHID = 128;
constexpr int PE_MAXCIN = 4;

// forward: pre (rows, hid) = x W1^T + b1 (kept for the backward pass), out = residual + gelu(pre) W2^T + b2
__global__ __launch_bounds__(256) void pos_mlp_fwd_kernel(int rows, int cin, int hid, int cout,
                                                          const float *__restrict__ x, const float *__restrict__ w1,
                                                          const float *__restrict__ b1, const float *__restrict__ w2,
                                                          const float *__restrict__ b2, const float *__restrict__ residual,
                                                          float *__restrict__ pre, float *__restrict__ out) {
  __shared__ float s_h[PE_ROWS][PE_MAXHID + 4];
  __shared__ float s_x[PE_ROWS][PE_MAXCIN];
  const int r0 = blockIdx.x * PE_ROWS, tid = threadIdx.x;
  if (tid < PE_ROWS * PE_MAXCIN) {
    const int r = tid / PE_MAXCIN, c = tid % PE_MAXCIN;
    s_x[r][c] = (r0 + r < rows && c < cin) ? x[(size_t)(r0 + r) * cin + c] : 0.f;
  }
  __syncthreads();
  for (int i = tid; i < PE_ROWS * hid; i += 256) {
    const int r = i / hid, j = i % hid;
    float u = b1[j];
    for (int c = 0; c < cin; ++c) u += w1[j * cin + c] * s_x[r][c];
    s_h[r][j] = sm_gelu(u);
    if (r0 + r < rows) pre[(size_t)(r0 + r) * hid + j] = u;
  }
  __syncthreads();
  for (int c = tid; c < cout; c += 256) {
    float acc[PE_ROWS];
    const float bias = b2[c];
#pragma unroll
    for (int r = 0; r < PE_ROWS; ++r) acc[r] = bias;
    const float4 *wr = reinterpret_cast<const float4 *>(w2 + (size_t)c * hid);
    for (int k4 = 0; k4 < hid / 4; ++k4) {
      const float4 w = wr[k4];
#pragma unroll
      for (int r = 0; r < PE_ROWS; ++r) {
        const float4 hv = *reinterpret_cast<const float4 *>(&s_h[r][4 * k4]);   // same address for every lane: broadcast
        acc[r] += w.x * hv.x + w.y * hv.y + w.z * hv.z + w.w * hv.w;
      }
    }
#pragma unroll
    for (int r = 0; r < PE_ROWS; ++r)
      if (r0 + r < rows) {
        const size_t o = (size_t)(r0 + r) * cout + c;
        out[o] = (residual ? residual[o] : 0.f) + acc[r];
      }
  }
}

// backward, input side: dH = dY W2 (columns of W2 read coalesced), dPre = dH * gelu'(pre) (written: the caller may need
// dX = dPre W1), dW1 += dPre^T x, db1 += column sums of dPre.  dW1 / db1 must be zero on entry.
__global__ __launch_bounds__(256) void pos_mlp_bwd_in_kernel(int rows, int cin, int hid, int cout,
                                                             const float *__restrict__ x, const float *__restrict__ w2,
                                                             const float *__restrict__ pre, const float *__restrict__ dy,
                                                             float *__restrict__ dpre, float *__restrict__ dw1,
                                                             float *__restrict__ db1) {
  extern __shared__ float s_dy[];             // [PE_ROWS][cout]
  __shared__ float s_x[PE_ROWS][PE_MAXCIN];
  __shared__ float s_p[2][PE_MAXHID][PE_MAXCIN + 1];
  const int r0 = blockIdx.x * PE_ROWS, tid = threadIdx.x;
  for (int i = tid; i < PE_ROWS * cout; i += 256) {
    const int r = i / cout, c = i % cout;
    s_dy[i] = (r0 + r < rows) ? dy[(size_t)(r0 + r) * cout + c] : 0.f;
  }
  if (tid < PE_ROWS * PE_MAXCIN) {
    const int r = tid / PE_MAXCIN, c = tid % PE_MAXCIN;
    s_x[r][c] = (r0 + r < rows && c < cin) ? x[(size_t)(r0 + r) * cin + c] : 0.f;
  }
  __syncthreads();
  // thread (k, half): hidden unit k, rows [8 half, 8 half + 8)
  const int k = tid % PE_MAXHID, hf = tid / PE_MAXHID;
  float part[PE_MAXCIN + 1];
#pragma unroll
  for (int c = 0; c <= PE_MAXCIN; ++c) part[c] = 0.f;
  if (k < hid) {
    float acc[PE_ROWS / 2];
#pragma unroll
    for (int r = 0; r < PE_ROWS / 2; ++r) acc[r] = 0.f;
    for (int c = 0; c < cout; ++c) {
      const float w = w2[(size_t)c * hid + k];   // consecutive k: coalesced
#pragma unroll
      for (int r = 0; r < PE_ROWS / 2; ++r) acc[r] += w * s_dy[(hf * (PE_ROWS / 2) + r) * cout + c];
    }
#pragma unroll
    for (int r = 0; r < PE_ROWS / 2; ++r) {
      const int row = r0 + hf * (PE_ROWS / 2) + r;
      if (row < rows) {
        const float g = acc[r] * sm_gelu_grad(pre[(size_t)row * hid + k]);
        dpre[(size_t)row * hid + k] = g;
        part[PE_MAXCIN] += g;
#pragma unroll
        for (int c = 0; c < PE_MAXCIN; ++c) part[c] += g * s_x[hf * (PE_ROWS / 2) + r][c];
      }
    }
  }
#pragma unroll
  for (int c = 0; c <= PE_MAXCIN; ++c) s_p[hf][k][c] = part[c];
  __syncthreads();
  if (tid < hid) {
    for (int c = 0; c < cin; ++c) unsafeAtomicAdd(dw1 + tid * cin + c, s_p[0][tid][c] + s_p[1][tid][c]);
    unsafeAtomicAdd(db1 + tid, s_p[0][tid][PE_MAXCIN] + s_p[1][tid][PE_MAXCIN]);
  }
}

// backward, weight side: dW2 (cout, hid) += dY^T gelu(pre), db2 += column sums of dY.  Grid (cout / 16, row chunks):
// a workgroup owns 16 output channels x all hidden units for one chunk of rows.  dW2 / db2 must be zero on entry.
constexpr int PE_WCH = 16, PE_WROWS = 32;
__global__ __launch_bounds__(256) void pos_mlp_bwd_w_kernel(int rows, int hid, int cout, int rows_per_wg,
                                                            const float *__restrict__ pre, const float *__restrict__ dy,
                                                            float *__restrict__ dw2, float *__restrict__ db2) {
  __shared__ float s_h[PE_WROWS][PE_MAXHID + 1];
  __shared__ float s_d[PE_WROWS][PE_WCH + 1];
  const int c0 = blockIdx.x * PE_WCH, tid = threadIdx.x;
  const int ra = blockIdx.y * rows_per_wg, rb = min(rows, ra + rows_per_wg);
  // thread (k, half): hidden unit k, channels c0 + 8 half .. + 8
  const int k = tid % PE_MAXHID, hf = tid / PE_MAXHID;
  float acc[PE_WCH / 2], bsum = 0.f;
#pragma unroll
  for (int c = 0; c < PE_WCH / 2; ++c) acc[c] = 0.f;
  for (int r0 = ra; r0 < rb; r0 += PE_WROWS) {
    __syncthreads();
    for (int i = tid; i < PE_WROWS * hid; i += 256) {
      const int r = i / hid, j = i % hid;
      s_h[r][j] = (r0 + r < rb) ? sm_gelu(pre[(size_t)(r0 + r) * hid + j]) : 0.f;
    }
    for (int i = tid; i < PE_WROWS * PE_WCH; i += 256) {
      const int r = i / PE_WCH, c = i % PE_WCH;
      s_d[r][c] = (r0 + r < rb && c0 + c < cout) ? dy[(size_t)(r0 + r) * cout + c0 + c] : 0.f;
    }
    __syncthreads();
    if (k < hid) {
#pragma unroll 4
      for (int r = 0; r < PE_WROWS; ++r) {
        const float hv = s_h[r][k];
#pragma unroll
        for (int c = 0; c < PE_WCH / 2; ++c) acc[c] += s_d[r][hf * (PE_WCH / 2) + c] * hv;
      }
    }
    if (tid < PE_WCH)
      for (int r = 0; r < PE_WROWS; ++r) bsum += s_d[r][tid];
  }
  if (k < hid) {
#pragma unroll
    for (int c = 0; c < PE_WCH / 2; ++c) {
      const int ch = c0 + hf * (PE_WCH / 2) + c;
      if (ch < cout) unsafeAtomicAdd(dw2 + (size_t)ch * hid + k, acc[c]);
    }
  }
  if (tid < PE_WCH && c0 + tid < cout) unsafeAtomicAdd(db2 + c0 + tid, bsum);
}

}  // namespace

extern "C" int sig3d_pos_mlp_fwd(int rows, int cin, int hid, int cout, const float *x, const float *w1, const float *b1,
                                 const float *w2, const float *b2, const float *residual, float *pre, float *out,
                                 void *stream_) {
  SIG3D_REQUIRE(rows >= 0 && cin >= 1 && cin <= PE_MAXCIN && hid >= 4 && hid <= PE_MAXHID && hid % 4 == 0 && cout >= 1,
                "pos_mlp: 1 <= cin <= 4, hid <= 128 (multiple of 4)");
  SIG3D_REQUIRE(x && w1 && b1 && w2 && b2 && pre && out, "null argument");
  if (rows == 0) return 0;
  hipLaunchKernelGGL(pos_mlp_fwd_kernel, dim3(sig3d_ceil_div(rows, PE_ROWS)), dim3(256), 0, (hipStream_t)stream_, rows, cin,
                     hid, cout, x, w1, b1, w2, b2, residual, pre, out);
  SIG3D_LAUNCH_CHECK("pos_mlp_fwd_kernel");
  return 0;
}

extern "C" int sig3d_pos_mlp_bwd(int rows, int cin, int hid, int cout, const float *x, const float *w2, const float *pre,
                                 const float *dy, float *dpre, float *grads, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SIG3D_REQUIRE(rows >= 0 && cin >= 1 && cin <= PE_MAXCIN && hid >= 4 && hid <= PE_MAXHID && hid % 4 == 0 && cout >= 1 &&
                    cout <= 1024, "pos_mlp: 1 <= cin <= 4, hid <= 128 (multiple of 4), cout <= 1024");
  SIG3D_REQUIRE(x && w2 && pre && dy && dpre && grads, "null argument");
  // grads = [dw1 (hid*cin) | db1 (hid) | dw2 (cout*hid) | db2 (cout)]: one buffer, one memset node
  float *dw1 = grads, *db1 = dw1 + (size_t)hid * cin, *dw2 = db1 + hid, *db2 = dw2 + (size_t)cout * hid;
  SIG3D_HIP_TRY(hipMemsetAsync(grads, 0, sizeof(float) * ((size_t)hid * cin + hid + (size_t)cout * hid + cout), stream));
  if (rows == 0) return 0;
  hipLaunchKernelGGL(pos_mlp_bwd_in_kernel, dim3(sig3d_ceil_div(rows, PE_ROWS)), dim3(256),
                     sizeof(float) * PE_ROWS * (size_t)cout, stream, rows, cin, hid, cout, x, w2, pre, dy, dpre, dw1, db1);
  int chunks = sig3d_ceil_div(rows, 256);
  if (chunks > 16) chunks = 16;
  const int rows_per_wg = sig3d_ceil_div(rows, chunks);
  hipLaunchKernelGGL(pos_mlp_bwd_w_kernel, dim3(sig3d_ceil_div(cout, PE_WCH), chunks), dim3(256), 0, stream, rows, hid, cout,
                     rows_per_wg, pre, dy, dw2, db2);
  SIG3D_LAUNCH_CHECK("pos_mlp_bwd kernels");
  return 0;
}
