// small_mlp.hip -- the small dense layers around the Q-Former as single launches.
//
// (1) The positional MLP of the scene tokens, sqa_module.py:274-278 / :319-321:
//         tokens = tok_feat + Linear(hid, cout)(GELU(Linear(cin, hid)(pos)))        cin = 2 or 3, hid = 128, cout = 256
//     torch runs it as addmm + gelu + addmm + add (4 launches) and ~12 in the backward pass (85 us for 2048 rows).
//     Here: one forward launch (a workgroup owns 16 token rows: hidden activations in LDS, a thread per output
//     channel), and two backward launches (input-side: dH = dY W2, dPre, dW1, db1; weight-side: dW2 = dY^T H, db2
//     with H recomputed from the stored pre-activation).
#include "sig3d_common.h"
#include "situational_pose.h"

namespace {

__device__ __forceinline__ float sm_gelu(float u) { return 0.5f * u * (1.f + erff(u * 0.70710678118654752440f)); }
__device__ __forceinline__ float sm_gelu_grad(float u) {
  const float cdf = 0.5f * (1.f + erff(u * 0.70710678118654752440f));
  return cdf + u * 0.39894228040143267794f * __expf(-0.5f * u * u);
}

// All three launches are 64 x 64 output tiles of a small f32 product on the vector ALUs (the shapes are far below
// anything a matrix-core tile pays for): 256 threads, a 4 x 4 micro-tile per thread, operands staged k-major in LDS
// 128 reduction steps at a time -- the whole reduction of the forward and weight-side products in ONE staging, so a
// workgroup pays one global round trip (staging 16 steps at a time made every step a round trip: 21 us for 67 MFLOP)
// -- with everything elementwise folded into the staging / the epilogue.
constexpr int PE_MAXHID = 128;
constexpr int PE_MAXCIN = 4;
constexpr int SG_T = 64, SG_K = 128, SG_LD = SG_T + 4;

struct SgAcc { float v[4][4]; };

__device__ __forceinline__ void sg_zero(SgAcc &a) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) a.v[i][j] = 0.f;
}
// acc += As^T Bs over `steps` staged steps; thread (ty, tx) owns rows 4 ty.., columns 4 tx..
__device__ __forceinline__ void sg_steps(SgAcc &a, const float (*As)[SG_LD], const float (*Bs)[SG_LD], int ty, int tx,
                                         int steps) {
#pragma unroll 8
  for (int k = 0; k < steps; ++k) {
    const float4 av = *reinterpret_cast<const float4 *>(&As[k][4 * ty]);
    const float4 bv = *reinterpret_cast<const float4 *>(&Bs[k][4 * tx]);
    const float ar[4] = {av.x, av.y, av.z, av.w}, br[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) a.v[i][j] = __builtin_fmaf(ar[i], br[j], a.v[i][j]);   // (the build runs -ffp-contract=off)
  }
}

// forward: out[m][n] = residual + b2[n] + sum_k gelu(pre[m][k]) w2[n][k],  pre[m][k] = b1[k] + sum_c w1[k][c] x[m][c]
// grid (row tiles, column tiles); column tile 0 also writes pre (kept for the backward pass).  hid <= 128: one staging.
// pose != null (cin == 3): x is not read but FORMED here -- the situational re-encode of the token positions,
// x[m] = R(q_b)^T (points[m] - t_b), b = m / tokens (situational_pose.h: the arithmetic of sig3d_situational_transform,
// bit for bit) -- and written to x_out by the first column tile: the transform rides in this launch (temp.py:86-97 +
// sqa_module.py:274-278, 319-321 as one kernel).
__global__ __launch_bounds__(256) void pos_mlp_fwd_kernel(int rows, int cin, int hid, int cout, int tokens, int inverse,
                                                          const float *__restrict__ pose, const float *__restrict__ points,
                                                          float *__restrict__ x_out,
                                                          const float *__restrict__ x, const float *__restrict__ w1,
                                                          const float *__restrict__ b1, const float *__restrict__ w2,
                                                          const float *__restrict__ b2, const float *__restrict__ residual,
                                                          float *__restrict__ pre, float *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float sg_smem[];
  float(*As)[SG_LD] = reinterpret_cast<float(*)[SG_LD]>(sg_smem);
  float(*Bs)[SG_LD] = As + SG_K;
  __shared__ float s_x[SG_T][PE_MAXCIN], s_w1[PE_MAXHID][PE_MAXCIN + 1];
  const int m0 = blockIdx.x * SG_T, n0 = blockIdx.y * SG_T, tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
  // Bs[k][n] = w2[n0 + n][k]: requested first (a row of 32 consecutive k per thread), consumed after the x / w1 phase
  const int rn = tid >> 2, kq = (tid & 3) * 32;
  float4 wv[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
    wv[i] = (n0 + rn < cout && kq + 4 * i < hid) ? *reinterpret_cast<const float4 *>(w2 + (size_t)(n0 + rn) * hid + kq + 4 * i)
                                                 : make_float4(0.f, 0.f, 0.f, 0.f);
  if (pose != nullptr) {
    if (tid < SG_T) {
      const int m = m0 + tid;
      float o[3] = {0.f, 0.f, 0.f};
      if (m < rows) {
        const Pose P = load_pose(pose + (size_t)(m / tokens) * 7);
        const float *pt = points + (size_t)m * 3;
        if (inverse) apply_pose<true>(P, pt[0], pt[1], pt[2], o);
        else apply_pose<false>(P, pt[0], pt[1], pt[2], o);
        if (blockIdx.y == 0) { x_out[(size_t)m * 3 + 0] = o[0]; x_out[(size_t)m * 3 + 1] = o[1]; x_out[(size_t)m * 3 + 2] = o[2]; }
      }
      s_x[tid][0] = o[0]; s_x[tid][1] = o[1]; s_x[tid][2] = o[2]; s_x[tid][3] = 0.f;
    }
  } else {
    for (int i = tid; i < SG_T * PE_MAXCIN; i += 256) {
      const int r = i / PE_MAXCIN, c = i % PE_MAXCIN;
      s_x[r][c] = (m0 + r < rows && c < cin) ? x[(size_t)(m0 + r) * cin + c] : 0.f;
    }
  }
  for (int i = tid; i < hid * (PE_MAXCIN + 1); i += 256) {
    const int k = i / (PE_MAXCIN + 1), c = i % (PE_MAXCIN + 1);
    s_w1[k][c] = c < cin ? w1[k * cin + c] : (c == PE_MAXCIN ? b1[k] : 0.f);
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    Bs[kq + 4 * i + 0][rn] = wv[i].x; Bs[kq + 4 * i + 1][rn] = wv[i].y;
    Bs[kq + 4 * i + 2][rn] = wv[i].z; Bs[kq + 4 * i + 3][rn] = wv[i].w;
  }
  {   // As[k][m] = gelu(pre[m][k]): thread -> row m = tid / 4, 32 consecutive k
    const int m = rn;
#pragma unroll 4
    for (int i = 0; i < 32; i += 4) {
      float u[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k = kq + i + e;
        u[e] = 0.f;
        if (k < hid) {
          u[e] = s_w1[k][PE_MAXCIN];
#pragma unroll
          for (int c = 0; c < PE_MAXCIN; ++c) u[e] = __builtin_fmaf(s_w1[k][c], s_x[m][c], u[e]);
        }
        As[k][m] = k < hid ? sm_gelu(u[e]) : 0.f;
      }
      if (blockIdx.y == 0 && m0 + m < rows && kq + i < hid)
        *reinterpret_cast<float4 *>(pre + (size_t)(m0 + m) * hid + kq + i) = make_float4(u[0], u[1], u[2], u[3]);
    }
  }
  __syncthreads();
  SgAcc acc;
  sg_zero(acc);
  sg_steps(acc, As, Bs, ty, tx, hid);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + 4 * ty + i;
    if (m >= rows || n0 + 4 * tx >= cout) continue;
    const size_t o = (size_t)m * cout + n0 + 4 * tx;
    const float4 bb = *reinterpret_cast<const float4 *>(b2 + n0 + 4 * tx);
    float4 rv = residual ? *reinterpret_cast<const float4 *>(residual + o) : make_float4(0.f, 0.f, 0.f, 0.f);
    rv.x += bb.x + acc.v[i][0]; rv.y += bb.y + acc.v[i][1]; rv.z += bb.z + acc.v[i][2]; rv.w += bb.w + acc.v[i][3];
    *reinterpret_cast<float4 *>(out + o) = rv;
  }
}

// backward, input side: dpre[m][j] = gelu'(pre[m][j]) * sum_c dy[m][c] w2[c][j];  dw1[j][:] += dpre[:, j]^T x,
// db1[j] += sum_m dpre[m][j] (float atomics: dw1 / db1 zero on entry).  grid (row tiles, hid tiles)
__device__ __forceinline__ void pos_mlp_bwd_in_body(int bx, int by, int rows, int cin, int hid, int cout,
                                                    const float *__restrict__ x, const float *__restrict__ w2,
                                                    const float *__restrict__ pre, const float *__restrict__ dy,
                                                    float *__restrict__ dpre, float *__restrict__ dw1,
                                                    float *__restrict__ db1) {
  extern __shared__ __attribute__((aligned(16))) float sg_smem[];
  float(*As)[SG_LD] = reinterpret_cast<float(*)[SG_LD]>(sg_smem);
  float(*Bs)[SG_LD] = As + SG_K;
  __shared__ float s_x[SG_T][PE_MAXCIN], s_p[16][SG_T][PE_MAXCIN + 1];
  const int m0 = bx * SG_T, n0 = by * SG_T, tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
  for (int i = tid; i < SG_T * PE_MAXCIN; i += 256) {
    const int r = i / PE_MAXCIN, c = i % PE_MAXCIN;
    s_x[r][c] = (m0 + r < rows && c < cin) ? x[(size_t)(m0 + r) * cin + c] : 0.f;
  }
  SgAcc acc;
  sg_zero(acc);
  const int am = tid >> 2, akq = (tid & 3) * 32;    // As[k][m] = dy[m0 + m][k0 + k]: 32 consecutive k of a row
  const int bk = tid >> 1, bnq = (tid & 1) * 32;    // Bs[k][n] = w2[k0 + k][n0 + n]: 32 consecutive n of a row
  for (int k0 = 0; k0 < cout; k0 += SG_K) {
    float4 dv[8], wv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      dv[i] = (m0 + am < rows && k0 + akq + 4 * i < cout)
                  ? *reinterpret_cast<const float4 *>(dy + (size_t)(m0 + am) * cout + k0 + akq + 4 * i)
                  : make_float4(0.f, 0.f, 0.f, 0.f);
      wv[i] = (k0 + bk < cout && n0 + bnq + 4 * i < hid)
                  ? *reinterpret_cast<const float4 *>(w2 + (size_t)(k0 + bk) * hid + n0 + bnq + 4 * i)
                  : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();   // the previous chunk has been consumed
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      As[akq + 4 * i + 0][am] = dv[i].x; As[akq + 4 * i + 1][am] = dv[i].y;
      As[akq + 4 * i + 2][am] = dv[i].z; As[akq + 4 * i + 3][am] = dv[i].w;
      *reinterpret_cast<float4 *>(&Bs[bk][bnq + 4 * i]) = wv[i];
    }
    __syncthreads();
    sg_steps(acc, As, Bs, ty, tx, min(SG_K, cout - k0));
  }
  float part[4][PE_MAXCIN + 1];   // this thread's four hidden units: sums over its four rows
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int c = 0; c <= PE_MAXCIN; ++c) part[j][c] = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ml = 4 * ty + i, m = m0 + ml;
    if (m >= rows || n0 + 4 * tx >= hid) continue;
    const float4 pv = *reinterpret_cast<const float4 *>(pre + (size_t)m * hid + n0 + 4 * tx);
    const float pr[4] = {pv.x, pv.y, pv.z, pv.w};
    float g[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      g[j] = acc.v[i][j] * sm_gelu_grad(pr[j]);
      part[j][PE_MAXCIN] += g[j];
#pragma unroll
      for (int c = 0; c < PE_MAXCIN; ++c) part[j][c] += g[j] * s_x[ml][c];
    }
    *reinterpret_cast<float4 *>(dpre + (size_t)m * hid + n0 + 4 * tx) = make_float4(g[0], g[1], g[2], g[3]);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int c = 0; c <= PE_MAXCIN; ++c) s_p[ty][4 * tx + j][c] = part[j][c];
  __syncthreads();
  for (int i = tid; i < SG_T * (PE_MAXCIN + 1); i += 256) {
    const int nl = i / (PE_MAXCIN + 1), c = i % (PE_MAXCIN + 1), n = n0 + nl;
    if (n >= hid || (c < PE_MAXCIN && c >= cin)) continue;
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) t += s_p[r][nl][c];
    if (c == PE_MAXCIN) unsafeAtomicAdd(db1 + n, t);
    else unsafeAtomicAdd(dw1 + n * cin + c, t);
  }
}

// backward, weight side: dw2[c][j] += sum_m dy[m][c] gelu(pre[m][j]) over this workgroup's 128 rows, db2[c] += sum_m
// dy[m][c] (float atomics: zero on entry).  grid (cout tiles, hid tiles, chunks of 128 rows)
__device__ __forceinline__ void pos_mlp_bwd_w_body(int bx, int by, int bz, int rows, int hid, int cout,
                                                   const float *__restrict__ pre, const float *__restrict__ dy,
                                                   float *__restrict__ dw2, float *__restrict__ db2) {
  extern __shared__ __attribute__((aligned(16))) float sg_smem[];
  float(*As)[SG_LD] = reinterpret_cast<float(*)[SG_LD]>(sg_smem);
  float(*Bs)[SG_LD] = As + SG_K;
  const int m0 = bx * SG_T, n0 = by * SG_T, tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
  const int r0 = bz * SG_K;
  const int k = tid >> 1, q = (tid & 1) * 32;   // row r0 + k, 32 consecutive columns
  const bool live = r0 + k < rows;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float4 dv = make_float4(0.f, 0.f, 0.f, 0.f), pv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live && m0 + q + 4 * i < cout) dv = *reinterpret_cast<const float4 *>(dy + (size_t)(r0 + k) * cout + m0 + q + 4 * i);
    if (live && n0 + q + 4 * i < hid) {
      pv = *reinterpret_cast<const float4 *>(pre + (size_t)(r0 + k) * hid + n0 + q + 4 * i);
      pv = make_float4(sm_gelu(pv.x), sm_gelu(pv.y), sm_gelu(pv.z), sm_gelu(pv.w));
    }
    *reinterpret_cast<float4 *>(&As[k][q + 4 * i]) = dv;      // As[k][m] = dy[r0 + k][m0 + m]
    *reinterpret_cast<float4 *>(&Bs[k][q + 4 * i]) = pv;      // Bs[k][n] = gelu(pre[r0 + k][n0 + n])
  }
  __syncthreads();
  SgAcc acc;
  sg_zero(acc);
  sg_steps(acc, As, Bs, ty, tx, SG_K);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = m0 + 4 * ty + i;
    if (c >= cout) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + 4 * tx + j;
      if (n < hid) unsafeAtomicAdd(dw2 + (size_t)c * hid + n, acc.v[i][j]);
    }
  }
  if (by == 0 && tid < SG_T && m0 + tid < cout) {   // bias gradient once per (cout tile, row chunk)
    float t = 0.f;
#pragma unroll 8
    for (int r = 0; r < SG_K; ++r) t += As[r][tid];
    unsafeAtomicAdd(db2 + m0 + tid, t);
  }
}

}  // namespace

static int pos_mlp_fwd_impl(int rows, int cin, int hid, int cout, int tokens, int inverse, const float *pose,
                            const float *points, float *x_out, const float *x, const float *w1, const float *b1,
                            const float *w2, const float *b2, const float *residual, float *pre, float *out, void *stream_) {
  SIG3D_REQUIRE(rows >= 0 && cin >= 1 && cin <= PE_MAXCIN && hid >= 4 && hid <= PE_MAXHID && hid % 4 == 0 && cout >= 4 && cout % 4 == 0,
                "pos_mlp: 1 <= cin <= 4, hid <= 128, hid and cout multiples of 4");
  SIG3D_REQUIRE((x || pose) && w1 && b1 && w2 && b2 && pre && out, "null argument");
  if (rows == 0) return 0;
  const size_t lds = sizeof(float) * 2 * SG_K * SG_LD;   // 68 KB of operand tiles
  SIG3D_HIP_TRY(hipFuncSetAttribute((const void *)pos_mlp_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(pos_mlp_fwd_kernel, dim3(sig3d_ceil_div(rows, SG_T), sig3d_ceil_div(cout, SG_T)), dim3(256), lds,
                     (hipStream_t)stream_, rows, cin, hid, cout, tokens, inverse, pose, points, x_out, x, w1, b1, w2, b2,
                     residual, pre, out);
  SIG3D_LAUNCH_CHECK("pos_mlp_fwd_kernel");
  return 0;
}

extern "C" int sig3d_pos_mlp_fwd(int rows, int cin, int hid, int cout, const float *x, const float *w1, const float *b1,
                                 const float *w2, const float *b2, const float *residual, float *pre, float *out,
                                 void *stream_) {
  SIG3D_REQUIRE(x != nullptr, "null argument");
  return pos_mlp_fwd_impl(rows, cin, hid, cout, 1, 0, nullptr, nullptr, nullptr, x, w1, b1, w2, b2, residual, pre, out, stream_);
}

// The situational re-encode folded in: x = R(q)^T (points - t) (inverse != 0) or R(q) points + t, formed from pose
// (b, 7) and points (b * tokens, 3) inside the launch and written to x_out (b * tokens, 3); then as sig3d_pos_mlp_fwd.
extern "C" int sig3d_pos_mlp_fwd_posed(int b, int tokens, int hid, int cout, int inverse, const float *pose,
                                       const float *points, float *x_out, const float *w1, const float *b1,
                                       const float *w2, const float *b2, const float *residual, float *pre, float *out,
                                       void *stream_) {
  SIG3D_REQUIRE(b >= 0 && tokens >= 1 && pose && points && x_out, "bad arguments");
  return pos_mlp_fwd_impl(b * tokens, 3, hid, cout, tokens, inverse, pose, points, x_out, nullptr, w1, b1, w2, b2, residual,
                          pre, out, stream_);
}

// Both sides of the backward pass in ONE launch: they read dy and pre and share nothing they write, and each is a few
// dozen workgroups of ~20 us -- in a row they cost the step two such waits.  Workgroups [0, n_in) are the input side's
// (row tile, hid tile) grid, the rest the weight side's (cout tile, hid tile, row chunk) grid.
__global__ __launch_bounds__(256) void pos_mlp_bwd_kernel(int rows, int cin, int hid, int cout, int in_x, int n_in, int w_x,
                                                          int w_y, const float *__restrict__ x,
                                                          const float *__restrict__ w2, const float *__restrict__ pre,
                                                          const float *__restrict__ dy, float *__restrict__ dpre,
                                                          float *__restrict__ dw1, float *__restrict__ db1,
                                                          float *__restrict__ dw2, float *__restrict__ db2) {
  const int blk = (int)blockIdx.x;       // (uniform branch: a workgroup runs one body)
  if (blk < n_in) {
    pos_mlp_bwd_in_body(blk % in_x, blk / in_x, rows, cin, hid, cout, x, w2, pre, dy, dpre, dw1, db1);
  } else {
    const int t = blk - n_in;
    pos_mlp_bwd_w_body(t % w_x, (t / w_x) % w_y, t / (w_x * w_y), rows, hid, cout, pre, dy, dw2, db2);
  }
}

static int pos_mlp_bwd_impl(int rows, int cin, int hid, int cout, const float *x, const float *w2, const float *pre,
                            const float *dy, float *dpre, float *grads, bool grads_zeroed, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SIG3D_REQUIRE(rows >= 0 && cin >= 1 && cin <= PE_MAXCIN && hid >= 4 && hid <= PE_MAXHID && hid % 4 == 0 && cout >= 4 &&
                    cout % 4 == 0, "pos_mlp: 1 <= cin <= 4, hid <= 128, hid and cout multiples of 4");
  SIG3D_REQUIRE(x && w2 && pre && dy && dpre && grads, "null argument");
  // grads = [dw1 (hid*cin) | db1 (hid) | dw2 (cout*hid) | db2 (cout)]: one buffer, one memset node
  float *dw1 = grads, *db1 = dw1 + (size_t)hid * cin, *dw2 = db1 + hid, *db2 = dw2 + (size_t)cout * hid;
  if (!grads_zeroed)
    SIG3D_HIP_TRY(hipMemsetAsync(grads, 0, sizeof(float) * ((size_t)hid * cin + hid + (size_t)cout * hid + cout), stream));
  if (rows == 0) return 0;
  const size_t lds = sizeof(float) * 2 * SG_K * SG_LD;
  SIG3D_HIP_TRY(hipFuncSetAttribute((const void *)pos_mlp_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int in_x = sig3d_ceil_div(rows, SG_T), n_in = in_x * sig3d_ceil_div(hid, SG_T);
  const int w_x = sig3d_ceil_div(cout, SG_T), w_y = sig3d_ceil_div(hid, SG_T), w_z = sig3d_ceil_div(rows, SG_K);
  hipLaunchKernelGGL(pos_mlp_bwd_kernel, dim3(n_in + w_x * w_y * w_z), dim3(256), lds, stream, rows, cin, hid, cout, in_x,
                     n_in, w_x, w_y, x, w2, pre, dy, dpre, dw1, db1, dw2, db2);
  SIG3D_LAUNCH_CHECK("pos_mlp_bwd kernels");
  return 0;
}

extern "C" int sig3d_pos_mlp_bwd(int rows, int cin, int hid, int cout, const float *x, const float *w2, const float *pre,
                                 const float *dy, float *dpre, float *grads, void *stream_) {
  return pos_mlp_bwd_impl(rows, cin, hid, cout, x, w2, pre, dy, dpre, grads, false, stream_);
}

// grads zeroed by the caller (scratch.py: one fill per step)
extern "C" int sig3d_pos_mlp_bwd_z(int rows, int cin, int hid, int cout, const float *x, const float *w2, const float *pre,
                                   const float *dy, float *dpre, float *grads, void *stream_) {
  return pos_mlp_bwd_impl(rows, cin, hid, cout, x, w2, pre, dy, dpre, grads, true, stream_);
}
