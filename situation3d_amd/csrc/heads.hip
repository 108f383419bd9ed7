// The two MLP heads on the pooled Q-Former output as six launches instead of thirty-one.
//
//   pooled = mean over the 32 query rows of a sample                        (sqa_module.py:  fuse_feat pooling)
//   aux    = Linear(H,7)(GELU(Linear(H,H)(pooled)))                         (aux_reg)
//   answer = Linear(H,A)(Dropout(GELU(Linear(H,H)(pooled))))                (answer_cls)
//
// B = 8 rows: every product is a handful of matrix-VECTOR products, bound by reading each weight once (7 MB forward,
// 7 MB + 7 MB of weight gradients backward).  torch runs them as 4 + 8 library GEMMs of 3-11 us (16x16 tiles on a few
// dozen workgroups) with 19 elementwise / reduction launches between them: ~120 us per step.  Here:
//   forward   pool_rows (1) -> rows_linear_fwd (both first layers: bias + GELU (+ dropout)) -> rows_linear_fwd (both
//             second layers)
//   backward  rows_linear_bwd (second layers: dW, db, partial d hidden) -> rows_linear_bwd (first layers: d pre =
//             sum(partials) * gelu' * dropout formed on load; dW, db, partial d pooled) -> spread_pooled_grad (both
//             heads' partials summed and written to the query rows)
// Sums: a row's dot product is split over the 64 lanes of a wave and folded by a butterfly; gradients wrt the inputs
// are folded over 16 row groups of a workgroup and 8 workgroups -- fixed orders, no atomics, deterministic.
#include "sig3d_common.h"

namespace {

constexpr int HD_THREADS = 256;

__device__ __forceinline__ float hd_gelu(float u) { return 0.5f * u * (1.f + erff(u * 0.70710678118654752440f)); }
__device__ __forceinline__ float hd_gelu_grad(float u) {
  return 0.5f * (1.f + erff(u * 0.70710678118654752440f)) + u * 0.39894228040143267794f * __expf(-0.5f * u * u);
}
__device__ __forceinline__ unsigned hd_mix32(unsigned x) {
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  return x;
}
// keep bit of element idx of dropout call `call_id` (same construction as rowops.hip)
__device__ __forceinline__ bool hd_keep(unsigned seed, unsigned idx, unsigned thresh) {
  return hd_mix32(seed ^ idx * 0x9E3779B9u) >= thresh;
}

// grid (hidden / 256 columns... see launch): a thread sums every 4th row of one float4 column; the four partial
// sums of a column meet in LDS
__global__ __launch_bounds__(HD_THREADS) void pool_rows_kernel(int q, int cols, const float *__restrict__ rows,
                                                               float *__restrict__ pooled) {
  __shared__ float4 s_part[4][64];
  const int c4 = blockIdx.x * 64 + (threadIdx.x & 63), qg = threadIdx.x >> 6, b = blockIdx.y;
  const int ncol4 = cols / 4;
  const float4 *p = reinterpret_cast<const float4 *>(rows + (size_t)b * q * cols) + min(c4, ncol4 - 1);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int i0 = qg; i0 < q; i0 += 32) {   // eight requests in flight (clamped row, masked afterwards)
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = p[(size_t)min(i0 + 4 * u, q - 1) * ncol4];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (i0 + 4 * u < q) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
  }
  s_part[qg][threadIdx.x & 63] = s;
  __syncthreads();
  if (qg == 0 && c4 < ncol4) {
    const float4 a = s_part[0][threadIdx.x], b1 = s_part[1][threadIdx.x], c = s_part[2][threadIdx.x], d = s_part[3][threadIdx.x];
    const float inv = 1.f / (float)q;
    reinterpret_cast<float4 *>(pooled + (size_t)b * cols)[c4] =
        make_float4(((a.x + b1.x) + (c.x + d.x)) * inv, ((a.y + b1.y) + (c.y + d.y)) * inv,
                    ((a.z + b1.z) + (c.z + d.z)) * inv, ((a.w + b1.w) + (c.w + d.w)) * inv);
  }
}

struct HdFwdSeg {
  const float *w;      // (n_out, k)
  const float *bias;   // (n_out)
  const float *x;      // (rows, k)
  float *pre;          // (rows, n_out) pre-activation, or NULL
  float *out;          // (rows, n_out)
  int n_out;
  int act;             // 0: none, 1: GELU, 2: GELU then dropout
  unsigned call_id;
};

// 8 outputs per workgroup (2 per wave), all rows at once; x of the workgroup's segment staged in LDS
template <int ROWS>
__global__ __launch_bounds__(HD_THREADS) void rows_linear_fwd_kernel(HdFwdSeg s0, HdFwdSeg s1, int rows, int k, float p_drop,
                                                                     const unsigned *__restrict__ rng_counter) {
  extern __shared__ __attribute__((aligned(16))) float s_x[];   // [rows][k]
  const int nb0 = (s0.n_out + 7) / 8;
  const bool second = (int)blockIdx.x >= nb0;
  const HdFwdSeg s = second ? s1 : s0;
  const int o_base = ((int)blockIdx.x - (second ? nb0 : 0)) * 8;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int o0 = o_base + wave * 2;
  const float *w0 = s.w + (size_t)min(o0, s.n_out - 1) * k, *w1 = s.w + (size_t)min(o0 + 1, s.n_out - 1) * k;
  // the weight rows (HBM) are requested before the inputs are staged: k <= 1024 -> at most 4 float4 per lane and row
  float4 wa[4], wb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = min(lane * 4 + 256 * i, k - 4);
    wa[i] = *reinterpret_cast<const float4 *>(w0 + c);
    wb[i] = *reinterpret_cast<const float4 *>(w1 + c);
  }
  for (int i = threadIdx.x; i < rows * k / 4; i += HD_THREADS)
    reinterpret_cast<float4 *>(s_x)[i] = reinterpret_cast<const float4 *>(s.x)[i];
  __syncthreads();
  float acc[2][ROWS];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < ROWS; ++r) acc[j][r] = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane * 4 + 256 * i;
    if (c >= k) break;
    const float4 a = wa[i], b = wb[i];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      if (r < rows) {
        const float4 xv = *reinterpret_cast<const float4 *>(s_x + r * k + c);
        acc[0][r] += a.x * xv.x + a.y * xv.y + a.z * xv.z + a.w * xv.w;
        acc[1][r] += b.x * xv.x + b.y * xv.y + b.z * xv.z + b.w * xv.w;
      }
    }
  }
  float mine = 0.f;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      float v = acc[j][r];
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
      if (lane == j * ROWS + r) mine = v;
    }
  if (lane < 2 * ROWS) {
    const int j = lane / ROWS, r = lane % ROWS, o = o0 + j;
    if (r < rows && o < s.n_out) {
      const float u = mine + s.bias[o];
      const size_t at = (size_t)r * s.n_out + o;
      if (s.pre) s.pre[at] = u;
      float y = u;
      if (s.act >= 1) y = hd_gelu(u);
      if (s.act == 2 && p_drop > 0.f) {
        const unsigned seed = hd_mix32((rng_counter ? *rng_counter : 0u) * 0x9E3779B9u + s.call_id);
        const unsigned thresh = (unsigned)((double)p_drop * 4294967296.0);
        y = hd_keep(seed, (unsigned)at, thresh) ? y * (1.f / (1.f - p_drop)) : 0.f;
      }
      s.out[at] = y;
    }
  }
}

struct HdBwdSeg {
  const float *w;       // (n_out, k) the layer's weight
  const float *dy;      // (rows, n_out) gradient wrt the layer's output, or NULL: then it is formed while loading from
  const float *dyp;     //   (n_parts, rows, n_out) partial sums of the gradient wrt the layer's ACTIVATED output,
  const float *pre;     //   (rows, n_out) the layer's pre-activation and
  int n_parts, act;     //   its activation (1 GELU, 2 GELU then dropout): dy = sum(parts) * gelu'(pre) * keep / (1 - p)
  unsigned call_id;
  const float *x;       // (rows, k) the layer's input
  float *dw;            // (n_out, k)
  float *db;            // (n_out)
  float *dxp;           // (HD_OS, rows, k) partial sums of the gradient wrt x, one per range of output rows
  int n_out;
};

constexpr int HD_OS = 8;     // ranges the output rows of a layer are dealt into (workgroups per 16 input columns)
__host__ __device__ inline int hd_chunk(int n_out) {
  const int c = ((n_out + HD_OS - 1) / HD_OS + 15) / 16 * 16;
  return c < 16 ? 16 : c;
}
__host__ __device__ inline int hd_parts(int n_out) { return (n_out + hd_chunk(n_out) - 1) / hd_chunk(n_out); }

// One workgroup: 16 input columns x one range of output rows of one layer; 16 column lanes x 16 output-row groups.
// dW[o][k] = sum_r dy[r][o] x[r][k] is written as it is formed; the range's share of dx[r][k] = sum_o dy[r][o] W[o][k] is
// folded over the row groups in LDS and written as a partial (the consumer adds the hd_parts(n_out) partials).
// Memory-latency bound (every weight element is read once, by one thread): 768 workgroups keep ~5 requests per thread
// in flight; the first version (one workgroup per 16 columns, 44 dependent trips) took 21 + 35 us.
template <int ROWS>
__global__ __launch_bounds__(HD_THREADS) void rows_linear_bwd_kernel(HdBwdSeg s0, HdBwdSeg s1, int rows, int k, float p_drop,
                                                                     const unsigned *__restrict__ rng_counter) {
  __shared__ __attribute__((aligned(16))) float s_mem[16 * ROWS * 16];   // dy [chunk <= 128][ROWS], then the fold
  const int kblocks = k / 16;
  const int seg = blockIdx.y, os = blockIdx.z, kb = blockIdx.x;
  const HdBwdSeg s = seg ? s1 : s0;
  const int chunk = hd_chunk(s.n_out);
  const int o_lo = os * chunk, o_hi = min(s.n_out, o_lo + chunk);
  if (o_lo >= s.n_out) return;
  (void)kblocks;
  const int kl = threadIdx.x & 15, og = threadIdx.x >> 4;
  const int kcol = kb * 16 + kl;
  // weights of the first trip and the inputs first (HBM), then dy of the range -> LDS
  float wv[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) wv[u] = s.w[(size_t)min(o_lo + og + 16 * u, s.n_out - 1) * k + kcol];
  float xr[ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) xr[r] = r < rows ? s.x[(size_t)r * k + kcol] : 0.f;
  float *s_dy = s_mem;
  const int n_rng = o_hi - o_lo;
  for (int i = threadIdx.x; i < ROWS * n_rng; i += HD_THREADS) {
    const int r = i / n_rng, j = i - r * n_rng, o = o_lo + j;
    float v = 0.f;
    if (r < rows) {
      const size_t at = (size_t)r * s.n_out + o;
      if (s.dy) {
        v = s.dy[at];
      } else {
        float t[HD_OS];   // all partials requested together (clamped, masked): a counted loop waits for each
#pragma unroll
        for (int pi = 0; pi < HD_OS; ++pi) t[pi] = s.dyp[(size_t)min(pi, s.n_parts - 1) * rows * s.n_out + at];
#pragma unroll
        for (int pi = 0; pi < HD_OS; ++pi) v += pi < s.n_parts ? t[pi] : 0.f;
        v *= hd_gelu_grad(s.pre[at]);
        if (s.act == 2 && p_drop > 0.f) {
          const unsigned seed = hd_mix32((rng_counter ? *rng_counter : 0u) * 0x9E3779B9u + s.call_id);
          const unsigned thresh = (unsigned)((double)p_drop * 4294967296.0);
          v = hd_keep(seed, (unsigned)at, thresh) ? v * (1.f / (1.f - p_drop)) : 0.f;
        }
      }
    }
    s_dy[j * ROWS + r] = v;
  }
  __syncthreads();
  float acc[ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) acc[r] = 0.f;
  for (int o_ = o_lo + og; o_ < o_hi; o_ += 64) {
    float wn[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) wn[u] = s.w[(size_t)min(o_ + 64 + 16 * u, s.n_out - 1) * k + kcol];   // next trip
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int o = o_ + 16 * u;
      if (o < o_hi) {
        float dw = 0.f, sum = 0.f;
#pragma unroll
        for (int r4 = 0; r4 < ROWS; r4 += 4) {
          const float4 d = *reinterpret_cast<const float4 *>(s_dy + (o - o_lo) * ROWS + r4);
          dw += d.x * xr[r4] + d.y * xr[r4 + 1] + d.z * xr[r4 + 2] + d.w * xr[r4 + 3];
          acc[r4] += d.x * wv[u]; acc[r4 + 1] += d.y * wv[u]; acc[r4 + 2] += d.z * wv[u]; acc[r4 + 3] += d.w * wv[u];
          sum += d.x + d.y + d.z + d.w;
        }
        s.dw[(size_t)o * k + kcol] = dw;
        if (kb == 0 && kl == 0) s.db[o] = sum;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) wv[u] = wn[u];
  }
  // fold the 16 row groups
  __syncthreads();
  float *s_red = s_mem;   // [16][ROWS][16]
#pragma unroll
  for (int r = 0; r < ROWS; ++r) s_red[(og * ROWS + r) * 16 + kl] = acc[r];
  __syncthreads();
  if ((int)threadIdx.x < ROWS * 16) {
    const int r = threadIdx.x >> 4, c = threadIdx.x & 15;
    if (r < rows) {
      float v = 0.f;
#pragma unroll
      for (int g = 0; g < 16; ++g) v += s_red[(g * ROWS + r) * 16 + c];
      s.dxp[((size_t)os * rows + r) * k + kb * 16 + c] = v;
    }
  }
}

// d rows: the partial gradients of the pooled rows (both heads, all ranges) summed, divided by q, into each of the q
// query rows of the sample
__global__ __launch_bounds__(HD_THREADS) void spread_pooled_grad_kernel(int rows, int k, int q, const float *__restrict__ p0,
                                                                        int n0, const float *__restrict__ p1, int n1,
                                                                        float *__restrict__ drows) {
  const int i = blockIdx.x * HD_THREADS + threadIdx.x;
  if (i >= rows * k) return;
  float v = 0.f, t0[HD_OS], t1[HD_OS];
#pragma unroll
  for (int j = 0; j < HD_OS; ++j) {
    t0[j] = p0[(size_t)min(j, n0 - 1) * rows * k + i];
    t1[j] = p1[(size_t)min(j, n1 - 1) * rows * k + i];
  }
#pragma unroll
  for (int j = 0; j < HD_OS; ++j) v += j < n0 ? t0[j] : 0.f;
#pragma unroll
  for (int j = 0; j < HD_OS; ++j) v += j < n1 ? t1[j] : 0.f;
  v /= (float)q;
  const int r = i / k, c = i - r * k;
  float *o = drows + (size_t)r * q * k + c;
  for (int j = 0; j < q; ++j) o[(size_t)j * k] = v;
}

template <int ROWS>
int launch_fwd(HdFwdSeg a, HdFwdSeg c, int rows, int k, float p_drop, const unsigned *rng, hipStream_t stream) {
  const size_t lds = sizeof(float) * (size_t)rows * k;
  const int grid = (a.n_out + 7) / 8 + (c.n_out + 7) / 8;
  hipLaunchKernelGGL((rows_linear_fwd_kernel<ROWS>), dim3(grid), dim3(HD_THREADS), lds, stream, a, c, rows, k, p_drop, rng);
  return 0;
}

template <int ROWS>
int launch_bwd(HdBwdSeg a, HdBwdSeg c, int rows, int k, float p_drop, const unsigned *rng, hipStream_t stream) {
  hipLaunchKernelGGL((rows_linear_bwd_kernel<ROWS>), dim3(k / 16, 2, HD_OS), dim3(HD_THREADS), 0, stream, a, c, rows, k,
                     p_drop, rng);
  return 0;
}

}  // namespace

extern "C" int sig3d_pooled_heads_fwd(int b, int q, int hidden, int n_aux, int n_ans, const float *rows,
                                      const float *w1a, const float *b1a, const float *w2a, const float *b2a,
                                      const float *w1c, const float *b1c, const float *w2c, const float *b2c,
                                      float p_drop, unsigned call_id, const unsigned *rng_counter, float *pooled,
                                      float *pre, float *h, float *aux, float *ans, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SIG3D_REQUIRE(b >= 0 && b <= 16 && q >= 1 && n_aux >= 1 && n_ans >= 1 && n_aux <= 1024 && n_ans <= 1024,
                "pooled heads: at most 16 samples, at most 1024 outputs per head");
  SIG3D_REQUIRE(hidden >= 16 && hidden % 16 == 0 && hidden <= 1024, "pooled heads: hidden a multiple of 16, <= 1024");
  SIG3D_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "dropout probability must be in [0, 1)");
  SIG3D_REQUIRE(rows && w1a && b1a && w2a && b2a && w1c && b1c && w2c && b2c && pooled && pre && h && aux && ans,
                "null argument");
  if (b == 0) return 0;
  hipLaunchKernelGGL(pool_rows_kernel, dim3(sig3d_ceil_div(hidden / 4, 64), b), dim3(HD_THREADS), 0, stream, q, hidden,
                     rows, pooled);
  const size_t bh = (size_t)b * hidden;
  HdFwdSeg a1 = {w1a, b1a, pooled, pre, h, hidden, 1, 0u};
  HdFwdSeg c1 = {w1c, b1c, pooled, pre + bh, h + bh, hidden, 2, call_id};
  HdFwdSeg a2 = {w2a, b2a, h, nullptr, aux, n_aux, 0, 0u};
  HdFwdSeg c2 = {w2c, b2c, h + bh, nullptr, ans, n_ans, 0, 0u};
  if (b <= 8) {
    launch_fwd<8>(a1, c1, b, hidden, p_drop, rng_counter, stream);
    launch_fwd<8>(a2, c2, b, hidden, p_drop, rng_counter, stream);
  } else {
    launch_fwd<16>(a1, c1, b, hidden, p_drop, rng_counter, stream);
    launch_fwd<16>(a2, c2, b, hidden, p_drop, rng_counter, stream);
  }
  SIG3D_LAUNCH_CHECK("pooled heads forward kernels");
  return 0;
}

extern "C" long sig3d_pooled_heads_work_floats(int b, int hidden) { return 4L * HD_OS * b * hidden; }

extern "C" int sig3d_pooled_heads_bwd(int b, int q, int hidden, int n_aux, int n_ans, const float *daux, const float *dans,
                                      const float *pooled, const float *pre, const float *h, const float *w1a,
                                      const float *w2a, const float *w1c, const float *w2c, float p_drop, unsigned call_id,
                                      const unsigned *rng_counter, float *work, float *grads, float *drows, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SIG3D_REQUIRE(b >= 0 && b <= 16 && q >= 1 && n_aux >= 1 && n_ans >= 1 && n_aux <= 1024 && n_ans <= 1024,
                "pooled heads: at most 16 samples, at most 1024 outputs per head");
  SIG3D_REQUIRE(hidden >= 16 && hidden % 16 == 0 && hidden <= 1024, "pooled heads: hidden a multiple of 16, <= 1024");
  SIG3D_REQUIRE(daux && dans && pooled && pre && h && w1a && w2a && w1c && w2c && work && grads && drows, "null argument");
  if (b == 0) return 0;
  const size_t hh = (size_t)hidden * hidden, bh = (size_t)b * hidden, part = (size_t)HD_OS * bh;
  // grads = [dw1a | db1a | dw2a | db2a | dw1c | db1c | dw2c | db2c]
  float *dw1a = grads, *db1a = dw1a + hh, *dw2a = db1a + hidden, *db2a = dw2a + (size_t)n_aux * hidden;
  float *dw1c = db2a + n_aux, *db1c = dw1c + hh, *dw2c = db1c + hidden, *db2c = dw2c + (size_t)n_ans * hidden;
  // work = partial input gradients [second layer aux | second layer answer | first layer aux | first layer answer]
  float *dh_a = work, *dh_c = work + part, *dp_a = work + 2 * part, *dp_c = work + 3 * part;
  HdBwdSeg a2 = {w2a, daux, nullptr, nullptr, 0, 0, 0u, h, dw2a, db2a, dh_a, n_aux};
  HdBwdSeg c2 = {w2c, dans, nullptr, nullptr, 0, 0, 0u, h + bh, dw2c, db2c, dh_c, n_ans};
  HdBwdSeg a1 = {w1a, nullptr, dh_a, pre, hd_parts(n_aux), 1, 0u, pooled, dw1a, db1a, dp_a, hidden};
  HdBwdSeg c1 = {w1c, nullptr, dh_c, pre + bh, hd_parts(n_ans), 2, call_id, pooled, dw1c, db1c, dp_c, hidden};
  if (b <= 8) {
    launch_bwd<8>(a2, c2, b, hidden, p_drop, rng_counter, stream);
    launch_bwd<8>(a1, c1, b, hidden, p_drop, rng_counter, stream);
  } else {
    launch_bwd<16>(a2, c2, b, hidden, p_drop, rng_counter, stream);
    launch_bwd<16>(a1, c1, b, hidden, p_drop, rng_counter, stream);
  }
  hipLaunchKernelGGL(spread_pooled_grad_kernel, dim3(sig3d_ceil_div((int)bh, HD_THREADS)), dim3(HD_THREADS), 0, stream, b,
                     hidden, q, dp_a, hd_parts(hidden), dp_c, hd_parts(hidden), drows);
  SIG3D_LAUNCH_CHECK("pooled heads backward kernels");
  return 0;
}
