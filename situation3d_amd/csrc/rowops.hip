// rowops.hip -- small row/column helpers of the Q-Former's dense layers for gfx950.
//
// column_sum: bias gradient of an nn.Linear, db[c] = sum_r dY[r][c]
// (the backward of `self.dense(hidden_states)` etc., Qformer.py:242,311,324; torch computes it
// with a generic reduce kernel that takes ~12 us for a 416 x 768 input on MI355X -- 122 launches
// per Q-Former forward+backward).  The op is pure latency (1.3 MB): a workgroup owns 64
// consecutive columns (every row read is one coalesced 256-byte segment per wave) and spreads
// the rows over SIXTEEN waves with eight independent loads in flight per lane, partials meet in
// LDS: no atomics, deterministic summation order.
#include "sig3d_common.h"

namespace {

constexpr int CS_WAVES = 16;

// blockIdx.y = part: x is (parts, rows, cols), out (parts, cols) -- the per-part bias gradients of a
// batched (strided) GEMM pair in one launch.
__global__ __launch_bounds__(CS_WAVES * 64) void column_sum_kernel(int rows, int cols,
                                                                   const float *__restrict__ x,
                                                                   float *__restrict__ out) {
  __shared__ float part[CS_WAVES][64];
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  x += (size_t)blockIdx.y * rows * cols;
  out += (size_t)blockIdx.y * cols;
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  if (c < cols) {
    int r = wave;
    for (; r + 7 * CS_WAVES < rows; r += 8 * CS_WAVES) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] += x[(size_t)(r + i * CS_WAVES) * cols + c];
    }
    for (; r < rows; r += CS_WAVES) acc[0] += x[(size_t)r * cols + c];
  }
  part[wave][lane] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  __syncthreads();
  if (wave == 0 && c < cols) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < CS_WAVES; ++w) s += part[w][lane];
    out[c] = s;
  }
}

// ---- bias + dropout + residual + LayerNorm (BertSelfOutput / BertOutput) ----------------------
// Qformer.py:241-246 / 323-328:   h = dense(x); h = dropout(h); out = LayerNorm(h + input_tensor)
// torch runs the tail as 3 kernels forward (dropout, add, layer_norm) and 5 backward (three
// LayerNorm-backward kernels, dropout backward, the bias-gradient reduce); here the bias add moves
// out of the GEMM epilogue so that the whole tail is ONE row kernel each way.  One wave per row,
// cols <= 64*LN_MAX_PER_LANE; lane l owns columns l, l+64, ... so every access is a coalesced
// 256-byte segment and the keep mask of a row is ONE 16-bit word per lane (bit i = column l+64i).
// The backward also yields d gamma, d beta and d bias: per-lane partial column sums over the rows
// of a wave, the four waves of a workgroup meet in LDS, each workgroup writes one partial row and
// column_sum_kernel folds the partial rows -- no atomics (the first version issued 3*cols float
// atomics per wave, 480 K per call at 416 x 768, and took 24 us), deterministic order.
constexpr int LN_MAX_PER_LANE = 16;  // hidden sizes up to 1024

__device__ __forceinline__ unsigned mix32(unsigned x) {
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  return x;
}

template <int PER_LANE>
__global__ __launch_bounds__(256) void dropout_add_ln_fwd_kernel(
    int rows, int cols, float p_drop, unsigned call_id, const unsigned *__restrict__ rng_counter,
    const float *__restrict__ x, const float *__restrict__ bias, const float *__restrict__ res,
    const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
    float *__restrict__ out, float *__restrict__ v_out, float *__restrict__ mean_out,
    float *__restrict__ rstd_out, unsigned short *__restrict__ mask_out, int part_rows, int mcan, int live_rows,
    int pad_copy, const float *__restrict__ x_slabs, int extra_slabs, size_t slab_stride) {
  const int lane = lane_id();
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  if (row >= live_rows) {
    // rows beyond the live ones: x / v hold live_rows rows only.  pad_copy == 0: padding rows of the
    // two-segment layout (no token lives there), kept finite (zeros) for the batched feed-forward GEMMs that
    // sweep all rows.  pad_copy != 0: rows the block does not touch (the text rows under a cross-attention
    // block, Qformer.py:375-402) pass through from the residual input -- no split / cat around the block.
#pragma unroll
    for (int i = 0; i < PER_LANE; ++i)
      if (lane + 64 * i < cols)
        out[(size_t)row * cols + lane + 64 * i] = pad_copy ? res[(size_t)row * cols + lane + 64 * i] : 0.f;
    if (lane == 0) { mean_out[row] = 0.f; rstd_out[row] = 0.f; }
    return;
  }
  // rows [p*part_rows, (p+1)*part_rows) use parameter set p: bias / gamma / beta are (parts, cols)
  const int poff = (row / part_rows) * cols;
  if (bias) bias += poff;
  gamma += poff;
  beta += poff;
  const unsigned seed = mix32((rng_counter ? *rng_counter : 0u) * 0x9E3779B9u + call_id);
  const unsigned thresh = (unsigned)((double)p_drop * 4294967296.0);
  const float keep_scale = 1.f / (1.f - p_drop);
  // Every load is UNCONDITIONAL (column index clamped, result masked afterwards): a load under
  // `if (c < cols)` becomes its own exec-masked block ending in s_waitcnt, which serialised the
  // 12 column groups into 12 memory round trips (14 us per call instead of ~5).
  const float *xr = x + (size_t)row * cols, *rr = res + (size_t)row * cols;
  float v[PER_LANE], u[PER_LANE], bb[PER_LANE], gam[PER_LANE], bet[PER_LANE];
#pragma unroll
  for (int i = 0; i < PER_LANE; ++i) {
    const int cc = min(lane + 64 * i, cols - 1);
    u[i] = xr[cc];
    v[i] = rr[cc];
    bb[i] = bias ? bias[cc] : 0.f;
    gam[i] = gamma[cc];
    bet[i] = beta[cc];
  }
  // x arrives as the slabs of a split reduction (sig3d_gemm16): they are added here, on the way in -- four slabs'
  // loads in flight at a time (one slab per trip was one memory round trip per slab: 9 -> 24 us per call)
  for (int z0 = 0; z0 < extra_slabs; z0 += 4) {
    float t[4][PER_LANE];
#pragma unroll
    for (int zz = 0; zz < 4; ++zz) {
      const float *xz = x_slabs + (size_t)min(z0 + zz, extra_slabs - 1) * slab_stride + (size_t)row * cols;
#pragma unroll
      for (int i = 0; i < PER_LANE; ++i) t[zz][i] = xz[min(lane + 64 * i, cols - 1)];
    }
#pragma unroll
    for (int zz = 0; zz < 4; ++zz)
#pragma unroll
      for (int i = 0; i < PER_LANE; ++i) u[i] += (z0 + zz < extra_slabs) ? t[zz][i] : 0.f;
  }
  float sum = 0.f;
  unsigned keep_bits = 0;
#pragma unroll
  for (int i = 0; i < PER_LANE; ++i) {
    const int c = lane + 64 * i;
    const unsigned idx = (unsigned)row * (unsigned)cols + (unsigned)c;
    const float t = u[i] + bb[i];
    const bool keep = (p_drop > 0.f) ? (mix32(seed ^ idx * 0x9E3779B9u) >= thresh) : true;
    keep_bits |= (keep ? 1u : 0u) << i;
    v[i] = (c < cols) ? v[i] + (keep ? t * keep_scale : 0.f) : 0.f;
    sum += v[i];
  }
  if (mask_out) mask_out[(size_t)row * 64 + lane] = (unsigned short)keep_bits;
  const float mean = wave_allreduce_sum_f32(sum) / cols;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < PER_LANE; ++i)
    if (lane + 64 * i < cols) sq += (v[i] - mean) * (v[i] - mean);
  // mcan: the MCAN blocks' own LayerNorm (mcan_sqa_module.py:57-69): unbiased std, eps added to the STD
  const float ssq = wave_allreduce_sum_f32(sq);
  const float rstd = mcan ? 1.f / (sqrtf(ssq / (cols - 1)) + eps) : rsqrtf(ssq / cols + eps);
#pragma unroll
  for (int i = 0; i < PER_LANE; ++i) {
    const int c = lane + 64 * i;
    if (c < cols) {
      const size_t idx = (size_t)row * cols + c;
      out[idx] = (v[i] - mean) * rstd * gam[i] + bet[i];
      v_out[idx] = v[i];
    }
  }
  if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
}

template <int PER_LANE>
__global__ __launch_bounds__(256) void dropout_add_ln_bwd_kernel(
    int rows, int cols, float p_drop, int rows_per_wave, const float *__restrict__ dy,
    const float *__restrict__ v, const float *__restrict__ mean, const float *__restrict__ rstd,
    const float *__restrict__ gamma, const unsigned short *__restrict__ mask,
    float *__restrict__ dx, float *__restrict__ dres, float *__restrict__ partial, int part_rows,
    int mcan, float eps, int live_rows, int pad_copy, const float *__restrict__ dy_slabs, int extra_slabs,
    size_t slab_stride, int slab_rows) {
  // partial: (gridDim.x, 3*cols) = per-workgroup [d gamma | d beta | d bias] column sums
  __shared__ float part[3][3][64 * PER_LANE];  // waves 1..3 park their sums here
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  const int wave_global = blockIdx.x * 4 + wave;
  const float keep_scale = 1.f / (1.f - p_drop);
  // a workgroup never straddles two parts (launcher: part_rows % (4*rows_per_wave) == 0)
  gamma += ((blockIdx.x * 4 * rows_per_wave) / part_rows) * cols;
  float ag[PER_LANE], ab[PER_LANE], abias[PER_LANE], gam[PER_LANE];
#pragma unroll
  for (int i = 0; i < PER_LANE; ++i) {
    ag[i] = ab[i] = abias[i] = 0.f;
    gam[i] = gamma[min(lane + 64 * i, cols - 1)];
  }
  for (int rr = 0; rr < rows_per_wave; ++rr) {
    const int row = wave_global * rows_per_wave + rr;
    if (row >= rows) break;
    if (row >= live_rows) {  // padding row: zero gradient to the residual (pass-through row: dy itself), nothing
                             // to the GEMM output (v, dx end earlier)
#pragma unroll
      for (int i = 0; i < PER_LANE; ++i)
        if (lane + 64 * i < cols) {
          float d = pad_copy ? dy[(size_t)row * cols + lane + 64 * i] : 0.f;
          if (pad_copy && row < slab_rows)
            for (int z = 0; z < extra_slabs; ++z) d += dy_slabs[z * slab_stride + (size_t)row * cols + lane + 64 * i];
          dres[(size_t)row * cols + lane + 64 * i] = d;
        }
      continue;
    }
    const float mu = mean[row], rs = rstd[row];
    const unsigned keep_bits = (p_drop > 0.f) ? mask[(size_t)row * 64 + lane] : 0xFFFFu;
    const float *dyr = dy + (size_t)row * cols, *vr = v + (size_t)row * cols;
    float g[PER_LANE], xh[PER_LANE];
#pragma unroll
    for (int i = 0; i < PER_LANE; ++i) {  // unconditional clamped loads, see the forward kernel
      const int cc = min(lane + 64 * i, cols - 1);
      g[i] = dyr[cc];
      xh[i] = vr[cc];
    }
    // dy arrives as the slabs of a split reduction (rows below slab_rows): added on the way in
    if (row < slab_rows)
      for (int z0 = 0; z0 < extra_slabs; z0 += 4) {   // four slabs' loads in flight at a time
        float t[4][PER_LANE];
#pragma unroll
        for (int zz = 0; zz < 4; ++zz) {
          const float *dz = dy_slabs + (size_t)min(z0 + zz, extra_slabs - 1) * slab_stride + (size_t)row * cols;
#pragma unroll
          for (int i = 0; i < PER_LANE; ++i) t[zz][i] = dz[min(lane + 64 * i, cols - 1)];
        }
#pragma unroll
        for (int zz = 0; zz < 4; ++zz)
#pragma unroll
          for (int i = 0; i < PER_LANE; ++i) g[i] += (z0 + zz < extra_slabs) ? t[zz][i] : 0.f;
      }
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int i = 0; i < PER_LANE; ++i) {
      const bool live = lane + 64 * i < cols;
      const float d = live ? g[i] : 0.f;
      xh[i] = live ? (xh[i] - mu) * rs : 0.f;
      g[i] = d * gam[i];
      c1 += g[i];
      c2 += g[i] * xh[i];
      ag[i] += d * xh[i];
      ab[i] += d;
    }
    c1 = wave_allreduce_sum_f32(c1) / cols;
    c2 = wave_allreduce_sum_f32(c2);
    // y = gamma*(v-mu)*r + beta.  standard: r = rsqrt(var+eps)  -> dv = r*(g - mean g - xh*mean(g xh));
    // mcan: r = 1/(s+eps), s = unbiased std       -> dv = r*(g - mean g) - xh*sum(g xh)/((n-1)*s)
    c2 = mcan ? c2 / ((cols - 1) * (1.f / rs - eps) * rs) : c2 / cols;
#pragma unroll
    for (int i = 0; i < PER_LANE; ++i) {
      const int c = lane + 64 * i;
      if (c < cols) {
        const size_t idx = (size_t)row * cols + c;
        const float dv = rs * (g[i] - c1 - xh[i] * c2);
        dres[idx] = dv;
        const float du = ((keep_bits >> i) & 1u) ? dv * keep_scale : 0.f;
        dx[idx] = du;
        abias[i] += du;
      }
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int i = 0; i < PER_LANE; ++i) {
      part[wave - 1][0][lane + 64 * i] = ag[i];
      part[wave - 1][1][lane + 64 * i] = ab[i];
      part[wave - 1][2][lane + 64 * i] = abias[i];
    }
  }
  __syncthreads();
  if (wave == 0) {
    float *dst = partial + (size_t)blockIdx.x * 3 * cols;
#pragma unroll
    for (int i = 0; i < PER_LANE; ++i) {
      const int c = lane + 64 * i;
      if (c < cols) {
        dst[c] = ag[i] + (part[0][0][c] + part[1][0][c] + part[2][0][c]);
        dst[cols + c] = ab[i] + (part[0][1][c] + part[1][1][c] + part[2][1][c]);
        dst[2 * cols + c] = abias[i] + (part[0][2][c] + part[1][2][c] + part[2][2][c]);
      }
    }
  }
}

// rows per wave of the backward tail: one below 2048 rows -- the Q-Former's 512-row matrices then spread over 128
// workgroups instead of 64 and a wave has ONE dependent pass instead of two (-0.09 ms per training step, A/B in the
// step; the partial-sum workspace doubles to 128 rows, folded once for all layers).  qformer._ln_bwd_blocks mirrors this.
inline int ln_bwd_rows_per_wave(int rows) { return rows >= 4096 ? 8 : (rows >= 2048 ? 2 : 1); }

// ---- bias + erf-GELU of BertIntermediate (Qformer.py:311-313), forward and backward -------------
// act = gelu(x + bias[part]),  gx = gy * gelu'(x + bias[part]);  x is the GEMM output WITHOUT bias
// (kept for the backward pass), bias is (parts, cols), rows [p*part_rows, (p+1)*part_rows) use set p.
__device__ __forceinline__ float gelu_erf(float u) { return 0.5f * u * (1.f + erff(u * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_grad(float u) {
  const float cdf = 0.5f * (1.f + erff(u * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * u * u);
  return cdf + u * pdf;
}

template <bool BWD>
__global__ __launch_bounds__(256) void bias_gelu_kernel(long n4, int cols4, int part_rows,
                                                       const float4 *__restrict__ x,
                                                       const float4 *__restrict__ bias,
                                                       const float4 *__restrict__ gy,
                                                       float4 *__restrict__ out) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const int row = (int)(i / cols4), c4 = (int)(i - (long)row * cols4);
  const float4 b = bias[(row / part_rows) * cols4 + c4];
  const float4 v = x[i];
  const float u[4] = {v.x + b.x, v.y + b.y, v.z + b.z, v.w + b.w};
  float4 o;
  if (BWD) {
    const float4 g = gy[i];
    o = make_float4(g.x * gelu_erf_grad(u[0]), g.y * gelu_erf_grad(u[1]), g.z * gelu_erf_grad(u[2]),
                    g.w * gelu_erf_grad(u[3]));
  } else {
    o = make_float4(gelu_erf(u[0]), gelu_erf(u[1]), gelu_erf(u[2]), gelu_erf(u[3]));
  }
  out[i] = o;
}

__global__ void counter_increment_kernel(unsigned *counter) { *counter += 1u; }

}  // namespace

extern "C" int sig3d_counter_increment(unsigned *counter, void *stream_) {
  hipLaunchKernelGGL(counter_increment_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream_, counter);
  SIG3D_LAUNCH_CHECK("counter_increment_kernel");
  return 0;
}

static int ln_tail_fwd(int mcan, int rows, int cols, int part_rows, int live_rows, int pad_copy, float p_drop, unsigned call_id,
                       const unsigned *rng_counter, const float *x, const float *bias, const float *res,
                       const float *gamma, const float *beta, float eps, float *out, float *v, float *mean,
                       float *rstd, unsigned short *mask, void *stream_, const float *x_slabs = nullptr,
                       int extra_slabs = 0, long slab_stride = 0) {
  hipStream_t stream = (hipStream_t)stream_;
  SIG3D_REQUIRE(extra_slabs >= 0 && (extra_slabs == 0 || x_slabs != nullptr), "slab count without slabs");
  SIG3D_REQUIRE(!mcan || cols >= 2, "the unbiased standard deviation needs at least two columns");
  SIG3D_REQUIRE(rows >= 0 && cols >= 1 && cols <= 64 * LN_MAX_PER_LANE, "hidden size must be <= 1024");
  SIG3D_REQUIRE((long)rows * cols < (1L << 32), "rows*cols must fit 32 bits (dropout hash index)");
  SIG3D_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "dropout probability must be in [0, 1)");
  SIG3D_REQUIRE(p_drop == 0.f || mask != nullptr, "a mask buffer is required when p_drop > 0");
  if (rows == 0) return 0;
  if (part_rows <= 0) part_rows = rows;  // one parameter set
  if (live_rows <= 0 || live_rows > rows) live_rows = rows;
  const dim3 grid(sig3d_ceil_div(rows, 4));
  if (cols <= 64 * 12)
    hipLaunchKernelGGL(dropout_add_ln_fwd_kernel<12>, grid, dim3(256), 0, stream, rows, cols, p_drop,
                       call_id, rng_counter, x, bias, res, gamma, beta, eps, out, v, mean, rstd, mask,
                       part_rows, mcan, live_rows, pad_copy, x_slabs, extra_slabs, (size_t)slab_stride);
  else
    hipLaunchKernelGGL(dropout_add_ln_fwd_kernel<LN_MAX_PER_LANE>, grid, dim3(256), 0, stream, rows, cols,
                       p_drop, call_id, rng_counter, x, bias, res, gamma, beta, eps, out, v, mean, rstd,
                       mask, part_rows, mcan, live_rows, pad_copy, x_slabs, extra_slabs, (size_t)slab_stride);
  SIG3D_LAUNCH_CHECK("dropout_add_ln_fwd_kernel");
  return 0;
}

extern "C" int sig3d_dropout_add_ln_fwd(int rows, int cols, int part_rows, int live_rows, float p_drop, unsigned call_id,
                                        const unsigned *rng_counter, const float *x,
                                        const float *bias, const float *res, const float *gamma,
                                        const float *beta, float eps, float *out, float *v,
                                        float *mean, float *rstd, unsigned short *mask,
                                        void *stream_) {
  return ln_tail_fwd(0, rows, cols, part_rows, live_rows < 0 ? -live_rows : live_rows, live_rows < 0, p_drop, call_id, rng_counter, x, bias, res, gamma, beta, eps, out, v,
                     mean, rstd, mask, stream_);
}

extern "C" int sig3d_dropout_add_ln_fwd_slabs(int rows, int cols, int part_rows, int live_rows, float p_drop,
                                              unsigned call_id, const unsigned *rng_counter, const float *x,
                                              const float *x_slabs, int extra_slabs, long slab_stride,
                                              const float *bias, const float *res, const float *gamma,
                                              const float *beta, float eps, float *out, float *v, float *mean,
                                              float *rstd, unsigned short *mask, void *stream_) {
  return ln_tail_fwd(0, rows, cols, part_rows, live_rows < 0 ? -live_rows : live_rows, live_rows < 0, p_drop, call_id,
                     rng_counter, x, bias, res, gamma, beta, eps, out, v, mean, rstd, mask, stream_, x_slabs, extra_slabs,
                     slab_stride);
}

extern "C" int sig3d_dropout_add_mcan_norm_fwd(int rows, int cols, int part_rows, int live_rows, float p_drop, unsigned call_id,
                                               const unsigned *rng_counter, const float *x,
                                               const float *bias, const float *res, const float *gamma,
                                               const float *beta, float eps, float *out, float *v,
                                               float *mean, float *rstd, unsigned short *mask,
                                               void *stream_) {
  return ln_tail_fwd(1, rows, cols, part_rows, live_rows < 0 ? -live_rows : live_rows, live_rows < 0, p_drop, call_id, rng_counter, x, bias, res, gamma, beta, eps, out, v,
                     mean, rstd, mask, stream_);
}

static int ln_tail_bwd(int mcan, float eps, int rows, int cols, int part_rows, int live_rows, int pad_copy, float p_drop, const float *dy,
                       const float *v, const float *mean, const float *rstd, const float *gamma,
                       const unsigned short *mask, float *dx, float *dres, float *dparams, float *workspace,
                       void *stream_, const float *dy_slabs = nullptr, int extra_slabs = 0, long slab_stride = 0,
                       int slab_rows = 0) {
  hipStream_t stream = (hipStream_t)stream_;
  SIG3D_REQUIRE(extra_slabs >= 0 && (extra_slabs == 0 || dy_slabs != nullptr), "slab count without slabs");
  SIG3D_REQUIRE(rows >= 0 && cols >= 1 && cols <= 64 * LN_MAX_PER_LANE, "hidden size must be <= 1024");
  SIG3D_REQUIRE(p_drop == 0.f || mask != nullptr, "the forward's mask buffer is required when p_drop > 0");
  if (rows == 0) {
    if (dparams) SIG3D_HIP_TRY(hipMemsetAsync(dparams, 0, sizeof(float) * 3 * cols, stream));
    return 0;
  }
  SIG3D_REQUIRE(workspace != nullptr, "workspace of 3*cols*ceil(rows/4) floats is required");
  if (live_rows <= 0 || live_rows > rows) live_rows = rows;
  int rpw = ln_bwd_rows_per_wave(rows);
  int parts = 1;
  if (part_rows > 0 && part_rows < rows) {
    SIG3D_REQUIRE(rows % part_rows == 0 && part_rows % 4 == 0, "rows must be parts * part_rows, part_rows % 4 == 0");
    parts = rows / part_rows;
    while (part_rows % (4 * rpw) != 0) rpw >>= 1;  // workgroups must not straddle parts
  } else {
    part_rows = rows;
  }
  const int blocks = sig3d_ceil_div(sig3d_ceil_div(rows, rpw), 4);
  if (cols <= 64 * 12)
    hipLaunchKernelGGL(dropout_add_ln_bwd_kernel<12>, dim3(blocks), dim3(256), 0, stream, rows, cols, p_drop,
                       rpw, dy, v, mean, rstd, gamma, mask, dx, dres, workspace, part_rows, mcan, eps, live_rows, pad_copy,
                       dy_slabs, extra_slabs, (size_t)slab_stride, slab_rows <= 0 ? rows : slab_rows);
  else
    hipLaunchKernelGGL(dropout_add_ln_bwd_kernel<LN_MAX_PER_LANE>, dim3(blocks), dim3(256), 0, stream, rows,
                       cols, p_drop, rpw, dy, v, mean, rstd, gamma, mask, dx, dres, workspace, part_rows, mcan,
                       eps, live_rows, pad_copy, dy_slabs, extra_slabs, (size_t)slab_stride,
                       slab_rows <= 0 ? rows : slab_rows);
  SIG3D_LAUNCH_CHECK("dropout_add_ln_bwd_kernel");
  // fold the per-workgroup partial rows, part by part: dparams is (parts, 3, cols).  dparams == NULL: the
  // caller folds the partial rows itself later (several tails in one sig3d_column_sum launch): the workspace
  // holds (blocks, 3*cols) rows, blocks = ceil(ceil(rows / rpw) / 4), blocks / parts consecutive rows per part
  if (dparams == nullptr) return 0;
  hipLaunchKernelGGL(column_sum_kernel, dim3(sig3d_ceil_div(3 * cols, 64), parts), dim3(CS_WAVES * 64), 0,
                     stream, blocks / parts, 3 * cols, workspace, dparams);
  SIG3D_LAUNCH_CHECK("column_sum_kernel");
  return 0;
}

extern "C" int sig3d_dropout_add_ln_bwd(int rows, int cols, int part_rows, int live_rows, float p_drop, const float *dy,
                                        const float *v, const float *mean, const float *rstd,
                                        const float *gamma, const unsigned short *mask, float *dx,
                                        float *dres, float *dparams, float *workspace,
                                        void *stream_) {
  return ln_tail_bwd(0, 0.f, rows, cols, part_rows, live_rows < 0 ? -live_rows : live_rows, live_rows < 0, p_drop, dy, v, mean, rstd, gamma, mask, dx, dres, dparams,
                     workspace, stream_);
}

extern "C" int sig3d_dropout_add_ln_bwd_slabs(int rows, int cols, int part_rows, int live_rows, float p_drop,
                                              const float *dy, const float *dy_slabs, int extra_slabs,
                                              long slab_stride, int slab_rows, const float *v, const float *mean,
                                              const float *rstd, const float *gamma, const unsigned short *mask,
                                              float *dx, float *dres, float *dparams, float *workspace,
                                              void *stream_) {
  return ln_tail_bwd(0, 0.f, rows, cols, part_rows, live_rows < 0 ? -live_rows : live_rows, live_rows < 0, p_drop, dy, v,
                     mean, rstd, gamma, mask, dx, dres, dparams, workspace, stream_, dy_slabs, extra_slabs, slab_stride,
                     slab_rows);
}

extern "C" int sig3d_dropout_add_mcan_norm_bwd(int rows, int cols, int part_rows, int live_rows, float p_drop, float eps,
                                               const float *dy, const float *v, const float *mean,
                                               const float *rstd, const float *gamma,
                                               const unsigned short *mask, float *dx, float *dres,
                                               float *dparams, float *workspace, void *stream_) {
  return ln_tail_bwd(1, eps, rows, cols, part_rows, live_rows < 0 ? -live_rows : live_rows, live_rows < 0, p_drop, dy, v, mean, rstd, gamma, mask, dx, dres, dparams,
                     workspace, stream_);
}

extern "C" int sig3d_bias_gelu(int rows, int cols, int part_rows, const float *x, const float *bias,
                               const float *gy, float *out, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SIG3D_REQUIRE(rows >= 0 && cols >= 4 && cols % 4 == 0, "cols must be a positive multiple of 4");
  if (rows == 0) return 0;
  if (part_rows <= 0) part_rows = rows;
  const long n4 = (long)rows * (cols / 4);
  const dim3 grid((unsigned)((n4 + 255) / 256));
  if (gy)
    hipLaunchKernelGGL(bias_gelu_kernel<true>, grid, dim3(256), 0, stream, n4, cols / 4, part_rows,
                       (const float4 *)x, (const float4 *)bias, (const float4 *)gy, (float4 *)out);
  else
    hipLaunchKernelGGL(bias_gelu_kernel<false>, grid, dim3(256), 0, stream, n4, cols / 4, part_rows,
                       (const float4 *)x, (const float4 *)bias, (const float4 *)gy, (float4 *)out);
  SIG3D_LAUNCH_CHECK("bias_gelu_kernel");
  return 0;
}

// several column sums in ONE launch: the weight-gradient flush of the Q-Former folds seven kinds of partial rows
// (bias gradients of the projections and the feed-forward pair, the LayerNorm tails' [d gamma | d beta | d bias])
// whose launches were 4-14 us each for 0.1-5 MB.  Same arithmetic and order per job as column_sum_kernel.
struct ColumnSumJobs {
  sig3d_column_sum_job job[SIG3D_COLUMN_SUM_MAX_JOBS];
  int first_block[SIG3D_COLUMN_SUM_MAX_JOBS + 1];     // job j owns workgroups [first_block[j], first_block[j + 1])
  int njobs;
};

__global__ __launch_bounds__(CS_WAVES * 64) void column_sum_multi_kernel(ColumnSumJobs js) {
  __shared__ float part[CS_WAVES][64];
  int j = 0;
  while (j + 1 < js.njobs && (int)blockIdx.x >= js.first_block[j + 1]) ++j;
  const sig3d_column_sum_job q = js.job[j];
  const int local = (int)blockIdx.x - js.first_block[j], cblocks = (q.cols + 63) / 64;
  const int pi = local / cblocks, cb = local - pi * cblocks;
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  const int c = cb * 64 + lane;
  const float *x = q.x + (size_t)pi * q.rows * q.cols;
  float *out = q.out + (size_t)pi * q.cols;
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  if (c < q.cols) {
    int r = wave;
    for (; r + 7 * CS_WAVES < q.rows; r += 8 * CS_WAVES) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] += x[(size_t)(r + i * CS_WAVES) * q.cols + c];
    }
    for (; r < q.rows; r += CS_WAVES) acc[0] += x[(size_t)r * q.cols + c];
  }
  part[wave][lane] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  __syncthreads();
  if (wave == 0 && c < q.cols) {
    float s2 = 0.f;
#pragma unroll
    for (int w = 0; w < CS_WAVES; ++w) s2 += part[w][lane];
    out[c] = s2;
  }
}

extern "C" int sig3d_column_sum_multi(int njobs, const sig3d_column_sum_job *jobs, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SIG3D_REQUIRE(njobs >= 0 && njobs <= SIG3D_COLUMN_SUM_MAX_JOBS, "at most SIG3D_COLUMN_SUM_MAX_JOBS jobs per launch");
  SIG3D_REQUIRE(njobs == 0 || jobs != nullptr, "null job list");
  ColumnSumJobs js = {};
  int blocks = 0;
  for (int j = 0; j < njobs; ++j) {
    const sig3d_column_sum_job &q = jobs[j];
    SIG3D_REQUIRE(q.parts >= 1 && q.rows >= 0 && q.cols >= 0, "negative size");
    if (q.cols == 0) continue;
    SIG3D_REQUIRE(q.out != nullptr && (q.rows == 0 || q.x != nullptr), "null operand");
    js.job[js.njobs] = q;
    js.first_block[js.njobs] = blocks;
    blocks += sig3d_ceil_div(q.cols, 64) * q.parts;
    ++js.njobs;
  }
  js.first_block[js.njobs] = blocks;
  if (blocks == 0) return 0;
  hipLaunchKernelGGL(column_sum_multi_kernel, dim3(blocks), dim3(CS_WAVES * 64), 0, stream, js);
  SIG3D_LAUNCH_CHECK("column_sum_multi_kernel");
  return 0;
}

extern "C" int sig3d_column_sum(int parts, int rows, int cols, const float *x, float *out, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SIG3D_REQUIRE(parts >= 1 && rows >= 0 && cols >= 0, "negative size");
  if (cols == 0) return 0;
  hipLaunchKernelGGL(column_sum_kernel, dim3(sig3d_ceil_div(cols, 64), parts), dim3(CS_WAVES * 64), 0, stream,
                     rows, cols, x, out);
  SIG3D_LAUNCH_CHECK("column_sum_kernel");
  return 0;
}
