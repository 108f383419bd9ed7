// shared_mlp.hip -- fused SharedMLP (1x1 conv + BatchNorm(train) + ReLU stacks) and the
// neighbourhood max-pool of a PointNet++ set-abstraction layer, for gfx950.
//
// Replaces, for the training-mode hot path, what the reference runs as separate torch ops:
//   pt_utils.SharedMLP = [Conv2d(1x1,bias=False) -> BatchNorm2d -> ReLU(inplace)] x L
//   (lib/pointnet2/pytorch_utils.py:11-36, 67-120) followed by
//   F.max_pool2d(new_features, [1, nsample]) (lib/pointnet2/pointnet2_modules.py:259-262).
// the grouped tensor and every layer output otherwise make
// ~7 full HBM round trips each (conv write, BN stats read, BN apply read+write, ReLU
// read+write, next conv read).  Here a layer is ONE kernel:
//   prologue : the previous layer's BatchNorm + ReLU are applied to its RAW conv output while
//              it is loaded as the MFMA B operand (scale/shift per input channel);
//   GEMM     : Y[co][e] = sum_ci W[co][ci] * a[ci][e] on exact-f32 MFMA (32x32x2), W staged in
//              LDS with an odd row stride (conflict-free A-operand reads);
//   epilogue : raw Y is written once and the per-channel sum / sum-of-squares needed for THIS
//              layer's batch statistics are accumulated (lane-local over the workgroup's tiles,
//              one DPP/LDS reduction, f64 atomics per workgroup).
// Each activation is therefore written once and read once; BatchNorm never gets its own pass.
// Batch statistics are finished in f64 (no E[x^2]-E[x]^2 cancellation), running statistics are
// updated exactly like nn.BatchNorm2d (momentum, unbiased running_var, num_batches_tracked).
//
// Layout: activations (B, C, E) with E = npoint*nsample contiguous -- the reference's
// (B, C, npoint, nsample).  MFMA map: rows (M) = output channels, cols (N) = 32 positions (one
// per lane&31), K = input channels; C/D: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
#include <cstdlib>

#include "gemm16_core.h"
#include "shared_mlp_fwd.h"
#include "sig3d_common.h"

extern "C" int sig3d_internal_dw_stream_problem(int b, int cin, int cout, long e, const float *dY, const float *x,
                                                const float *pscale, const float *pshift, const int *n_act, float *dW,
                                                float *work, void *problem_out, int *grid, int *usable);   // gemm16.hip
extern "C" int sig3d_mlp_layer_dw_stream_nofold(int b, int cin, int cout, long e, const float *dY, const float *x,
                                                const float *pscale, const float *pshift, const int *n_act,
                                                float *dW, float *work, void *stream);

namespace {

// ---- BatchNorm statistics -> affine (scale, shift), saved stats, running stats ---------------
// nn.BatchNorm2d training semantics: biased variance for normalisation, unbiased for
// running_var, running = (1-momentum)*running + momentum*batch, num_batches_tracked += 1.
__global__ void bn_finalize_kernel(int c, double count, float eps, float momentum,
                                   const double *__restrict__ stat_sum,
                                   const double *__restrict__ stat_sq,
                                   const float *__restrict__ gamma, const float *__restrict__ beta,
                                   float *__restrict__ scale, float *__restrict__ shift,
                                   float *__restrict__ save_mean, float *__restrict__ save_invstd,
                                   float *__restrict__ running_mean, float *__restrict__ running_var,
                                   long long *__restrict__ num_batches_tracked) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0 && num_batches_tracked) *num_batches_tracked += 1;
  if (i >= c) return;
  const double mean = stat_sum[i] / count;
  double var = stat_sq[i] / count - mean * mean;
  if (var < 0.0) var = 0.0;
  const double invstd = 1.0 / sqrt(var + (double)eps);
  const float g = gamma ? gamma[i] : 1.f, bt = beta ? beta[i] : 0.f;
  scale[i] = (float)((double)g * invstd);
  shift[i] = (float)((double)bt - mean * (double)g * invstd);
  save_mean[i] = (float)mean;
  save_invstd[i] = (float)invstd;
  if (running_mean) {
    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    running_mean[i] = (float)((1.0 - momentum) * (double)running_mean[i] + momentum * mean);
    running_var[i] = (float)((1.0 - momentum) * (double)running_var[i] + momentum * unbiased);
  }
}

// ---- last layer: BN + ReLU + max over the nsample axis (+ arg-max for the backward pass) -----
// y (B,C,P,S) raw -> out (B,C,P), arg (B,C,P) int32 (first maximum, like max_pool2d)
__global__ __launch_bounds__(256) void bn_relu_maxpool_kernel(long groups, int c, int P, int S,
                                                              const float *__restrict__ y,
                                                              const float *__restrict__ scale,
                                                              const float *__restrict__ shift,
                                                              float *__restrict__ out,
                                                              int *__restrict__ arg) {
  // 16 lanes (one DPP row) per group of S samples: coalesced 64-byte segments, row reduction.
  // groups are ordered (b, c, p): channel = (g / P) % c.
  // 32-bit index math (the launcher checks groups < 2^31): a 64-bit divide costs more VALU work
  // than the four loads this thread issues
  const unsigned g = (blockIdx.x * 256u + threadIdx.x) >> 4;
  const int sub = threadIdx.x & 15;
  const unsigned gc = g < (unsigned)groups ? g : (unsigned)groups - 1u;
  const int ch = (int)((gc / (unsigned)P) % (unsigned)c);
  const float sc = scale[ch], sh = shift[ch];
  const float *row = y + (size_t)gc * S;
  float best = -1.f;  // relu output is >= 0, so any sample beats -1
  int bi = 0;
  for (int s = sub; s < S; s += 16) {
    const float v = fmaxf(0.f, row[s] * sc + sh);
    if (v > best) { best = v; bi = s; }  // ascending s + strict '>' : first maximum of this lane
  }
  const int bv = __builtin_bit_cast(int, best);  // -1 or >= 0: ordered as signed ints
  const int mv = row_allreduce_max_i32(bv);
  const unsigned mi = row_allreduce_min_u32(bv == mv ? (unsigned)bi : 0xFFFFFFFFu);
  if (sub == 0 && g < groups) {
    out[g] = __builtin_bit_cast(float, mv);
    arg[g] = (int)mi;
  }
}

// nsample = 4*LPG (16 / 32 / 64, every level of the reference's backbones): one float4 per lane
// and LPG lanes per group, so a group is ONE coalesced 16*LPG-byte segment read by a single load
// instruction (the generic kernel above walks a group in S/16 dependent round trips).  Ties go to
// the smaller sample index at every step: first maximum, like max_pool2d.
template <int LPG>
__global__ __launch_bounds__(256) void bn_relu_maxpool_vec_kernel(unsigned groups, int c, int P,
                                                                  const float *__restrict__ y,
                                                                  const float *__restrict__ scale,
                                                                  const float *__restrict__ shift,
                                                                  float *__restrict__ out,
                                                                  int *__restrict__ arg) {
  const unsigned tid = blockIdx.x * 256u + threadIdx.x;
  const unsigned g = tid / LPG;
  const int sub = (int)(tid % LPG);
  const unsigned gc = g < groups ? g : groups - 1u;
  const int ch = (int)((gc / (unsigned)P) % (unsigned)c);
  const float sc = scale[ch], sh = shift[ch];
  const float4 v = reinterpret_cast<const float4 *>(y + (size_t)gc * (4 * LPG))[sub];
  const float r[4] = {fmaxf(0.f, v.x * sc + sh), fmaxf(0.f, v.y * sc + sh), fmaxf(0.f, v.z * sc + sh),
                      fmaxf(0.f, v.w * sc + sh)};
  float best = r[0];
  int bi = 4 * sub;
#pragma unroll
  for (int q = 1; q < 4; ++q)
    if (r[q] > best) { best = r[q]; bi = 4 * sub + q; }
#pragma unroll
  for (int off = 1; off < LPG; off <<= 1) {
    const float ob = __shfl_xor(best, off);
    const int oi = __shfl_xor(bi, off);
    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  if (sub == 0 && g < groups) {
    out[g] = best;
    arg[g] = bi;
  }
}


// ---- max-pool with a POINT-MAJOR second output --------------------------------------------------------------
// The next set-abstraction level gathers whole feature rows of its neighbours (point-major, (b, p, c)), and the
// Q-Former reads its scene tokens row by row, while the SharedMLP above produced channel-major rows.  Round 2 ran
// a transpose launch between every two levels (six per step, 84 MB of traffic for no algorithmic byte).  These
// kernels write BOTH layouts from the pooling pass: a workgroup owns a tile of PM_TC channels x TP centres,
// pools channel by channel (coalesced reads of y, the channel-major outputs as before), parks the pooled values
// in LDS and writes the tile once more with the channels of a centre adjacent (64-byte runs).
constexpr int PM_TC = 16;

template <int LPG>
__global__ __launch_bounds__(256) void bn_relu_maxpool_pm_kernel(int c, int P, const float *__restrict__ y,
                                                                 const float *__restrict__ scale,
                                                                 const float *__restrict__ shift,
                                                                 float *__restrict__ out, int *__restrict__ arg,
                                                                 float *__restrict__ out_pm) {
  constexpr int TP = 256 / LPG;
  __shared__ float tile[PM_TC][TP + 1];
  const int bi = blockIdx.z, c0 = blockIdx.y * PM_TC, p0 = blockIdx.x * TP;
  const int tid = threadIdx.x, sub = tid % LPG, pl = tid / LPG;
  const int gp = min(p0 + pl, P - 1);
  float4 v[PM_TC];
#pragma unroll
  for (int it = 0; it < PM_TC; ++it) {   // every load of the tile in flight before the first use
    const int ch = min(c0 + it, c - 1);
    v[it] = reinterpret_cast<const float4 *>(y + (((size_t)bi * c + ch) * P + gp) * (4 * LPG))[sub];
  }
#pragma unroll
  for (int it = 0; it < PM_TC; ++it) {
    const int ch = min(c0 + it, c - 1);
    const float sc = scale[ch], sh = shift[ch];
    const float r[4] = {fmaxf(0.f, v[it].x * sc + sh), fmaxf(0.f, v[it].y * sc + sh), fmaxf(0.f, v[it].z * sc + sh),
                        fmaxf(0.f, v[it].w * sc + sh)};
    float best = r[0];
    int bidx = 4 * sub;
#pragma unroll
    for (int q = 1; q < 4; ++q)
      if (r[q] > best) { best = r[q]; bidx = 4 * sub + q; }
#pragma unroll
    for (int off = 1; off < LPG; off <<= 1) {   // ties go to the smaller sample index: first maximum
      const float ob = __shfl_xor(best, off);
      const int oi = __shfl_xor(bidx, off);
      if (ob > best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
    }
    if (sub == 0) {
      tile[it][pl] = best;
      if (c0 + it < c && p0 + pl < P) {
        const size_t g = ((size_t)bi * c + ch) * P + gp;
        out[g] = best;
        arg[g] = bidx;
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < TP * PM_TC; i += 256) {
    const int pp = i / PM_TC, cc = i % PM_TC;
    if (p0 + pp < P && c0 + cc < c) out_pm[((size_t)bi * P + p0 + pp) * c + c0 + cc] = tile[cc][pp];
  }
}

// compact lists: thread = (channel, centre) of a 16 x 16 tile, the centre's segment [seg[j], seg[j+1])
__global__ __launch_bounds__(256) void bn_relu_maxpool_seg_pm_kernel(int c, int P, long E, const float *__restrict__ y,
                                                                     const float *__restrict__ scale,
                                                                     const float *__restrict__ shift,
                                                                     const int *__restrict__ seg,
                                                                     float *__restrict__ out, int *__restrict__ arg,
                                                                     float *__restrict__ out_pm) {
  __shared__ float tile[PM_TC][17];
  const int bi = blockIdx.z, c0 = blockIdx.y * PM_TC, p0 = blockIdx.x * 16;
  const int cl = threadIdx.x >> 4, pl = threadIdx.x & 15;
  const int ch = min(c0 + cl, c - 1), j = min(p0 + pl, P - 1);
  const int *sg = seg + (size_t)bi * (P + 1);
  const int lo = sg[j], hi = sg[j + 1];
  const float sc = scale[ch], sh = shift[ch];
  const float *row = y + ((size_t)bi * c + ch) * E;
  float best = -1.f;
  int bt = 0;
  // eight elements of the segment requested per trip (clamped, masked): one element per trip made every element a
  // memory round trip -- 31 us at 7 distinct neighbours per centre, 246 us at 24 (surface-shaped scenes)
  for (int u0 = lo; u0 < hi; u0 += 8) {
    float t[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) t[q] = row[min(u0 + q, hi - 1)];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float v = fmaxf(0.f, t[q] * sc + sh);
      if (u0 + q < hi && v > best) { best = v; bt = u0 + q - lo; }  // first maximum, like max_pool2d over the padded list
    }
  }
  tile[cl][pl] = best;
  if (c0 + cl < c && p0 + pl < P) {
    out[((size_t)bi * c + ch) * P + j] = best;
    arg[((size_t)bi * c + ch) * P + j] = bt;
  }
  __syncthreads();
  const int pp = threadIdx.x >> 4, cc = threadIdx.x & 15;
  if (p0 + pp < P && c0 + cc < c) out_pm[((size_t)bi * P + p0 + pp) * c + c0 + cc] = tile[cc][pp];
}

// ---- backward: BatchNorm(train) + ReLU ------------------------------------------------------
// With z = y*scale + shift, a = relu(z), xhat = (y - mean)*invstd and upstream gradient dA:
//   dZ = dA * [z > 0];  S1 = sum dZ (= d beta);  S2 = sum dZ*xhat (= d gamma)
//   dY = gamma*invstd * (dZ - S1/n - xhat*S2/n)
// TOP = true: dA is not a dense tensor but the max-pool gradient, i.e. dOut[b,c,j] routed to the
// arg-max sample of each group (and only where the pooled output is > 0).
// `grow` = index of the first group of this (b, c) row, e = position inside the row (< 2^31)
template <bool TOP>
__device__ __forceinline__ float upstream_grad(const float *__restrict__ dA, const float *__restrict__ dOut,
                                               const int *__restrict__ arg, size_t row, size_t grow,
                                               unsigned e, unsigned S, int s_shift) {
  if (!TOP) return dA[row + e];
  // nsample is a power of two in every reference config: shift instead of a divide per element
  const unsigned j = s_shift >= 0 ? e >> s_shift : e / S;
  const int s = (int)(e - j * S);
  return (arg[grow + j] == s) ? dOut[grow + j] : 0.f;
}

constexpr int BNB_THREADS = 256;
constexpr int BNB_CHUNK = 8192;  // elements of one (b, c) row per workgroup
constexpr int BNB_CGRID = 4;     // compact rows: workgroups per (batch, channel) row, each walking chunks up to n_act
                                 // (16 = the dense grid: 7.92-7.99 ms per step; 4: 7.85-7.87; 2: 7.84-7.86, same box)

template <bool TOP>
__global__ __launch_bounds__(BNB_THREADS) void bn_relu_bwd_stats_kernel(
    int c, long E, int S, const float *__restrict__ dA, const float *__restrict__ dOut,
    const int *__restrict__ arg, const float *__restrict__ y, const float *__restrict__ scale,
    const float *__restrict__ shift, const float *__restrict__ mean, const float *__restrict__ invstd,
    double *__restrict__ s1, double *__restrict__ s2) {
  __shared__ float red[2][BNB_THREADS / 64];
  const int ch = blockIdx.y, bi = blockIdx.z;
  const size_t row = ((size_t)bi * c + ch) * E;
  const size_t grow = ((size_t)bi * c + ch) * (size_t)(E / S);
  const int s_shift = (S & (S - 1)) == 0 ? __builtin_ctz(S) : -1;
  const float sc = scale[ch], sh = shift[ch], mu = mean[ch], is = invstd[ch];
  const unsigned e0 = blockIdx.x * (unsigned)BNB_CHUNK, e1 = (unsigned)min(E, (long)e0 + BNB_CHUNK);
  float a1 = 0.f, a2 = 0.f;
  for (unsigned e = e0 + threadIdx.x; e < e1; e += BNB_THREADS) {
    const float yv = y[row + e];
    const float g = upstream_grad<TOP>(dA, dOut, arg, row, grow, e, (unsigned)S, s_shift);
    const float dz = (yv * sc + sh > 0.f) ? g : 0.f;
    a1 += dz;
    a2 += dz * ((yv - mu) * is);
  }
  a1 = wave_allreduce_sum_f32(a1);
  a2 = wave_allreduce_sum_f32(a2);
  if (lane_id() == 0) { red[0][threadIdx.x >> 6] = a1; red[1][threadIdx.x >> 6] = a2; }
  __syncthreads();
  if (threadIdx.x < 2) {
    double t = 0.0;
    for (int w = 0; w < BNB_THREADS / 64; ++w) t += (double)red[threadIdx.x][w];
    unsafeAtomicAdd((threadIdx.x ? s2 : s1) + ch, t);
  }
}

// Max-pool gradient statistics: only the arg-max sample of each group carries gradient, so S1/S2
// need one gathered y per group (b*c*p values) instead of a pass over the whole (b,c,p,s) tensor.
__global__ __launch_bounds__(BNB_THREADS) void bn_relu_bwd_top_stats_kernel(
    int c, int P, int S, const float *__restrict__ dOut, const int *__restrict__ arg,
    const float *__restrict__ y, const float *__restrict__ scale, const float *__restrict__ shift,
    const float *__restrict__ mean, const float *__restrict__ invstd, double *__restrict__ s1,
    double *__restrict__ s2) {
  __shared__ float red[2][BNB_THREADS / 64];
  const int ch = blockIdx.y, bi = blockIdx.z;
  const size_t grow = ((size_t)bi * c + ch) * P;
  const float sc = scale[ch], sh = shift[ch], mu = mean[ch], is = invstd[ch];
  float a1 = 0.f, a2 = 0.f;
  for (int j = blockIdx.x * BNB_THREADS + threadIdx.x; j < P; j += gridDim.x * BNB_THREADS) {
    const float yv = y[(grow + j) * S + arg[grow + j]];
    const float dz = (yv * sc + sh > 0.f) ? dOut[grow + j] : 0.f;
    a1 += dz;
    a2 += dz * ((yv - mu) * is);
  }
  a1 = wave_allreduce_sum_f32(a1);
  a2 = wave_allreduce_sum_f32(a2);
  if (lane_id() == 0) { red[0][threadIdx.x >> 6] = a1; red[1][threadIdx.x >> 6] = a2; }
  __syncthreads();
  if (threadIdx.x < 2) {
    double t = 0.0;
    for (int w = 0; w < BNB_THREADS / 64; ++w) t += (double)red[threadIdx.x][w];
    unsafeAtomicAdd((threadIdx.x ? s2 : s1) + ch, t);
  }
}

template <bool TOP>
__global__ __launch_bounds__(BNB_THREADS) void bn_relu_bwd_apply_kernel(
    int c, long E, int S, double count, const float *__restrict__ dA, const float *__restrict__ dOut,
    const int *__restrict__ arg, const float *__restrict__ y, const float *__restrict__ scale,
    const float *__restrict__ shift, const float *__restrict__ mean, const float *__restrict__ invstd,
    const double *__restrict__ s1, const double *__restrict__ s2, float *__restrict__ dY) {
  const int ch = blockIdx.y, bi = blockIdx.z;
  const size_t row = ((size_t)bi * c + ch) * E;
  const size_t grow = ((size_t)bi * c + ch) * (size_t)(E / S);
  const int s_shift = (S & (S - 1)) == 0 ? __builtin_ctz(S) : -1;
  const float sc = scale[ch], sh = shift[ch], mu = mean[ch], is = invstd[ch];
  const float m1 = (float)(s1[ch] / count), m2 = (float)(s2[ch] / count);
  const unsigned e0 = blockIdx.x * (unsigned)BNB_CHUNK, e1 = (unsigned)min(E, (long)e0 + BNB_CHUNK);
  if (e1 - e0 == (unsigned)BNB_CHUNK && (E & 3) == 0 && (!TOP || (S & 3) == 0)) {
    // full chunk: float4 per thread, all loads of the 8 iterations issued before the first use.
    // TOP: the 4 positions of a float4 share one group (S % 4 == 0) -> ONE unconditional
    // (arg, dOut) pair per float4, selected afterwards (a load under `arg == s ?` serialises).
    constexpr int ITERS = BNB_CHUNK / 4 / BNB_THREADS;
    const float4 *y4 = reinterpret_cast<const float4 *>(y + row + e0);
    const float4 *a4 = TOP ? nullptr : reinterpret_cast<const float4 *>(dA + row + e0);
    float4 *o4 = reinterpret_cast<float4 *>(dY + row + e0);
    float4 yv[ITERS], g[ITERS];
    int sel[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const unsigned q = it * BNB_THREADS + threadIdx.x;
      yv[it] = y4[q];
      if (TOP) {
        const unsigned e = e0 + 4 * q;
        const unsigned j = s_shift >= 0 ? e >> s_shift : e / (unsigned)S;
        sel[it] = arg[grow + j] - (int)(e - j * (unsigned)S);  // 0..3: which lane of the float4
        g[it].x = dOut[grow + j];
      } else {
        g[it] = a4[q];
      }
    }
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const unsigned q = it * BNB_THREADS + threadIdx.x;
      float4 gg = g[it];
      if (TOP) {
        const float d = g[it].x;
        gg.x = sel[it] == 0 ? d : 0.f;
        gg.y = sel[it] == 1 ? d : 0.f;
        gg.z = sel[it] == 2 ? d : 0.f;
        gg.w = sel[it] == 3 ? d : 0.f;
      }
      const float4 v = yv[it];
      float4 o;
      o.x = sc * (((v.x * sc + sh > 0.f) ? gg.x : 0.f) - m1 - (v.x - mu) * is * m2);
      o.y = sc * (((v.y * sc + sh > 0.f) ? gg.y : 0.f) - m1 - (v.y - mu) * is * m2);
      o.z = sc * (((v.z * sc + sh > 0.f) ? gg.z : 0.f) - m1 - (v.z - mu) * is * m2);
      o.w = sc * (((v.w * sc + sh > 0.f) ? gg.w : 0.f) - m1 - (v.w - mu) * is * m2);
      o4[q] = o;
    }
    return;
  }
  for (unsigned e = e0 + threadIdx.x; e < e1; e += BNB_THREADS) {
    const float yv = y[row + e];
    const float g = upstream_grad<TOP>(dA, dOut, arg, row, grow, e, (unsigned)S, s_shift);
    const float dz = (yv * sc + sh > 0.f) ? g : 0.f;
    dY[row + e] = sc * (dz - m1 - (yv - mu) * is * m2);  // sc == gamma * invstd
  }
}

// ---- compact mode (compact.hip): BatchNorm+ReLU+max-pool and their backward over the distinct neighbours ----
// Row layout (b, c, E) as in dense mode, but only positions [0, n_act[b]) exist; the distinct neighbours of
// centre j are positions [seg[j], seg[j+1]), position u stands for mult[u] equal columns.  These tensors are
// a few per cent of the dense ones, so the kernels are plain one-element-per-thread loops.
__global__ __launch_bounds__(256) void bn_relu_maxpool_seg_kernel(int c, int P, long E, const float *__restrict__ y,
                                                                  const float *__restrict__ scale,
                                                                  const float *__restrict__ shift,
                                                                  const int *__restrict__ seg,
                                                                  float *__restrict__ out, int *__restrict__ arg) {
  const int bi = blockIdx.z, ch = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
  if (j >= P) return;
  const int *sg = seg + (size_t)bi * (P + 1);
  const int lo = sg[j], hi = sg[j + 1];
  const float sc = scale[ch], sh = shift[ch];
  const float *row = y + ((size_t)bi * c + ch) * E;
  float best = -1.f;
  int bt = 0;
  for (int u0 = lo; u0 < hi; u0 += 8) {   // eight requests per trip (see bn_relu_maxpool_seg_pm_kernel)
    float t[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) t[q] = row[min(u0 + q, hi - 1)];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float v = fmaxf(0.f, t[q] * sc + sh);
      if (u0 + q < hi && v > best) { best = v; bt = u0 + q - lo; }  // first maximum, like max_pool2d over the padded list
    }
  }
  out[((size_t)bi * c + ch) * P + j] = best;
  arg[((size_t)bi * c + ch) * P + j] = bt;
}

template <bool TOP>
__device__ __forceinline__ float upstream_grad_c(const float *__restrict__ dA, const float *__restrict__ dOut,
                                                 const int *__restrict__ arg, const int *__restrict__ cent,
                                                 const int *__restrict__ sg, size_t row, size_t grow, unsigned e) {
  if (!TOP) return dA[row + e];
  const int j = cent[e];
  return (arg[grow + j] == (int)e - sg[j]) ? dOut[grow + j] : 0.f;
}

template <bool TOP>
__global__ __launch_bounds__(BNB_THREADS) void bn_relu_bwd_stats_c_kernel(
    int c, long E, int P, const float *__restrict__ dA, const float *__restrict__ dOut,
    const int *__restrict__ arg, const float *__restrict__ y, const float *__restrict__ scale,
    const float *__restrict__ shift, const float *__restrict__ mean, const float *__restrict__ invstd,
    const int *__restrict__ n_act, const int *__restrict__ cent_all, const int *__restrict__ seg_all,
    double *__restrict__ s1, double *__restrict__ s2) {
  __shared__ float red[2][BNB_THREADS / 64];
  const int ch = blockIdx.y, bi = blockIdx.z;
  const unsigned En = (unsigned)n_act[bi];
  // The grid is static (a captured step cannot follow the data) but no longer covers the dense row: BNB_CGRID
  // workgroups per (batch, channel) row walk the live positions chunk by chunk -- at SA1 11 % of a row is live and
  // 14 of every 16 workgroups used to load n_act only to exit.
  if (blockIdx.x * (unsigned)BNB_CHUNK >= En) return;
  const size_t row = ((size_t)bi * c + ch) * E, grow = ((size_t)bi * c + ch) * (size_t)P;
  const int *cent = cent_all + (size_t)bi * E, *sg = seg_all + (size_t)bi * (P + 1);
  const float sc = scale[ch], sh = shift[ch], mu = mean[ch], is = invstd[ch];
  // upstream gradients are already summed over the columns a position stands for: no weights here
  float a1 = 0.f, a2 = 0.f;
  for (unsigned e0 = blockIdx.x * (unsigned)BNB_CHUNK; e0 < En; e0 += gridDim.x * (unsigned)BNB_CHUNK) {
  const unsigned e1 = min(En, e0 + (unsigned)BNB_CHUNK);
  unsigned es = e0;   // first position of the scalar loop below
  if (!TOP && (E & 3) == 0 && e1 - e0 >= 4) {
    // float4 per lane, the eight loads of a thread requested before the first use (the scalar loop below is a
    // load -> wait per trip: 33-60 us for the 57 MB of an SA1 layer, i.e. ~1.5 TB/s); indices clamped, not branched
    constexpr int ITERS = BNB_CHUNK / 4 / BNB_THREADS;
    const unsigned nfull = (e1 - e0) >> 2;
    const float4 *y4 = reinterpret_cast<const float4 *>(y + row + e0), *a4 = reinterpret_cast<const float4 *>(dA + row + e0);
    float4 yv[ITERS], gv[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const unsigned q = min((unsigned)(it * BNB_THREADS) + threadIdx.x, nfull - 1);
      yv[it] = y4[q];
      gv[it] = a4[q];
    }
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      if ((unsigned)(it * BNB_THREADS) + threadIdx.x >= nfull) continue;
      const float yy[4] = {yv[it].x, yv[it].y, yv[it].z, yv[it].w}, gg[4] = {gv[it].x, gv[it].y, gv[it].z, gv[it].w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float dz = (yy[k] * sc + sh > 0.f) ? gg[k] : 0.f;
        a1 += dz;
        a2 += dz * ((yy[k] - mu) * is);
      }
    }
    es = e0 + 4 * nfull;
  }
  for (unsigned e = es + threadIdx.x; e < e1; e += BNB_THREADS) {
    const float yv = y[row + e];
    const float g = upstream_grad_c<TOP>(dA, dOut, arg, cent, sg, row, grow, e);
    const float dz = (yv * sc + sh > 0.f) ? g : 0.f;
    a1 += dz;
    a2 += dz * ((yv - mu) * is);
  }
  }
  a1 = wave_allreduce_sum_f32(a1);
  a2 = wave_allreduce_sum_f32(a2);
  if (lane_id() == 0) { red[0][threadIdx.x >> 6] = a1; red[1][threadIdx.x >> 6] = a2; }
  __syncthreads();
  if (threadIdx.x < 2) {
    double t = 0.0;
    for (int w = 0; w < BNB_THREADS / 64; ++w) t += (double)red[threadIdx.x][w];
    unsafeAtomicAdd((threadIdx.x ? s2 : s1) + ch, t);
  }
}

// dY[u] = sum over the mult[u] equal columns of gamma*invstd*(dz - S1/n - xhat*S2/n)
//       = gamma*invstd*(dZ[u] - mult[u]*(S1/n + xhat[u]*S2/n)),  dZ[u] = the summed upstream gradient
template <bool TOP>
__global__ __launch_bounds__(BNB_THREADS) void bn_relu_bwd_apply_c_kernel(
    int c, long E, int P, double count, const float *__restrict__ dA, const float *__restrict__ dOut,
    const int *__restrict__ arg, const float *__restrict__ y, const float *__restrict__ scale,
    const float *__restrict__ shift, const float *__restrict__ mean, const float *__restrict__ invstd,
    const int *__restrict__ n_act, const float *__restrict__ mult_all, const int *__restrict__ cent_all,
    const int *__restrict__ seg_all, const double *__restrict__ s1, const double *__restrict__ s2,
    float *__restrict__ dY) {
  const int ch = blockIdx.y, bi = blockIdx.z;
  const unsigned En = (unsigned)n_act[bi];
  if (blockIdx.x * (unsigned)BNB_CHUNK >= En) return;
  const size_t row = ((size_t)bi * c + ch) * E, grow = ((size_t)bi * c + ch) * (size_t)P;
  const int *cent = cent_all + (size_t)bi * E, *sg = seg_all + (size_t)bi * (P + 1);
  const float *mult = mult_all + (size_t)bi * E;
  const float sc = scale[ch], sh = shift[ch], mu = mean[ch], is = invstd[ch];
  const float m1 = (float)(s1[ch] / count), m2 = (float)(s2[ch] / count);
  for (unsigned e0 = blockIdx.x * (unsigned)BNB_CHUNK; e0 < En; e0 += gridDim.x * (unsigned)BNB_CHUNK) {
  const unsigned e1 = min(En, e0 + (unsigned)BNB_CHUNK);
  unsigned es = e0;
  if (!TOP && (E & 3) == 0 && e1 - e0 >= 4) {   // float4 per lane, loads batched (see the statistics kernel)
    constexpr int ITERS = BNB_CHUNK / 4 / BNB_THREADS;
    const unsigned nfull = (e1 - e0) >> 2;
    const float4 *y4 = reinterpret_cast<const float4 *>(y + row + e0), *a4 = reinterpret_cast<const float4 *>(dA + row + e0);
    const float4 *m4 = reinterpret_cast<const float4 *>(mult + e0);
    float4 *o4 = reinterpret_cast<float4 *>(dY + row + e0);
    float4 yv[ITERS], gv[ITERS], mv[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const unsigned q = min((unsigned)(it * BNB_THREADS) + threadIdx.x, nfull - 1);
      yv[it] = y4[q];
      gv[it] = a4[q];
      mv[it] = m4[q];
    }
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const unsigned q = (unsigned)(it * BNB_THREADS) + threadIdx.x;
      if (q >= nfull) continue;
      float4 o;
      o.x = sc * (((yv[it].x * sc + sh > 0.f) ? gv[it].x : 0.f) - mv[it].x * (m1 + (yv[it].x - mu) * is * m2));
      o.y = sc * (((yv[it].y * sc + sh > 0.f) ? gv[it].y : 0.f) - mv[it].y * (m1 + (yv[it].y - mu) * is * m2));
      o.z = sc * (((yv[it].z * sc + sh > 0.f) ? gv[it].z : 0.f) - mv[it].z * (m1 + (yv[it].z - mu) * is * m2));
      o.w = sc * (((yv[it].w * sc + sh > 0.f) ? gv[it].w : 0.f) - mv[it].w * (m1 + (yv[it].w - mu) * is * m2));
      o4[q] = o;
    }
    es = e0 + 4 * nfull;
  }
  for (unsigned e = es + threadIdx.x; e < e1; e += BNB_THREADS) {
    const float yv = y[row + e];
    const float g = upstream_grad_c<TOP>(dA, dOut, arg, cent, sg, row, grow, e);
    const float dz = (yv * sc + sh > 0.f) ? g : 0.f;
    dY[row + e] = sc * (dz - mult[e] * (m1 + (yv - mu) * is * m2));
  }
  }
}

// Top layer (upstream gradient = max-pool gradient): only the arg-max entry of a centre's segment carries
// gradient, so the statistics need ONE gathered y per (batch, channel, centre) and the apply pass splits into
// a coalesced sweep (the correction term of every position) plus one fix-up per centre.
__global__ __launch_bounds__(BNB_THREADS) void bn_relu_bwd_top_stats_c_kernel(
    int c, long E, int P, const float *__restrict__ dOut, const int *__restrict__ arg,
    const float *__restrict__ y, const float *__restrict__ scale, const float *__restrict__ shift,
    const float *__restrict__ mean, const float *__restrict__ invstd, const int *__restrict__ seg_all,
    double *__restrict__ s1, double *__restrict__ s2) {
  __shared__ float red[2][BNB_THREADS / 64];
  const int ch = blockIdx.y, bi = blockIdx.z;
  const size_t row = ((size_t)bi * c + ch) * E, grow = ((size_t)bi * c + ch) * (size_t)P;
  const int *sg = seg_all + (size_t)bi * (P + 1);
  const float sc = scale[ch], sh = shift[ch], mu = mean[ch], is = invstd[ch];
  float a1 = 0.f, a2 = 0.f;
  for (int j = blockIdx.x * BNB_THREADS + threadIdx.x; j < P; j += gridDim.x * BNB_THREADS) {
    const float yv = y[row + sg[j] + arg[grow + j]];
    const float dz = (yv * sc + sh > 0.f) ? dOut[grow + j] : 0.f;
    a1 += dz;
    a2 += dz * ((yv - mu) * is);
  }
  a1 = wave_allreduce_sum_f32(a1);
  a2 = wave_allreduce_sum_f32(a2);
  if (lane_id() == 0) { red[0][threadIdx.x >> 6] = a1; red[1][threadIdx.x >> 6] = a2; }
  __syncthreads();
  if (threadIdx.x < 2) {
    double t = 0.0;
    for (int w = 0; w < BNB_THREADS / 64; ++w) t += (double)red[threadIdx.x][w];
    unsafeAtomicAdd((threadIdx.x ? s2 : s1) + ch, t);
  }
}

__global__ __launch_bounds__(BNB_THREADS) void bn_relu_bwd_top_sweep_c_kernel(
    int c, long E, double count, const float *__restrict__ y, const float *__restrict__ scale,
    const float *__restrict__ mean, const float *__restrict__ invstd, const int *__restrict__ n_act,
    const float *__restrict__ mult_all, const double *__restrict__ s1, const double *__restrict__ s2,
    float *__restrict__ dY) {
  const int ch = blockIdx.y, bi = blockIdx.z;
  const unsigned En = (unsigned)n_act[bi];
  if (blockIdx.x * (unsigned)BNB_CHUNK >= En) return;
  const size_t row = ((size_t)bi * c + ch) * E;
  const float *mult = mult_all + (size_t)bi * E;
  const float sc = scale[ch], mu = mean[ch], is = invstd[ch];
  const float m1 = (float)(s1[ch] / count), m2 = (float)(s2[ch] / count);
  for (unsigned e0 = blockIdx.x * (unsigned)BNB_CHUNK; e0 < En; e0 += gridDim.x * (unsigned)BNB_CHUNK) {
  const unsigned e1 = min(En, e0 + (unsigned)BNB_CHUNK);
  unsigned es = e0;
  if ((E & 3) == 0 && e1 - e0 >= 4) {   // float4 per lane, loads batched (see the statistics kernel)
    constexpr int ITERS = BNB_CHUNK / 4 / BNB_THREADS;
    const unsigned nfull = (e1 - e0) >> 2;
    const float4 *y4 = reinterpret_cast<const float4 *>(y + row + e0), *m4 = reinterpret_cast<const float4 *>(mult + e0);
    float4 *o4 = reinterpret_cast<float4 *>(dY + row + e0);
    float4 yv[ITERS], mv[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const unsigned q = min((unsigned)(it * BNB_THREADS) + threadIdx.x, nfull - 1);
      yv[it] = y4[q];
      mv[it] = m4[q];
    }
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const unsigned q = (unsigned)(it * BNB_THREADS) + threadIdx.x;
      if (q >= nfull) continue;
      o4[q] = make_float4(-sc * mv[it].x * (m1 + (yv[it].x - mu) * is * m2), -sc * mv[it].y * (m1 + (yv[it].y - mu) * is * m2),
                          -sc * mv[it].z * (m1 + (yv[it].z - mu) * is * m2), -sc * mv[it].w * (m1 + (yv[it].w - mu) * is * m2));
    }
    es = e0 + 4 * nfull;
  }
  for (unsigned e = es + threadIdx.x; e < e1; e += BNB_THREADS)
    dY[row + e] = -sc * mult[e] * (m1 + (y[row + e] - mu) * is * m2);
  }
}

__global__ __launch_bounds__(BNB_THREADS) void bn_relu_bwd_top_fix_c_kernel(
    int c, long E, int P, const float *__restrict__ dOut, const int *__restrict__ arg,
    const float *__restrict__ y, const float *__restrict__ scale, const float *__restrict__ shift,
    const int *__restrict__ seg_all, float *__restrict__ dY) {
  const int ch = blockIdx.y, bi = blockIdx.z, j = blockIdx.x * BNB_THREADS + threadIdx.x;
  if (j >= P) return;
  const size_t row = ((size_t)bi * c + ch) * E, grow = ((size_t)bi * c + ch) * (size_t)P;
  const size_t u = row + seg_all[(size_t)bi * (P + 1) + j] + arg[grow + j];
  const float sc = scale[ch], sh = shift[ch];
  if (y[u] * sc + sh > 0.f) dY[u] += sc * dOut[grow + j];  // one writer per position: segments are disjoint
}

// ---- the max-pool gradient arriving POINT-MAJOR ---------------------------------------------------------------
// A level that handed its pooled features on point-major (bn_relu_maxpool_pm) gets their gradient back the same
// way: dOut_pm (b, P, c) -- rows scattered by the level above, or the Q-Former's token gradient.  The BatchNorm
// backward below walks channel-major rows; instead of a transpose launch in front of it, this kernel turns the
// tile through LDS (64-byte runs in, coalesced rows out), writes the channel-major copy the apply / fix kernels
// read, and takes the top layer's statistics S1 = sum dZ, S2 = sum dZ * xhat on the way (what
// bn_relu_bwd_top_stats[_c] computes): transpose + statistics pass = ONE launch.
// seg == nullptr: dense lists (arg-max sample of centre j is y[(grow + j) * S + arg]); else compact segments.
__global__ __launch_bounds__(256) void bn_relu_bwd_top_from_pm_kernel(
    int c, int P, int S, long E, const float *__restrict__ dout_pm, const int *__restrict__ arg,
    const float *__restrict__ y, const float *__restrict__ scale, const float *__restrict__ shift,
    const float *__restrict__ mean, const float *__restrict__ invstd, const int *__restrict__ seg_all,
    float *__restrict__ dout_cm, double *__restrict__ s1, double *__restrict__ s2) {
  __shared__ float tile[PM_TC][65];
  const int bi = blockIdx.z, c0 = blockIdx.y * PM_TC, p0 = blockIdx.x * 64;
  const int tid = threadIdx.x;
#pragma unroll
  for (int it = 0; it < 4; ++it) {   // 64 centres x 16 channels, a centre's 16 channels adjacent
    const int i = it * 256 + tid, pp = i / PM_TC, cc = i % PM_TC;
    tile[cc][pp] = (p0 + pp < P && c0 + cc < c) ? dout_pm[((size_t)bi * P + p0 + pp) * c + c0 + cc] : 0.f;
  }
  __syncthreads();
  const int lane = tid & 63, w = tid >> 6;
  const int j = p0 + lane;
#pragma unroll
  for (int it = 0; it < PM_TC / 4; ++it) {   // a wave = one channel's 64 centres
    const int cc = it * 4 + w, ch = c0 + cc;
    if (ch >= c) break;   // wave-uniform
    float a1 = 0.f, a2 = 0.f;
    if (j < P) {
      const float g = tile[cc][lane];
      const size_t grow = ((size_t)bi * c + ch) * (size_t)P;
      dout_cm[grow + j] = g;
      const int a = arg[grow + j];
      const size_t u = seg_all ? ((size_t)bi * c + ch) * (size_t)E + seg_all[(size_t)bi * (P + 1) + j] + a
                               : (grow + j) * (size_t)S + a;
      const float yv = y[u];
      const float dz = (yv * scale[ch] + shift[ch] > 0.f) ? g : 0.f;
      a1 = dz;
      a2 = dz * ((yv - mean[ch]) * invstd[ch]);
    }
    a1 = wave_allreduce_sum_f32(a1);
    a2 = wave_allreduce_sum_f32(a2);
    if (lane == 0) {
      unsafeAtomicAdd(s1 + ch, (double)a1);
      unsafeAtomicAdd(s2 + ch, (double)a2);
    }
  }
}

// ---- stand-alone batch statistics / BN+ReLU apply (small levels: the 1x1 conv is a library GEMM) ----
// sum(y), sum(y^2) per channel of y (B, C, E); same launch geometry as the backward statistics.
__global__ __launch_bounds__(BNB_THREADS) void channel_stats_kernel(int c, long E, const float *__restrict__ y,
                                                                    double *__restrict__ s1,
                                                                    double *__restrict__ s2) {
  __shared__ float red[2][BNB_THREADS / 64];
  const int ch = blockIdx.y, bi = blockIdx.z;
  const size_t row = ((size_t)bi * c + ch) * E;
  const unsigned e0 = blockIdx.x * (unsigned)BNB_CHUNK, e1 = (unsigned)min(E, (long)e0 + BNB_CHUNK);
  float a1 = 0.f, a2 = 0.f;
  if (e1 - e0 == (unsigned)BNB_CHUNK && (E & 3) == 0) {
    const float4 *y4 = reinterpret_cast<const float4 *>(y + row + e0);
    float4 v[BNB_CHUNK / 4 / BNB_THREADS];
#pragma unroll
    for (int it = 0; it < BNB_CHUNK / 4 / BNB_THREADS; ++it) v[it] = y4[it * BNB_THREADS + threadIdx.x];
#pragma unroll
    for (int it = 0; it < BNB_CHUNK / 4 / BNB_THREADS; ++it) {
      a1 += (v[it].x + v[it].y) + (v[it].z + v[it].w);
      a2 += (v[it].x * v[it].x + v[it].y * v[it].y) + (v[it].z * v[it].z + v[it].w * v[it].w);
    }
  } else {
    for (unsigned e = e0 + threadIdx.x; e < e1; e += BNB_THREADS) {
      const float yv = y[row + e];
      a1 += yv;
      a2 += yv * yv;
    }
  }
  a1 = wave_allreduce_sum_f32(a1);
  a2 = wave_allreduce_sum_f32(a2);
  if (lane_id() == 0) { red[0][threadIdx.x >> 6] = a1; red[1][threadIdx.x >> 6] = a2; }
  __syncthreads();
  if (threadIdx.x < 2) {
    double t = 0.0;
    for (int w = 0; w < BNB_THREADS / 64; ++w) t += (double)red[threadIdx.x][w];
    unsafeAtomicAdd((threadIdx.x ? s2 : s1) + ch, t);
  }
}

// out = relu(y * scale[c] + shift[c])  (BatchNorm2d(train) + ReLU of a middle layer)
__global__ __launch_bounds__(BNB_THREADS) void bn_relu_apply_kernel(int c, long E, const float *__restrict__ y,
                                                                    const float *__restrict__ scale,
                                                                    const float *__restrict__ shift,
                                                                    float *__restrict__ out) {
  const int ch = blockIdx.y, bi = blockIdx.z;
  const size_t row = ((size_t)bi * c + ch) * E;
  const float sc = scale[ch], sh = shift[ch];
  const unsigned e0 = blockIdx.x * (unsigned)BNB_CHUNK, e1 = (unsigned)min(E, (long)e0 + BNB_CHUNK);
  if (e1 - e0 == (unsigned)BNB_CHUNK && (E & 3) == 0) {
    const float4 *y4 = reinterpret_cast<const float4 *>(y + row + e0);
    float4 *o4 = reinterpret_cast<float4 *>(out + row + e0);
    float4 v[BNB_CHUNK / 4 / BNB_THREADS];
#pragma unroll
    for (int it = 0; it < BNB_CHUNK / 4 / BNB_THREADS; ++it) v[it] = y4[it * BNB_THREADS + threadIdx.x];
#pragma unroll
    for (int it = 0; it < BNB_CHUNK / 4 / BNB_THREADS; ++it)
      o4[it * BNB_THREADS + threadIdx.x] = make_float4(fmaxf(0.f, v[it].x * sc + sh), fmaxf(0.f, v[it].y * sc + sh),
                                                       fmaxf(0.f, v[it].z * sc + sh), fmaxf(0.f, v[it].w * sc + sh));
    return;
  }
  for (unsigned e = e0 + threadIdx.x; e < e1; e += BNB_THREADS) out[row + e] = fmaxf(0.f, y[row + e] * sc + sh);
}

// ---- backward: weight gradient dW[co][ci] = sum_{b,e} dY[b,co,e] * a[b,ci,e] ----------------
// a = x (first layer) or relu(x*pscale + pshift) (x = previous layer's raw conv output).
// A reduction over ALL positions with a small (cout x cin) result: no LDS, no barriers.  One
// wave owns a 2x2 block of 32x32 output tiles and a chunk of positions; both MFMA operands are
// rows of global tensors, so a lane reads a CONTIGUOUS 64-byte run of its own row per 32-position
// step (the K index is permuted: step s pairs position s of the first 16 with position 16+s of
// the last 16, identically for both operands).  Fragments of the next step are loaded while the
// MFMAs of the current one issue.  Partial tiles are combined across the workgroup's waves with
// LDS atomics and flushed with one global f32 atomic per element per workgroup.
constexpr int DW_WAVES = 4;

// GATHER: the activation operand of a FIRST layer is gathered on load like in mlp_layer_fwd_kernel<.., GATHER>
// (the grouped tensor is never stored): lane = input channel ci in the reference's order (0..2 = xyz offset,
// 3.. = feature ci - 3), so for one position the 32 lanes of a half-wave read 32 consecutive floats of the
// neighbour's point-major feature row; the 16 neighbour indices of a fragment are the same for every lane.
template <bool PROLOGUE, bool VEC, bool GATHER>
__global__ __launch_bounds__(DW_WAVES * 64, 2) void mlp_dw_kernel(
    int cin, int cout, long E, int steps_per_wave, int nblk_n, const float *__restrict__ dY,
    const float *__restrict__ x, const float *__restrict__ pscale, const float *__restrict__ pshift,
    float *__restrict__ dW, const int *__restrict__ n_act, MlpGather ga) {
  // one private image of the 2x2 tile block per wave (plain stores), summed once at the end: four
  // waves doing ds_add_f32 onto ONE image cost ~10 us per workgroup (cf. the attention backward)
  __shared__ float s_tile[DW_WAVES][4][32][33];
  const int lane = lane_id(), l31 = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int bi = blockIdx.z;
  const int co0 = (blockIdx.y / nblk_n) * 64, ci0 = (blockIdx.y % nblk_n) * 64;

  // per-lane row pointers / prologue constants of the two A rows (co) and two B rows (ci)
  const float *arow[2];
  const float *brow[2];
  float bsc[2], bsh[2];
  bool bok[2], aok[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int co = co0 + 32 * t + l31, ci = ci0 + 32 * t + l31;
    aok[t] = co < cout;
    bok[t] = ci < cin;
    arow[t] = dY + ((size_t)bi * cout + (aok[t] ? co : 0)) * E;
    brow[t] = GATHER ? nullptr : x + ((size_t)bi * cin + (bok[t] ? ci : 0)) * E;
    bsc[t] = (PROLOGUE && bok[t]) ? pscale[ci] : 1.f;
    bsh[t] = (PROLOGUE && bok[t]) ? pshift[ci] : 0.f;
  }
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x16){0};
  // ragged blocks (cin = 6, 131, 259: the last 64-wide block holds 3-6 channels; cout = 64 exactly):
  // 32-row tiles without any channel are skipped, loads and MFMAs alike (uniform per workgroup)
  const int ni = (cout - co0 > 32) ? 2 : 1, nj = (cin - ci0 > 32) ? 2 : 1;

  // compact mode: positions [0, n_act[b]) of every row exist (E stays the row stride); the steps are
  // re-dealt over the launched waves so that the whole grid shares the shorter range
  const long En = n_act ? (long)n_act[bi] : E;
  const long n_steps = (En + 31) / 32;
  if (n_act) steps_per_wave = (int)((n_steps + (long)gridDim.x * DW_WAVES - 1) / ((long)gridDim.x * DW_WAVES));
  const long st_begin = ((long)blockIdx.x * DW_WAVES + wave) * steps_per_wave;
  const long st_end = min(n_steps, st_begin + steps_per_wave);
  if (n_act && (long)blockIdx.x * DW_WAVES * steps_per_wave >= n_steps) return;  // whole workgroup idle

  auto load_frag = [&](float (&f)[16], const float *row, long e0, bool ok) {
    const long eb = e0 + 16 * half;
    if (VEC) {
      const float4 *p4 = reinterpret_cast<const float4 *>(row + eb);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = p4[q];
        f[4 * q + 0] = v.x; f[4 * q + 1] = v.y; f[4 * q + 2] = v.z; f[4 * q + 3] = v.w;
      }
    } else {
#pragma unroll
      for (int q = 0; q < 16; ++q) f[q] = (eb + q < E) ? row[eb + q] : 0.f;
    }
    (void)ok;
  };

  // gathered B fragments of BOTH channel tiles of a step (they share the 16 neighbour indices).  The indices
  // are loaded ONE STEP BEFORE the gathers that use them (gi: 16 registers, refilled right after the gathers
  // have been issued): an index load followed at once by its dependent gathers makes every step wait for a
  // full memory round trip in the middle of its prefetch (in-order vmcnt), 2.3x the stored-tensor kernel.
  int gi[16];
  auto load_idx = [&](long e0) {
    const long eb = e0 + 16 * half;
    const int *ip = ga.idx + (size_t)bi * E;
#pragma unroll
    for (int q = 0; q < 16; ++q) gi[q] = ip[min(eb + q, En - 1)];   // positions past En are zeroed by the tail logic
  };
  auto load_gather = [&](float (&f0)[16], float (&f1)[16], long e0) {
    const long eb = e0 + 16 * half;
    const int fc0 = max(ci0 + l31 - 3, 0), fc1 = min(ci0 + 32 + l31 - 3, ga.C - 1);   // clamped feature columns
    const bool xyz_lane = ci0 == 0 && l31 < 3;                                        // tile 0 of block 0 only
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const long pos = min(eb + q, En - 1);
      const float *frow = ga.feat_pm + ((size_t)bi * ga.N + gi[q]) * ga.C;
      float v0 = frow[min(fc0, ga.C - 1)];
      if (ci0 == 0) {   // uniform: the block that holds the three xyz channels
        const int ctr = ga.centre_of ? ga.centre_of[(size_t)bi * E + pos] : (int)(pos / ga.S);
        const int j = min(l31, 2);
        float d = __fsub_rn(ga.xyz[((size_t)bi * ga.N + gi[q]) * 3 + j], ga.centre[((size_t)bi * ga.P + ctr) * 3 + j]);
        if (ga.normalize) d = __fdiv_rn(d, ga.radius);
        v0 = xyz_lane ? d : v0;
      }
      f0[q] = v0;
      if (nj > 1) f1[q] = frow[fc1];
    }
  };

  float fa[2][2][16], fb[2][2][16];  // [buffer][tile][k]
  if (st_begin < st_end) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (t < ni) load_frag(fa[0][t], arow[t], st_begin * 32, aok[t]);
      if (!GATHER && t < nj) load_frag(fb[0][t], brow[t], st_begin * 32, bok[t]);
    }
    if (GATHER) {
      load_idx(st_begin * 32);
      load_gather(fb[0][0], fb[0][1], st_begin * 32);
      load_idx(min(st_begin + 1, st_end - 1) * 32);
    }
  }
  for (long st = st_begin; st < st_end; st += 2) {
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
      const long cur = st + ph;
      if (cur < st_end) {
        if (cur + 1 < st_end) {
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            if (t < ni) load_frag(fa[ph ^ 1][t], arow[t], (cur + 1) * 32, aok[t]);
            if (!GATHER && t < nj) load_frag(fb[ph ^ 1][t], brow[t], (cur + 1) * 32, bok[t]);
          }
          if (GATHER) {
            load_gather(fb[ph ^ 1][0], fb[ph ^ 1][1], (cur + 1) * 32);   // indices: loaded during the previous step
            load_idx(min(cur + 2, st_end - 1) * 32);
          }
        }
        const bool tail = (!VEC || n_act != nullptr || GATHER) && (cur * 32 + 32 > En);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          float av[2], bv[2];
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            av[t] = aok[t] ? fa[ph][t][k] : 0.f;
            float v = fb[ph][t][k];
            if (PROLOGUE) v = fmaxf(0.f, v * bsc[t] + bsh[t]);
            if (tail && (cur * 32 + 16 * half + k >= En)) { v = 0.f; av[t] = 0.f; }  // (memory past En is not ours)
            bv[t] = bok[t] ? v : 0.f;
          }
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              if (i < ni && j < nj)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
      }
    }
  }
  // combine the workgroup's waves through LDS, then one global atomic per element
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) s_tile[wave][i * 2 + j][mrow(r, half)][l31] = acc[i][j][r];
  __syncthreads();
  for (int idx = threadIdx.x; idx < 4 * 32 * 32; idx += DW_WAVES * 64) {
    const int tile = idx >> 10, r = (idx >> 5) & 31, c = idx & 31;
    const int co = co0 + 32 * (tile >> 1) + r, ci = ci0 + 32 * (tile & 1) + c;
    const float v = (s_tile[0][tile][r][c] + s_tile[1][tile][r][c]) + (s_tile[2][tile][r][c] + s_tile[3][tile][r][c]);
    if (co < cout && ci < cin) unsafeAtomicAdd(dW + (size_t)co * cin + ci, v);
  }
}

}  // namespace

// ---- C ABI ----------------------------------------------------------------------------------
#ifdef SIG3D_MLP_TIMING
extern "C" int sig3d_debug_mlp_cycles(unsigned long long *host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_ml_cycles), sizeof(unsigned long long) * 64);
}

extern "C" int sig3d_debug_mlp_marks(unsigned long long *host_out, int *n) {
  int zero = 0;
  hipError_t e = hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_ml_marks), sizeof(unsigned long long) * 64);
  if (e == hipSuccess) e = hipMemcpyFromSymbol(n, HIP_SYMBOL(g_ml_nmarks), sizeof(int));
  if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(g_ml_nmarks), &zero, sizeof(int));
  return (int)e;
}
#endif

// mlp_layer_fwd_kernel's instances by channel-tile width: shared_mlp_fwd1/2/4/4r.hip (one translation unit when the phase
// timing of tools/mlp_timing.py is compiled in: its marks are device globals of the unit that holds the kernel)
#ifdef SIG3D_MLP_TIMING
#define SIG3D_MLP_FWD_DECL(NT, R) SIG3D_MLP_FWD_INSTANCES(NT, R)
#else
#define SIG3D_MLP_FWD_DECL(NT, R) extern "C" int SIG3D_MLP_FWD_NAME(NT, R)(SIG3D_MLP_FWD_ARGS);
#endif
SIG3D_MLP_FWD_DECL(1, 0)
SIG3D_MLP_FWD_DECL(1, 1)
SIG3D_MLP_FWD_DECL(2, 0)
SIG3D_MLP_FWD_DECL(2, 1)
SIG3D_MLP_FWD_DECL(4, 0)
SIG3D_MLP_FWD_DECL(4, 1)

// compact-mode operands of the call in flight (set by the *_compact entry points around the regular
// dispatch, so that the entry points keep their signatures)
static thread_local const int *tl_n_act = nullptr;
static thread_local const float *tl_mult = nullptr;

static thread_local const MlpGather *tl_gather = nullptr;
static thread_local int tl_w_t = 0;   // the weight operand of the call in flight is given transposed (sig3d_mlp_layer_dx)

template <int NT>
static int layer_fwd_tile(bool prologue, bool vec, int b, int cin, int cout, long e, const float *x,
                          const float *w, const float *pscale, const float *pshift, float *y,
                          double *stat_sum, double *stat_sq, hipStream_t stream) {
  static_assert(NT == 1 || NT == 2 || NT == 4, "channel-tile widths built: 32, 64, 128");
  const MlpFwdCall call = {tl_n_act, tl_mult, tl_gather, tl_w_t};
  const bool ragged = mlp_fwd_ragged<NT>(cin, cout);
  auto fn = NT == 4 ? (ragged ? SIG3D_MLP_FWD_NAME(4, 1) : SIG3D_MLP_FWD_NAME(4, 0))
          : NT == 2 ? (ragged ? SIG3D_MLP_FWD_NAME(2, 1) : SIG3D_MLP_FWD_NAME(2, 0))
                    : (ragged ? SIG3D_MLP_FWD_NAME(1, 1) : SIG3D_MLP_FWD_NAME(1, 0));
  return fn(&call, prologue, vec, b, cin, cout, e, x, w, pscale, pshift, y, stat_sum, stat_sq, stream);
}

// accumulate == 0: statistic / gradient accumulators are zeroed by the call; != 0: the caller zeroed
// them (e.g. all layers of a stack with ONE fill) and the call only adds.
static int zero_pair(double *a, double *b2, int n, int accumulate, hipStream_t stream) {
  if (accumulate || a == nullptr) return 0;
  if (b2 == a + n) {  // the usual (2, n) allocation: one memset node
    SIG3D_HIP_TRY(hipMemsetAsync(a, 0, sizeof(double) * 2 * n, stream));
  } else {
    SIG3D_HIP_TRY(hipMemsetAsync(a, 0, sizeof(double) * n, stream));
    SIG3D_HIP_TRY(hipMemsetAsync(b2, 0, sizeof(double) * n, stream));
  }
  return 0;
}

extern "C" int sig3d_mlp_layer_fwd(int b, int cin, int cout, long e, const float *x, const float *w,
                                   const float *pscale, const float *pshift, float *y,
                                   double *stat_sum, double *stat_sq, int accumulate, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SIG3D_REQUIRE(b >= 0 && cin >= 1 && cout >= 1 && e >= 0, "bad size");
  SIG3D_REQUIRE((pscale == nullptr) == (pshift == nullptr), "pscale/pshift must come together");
  SIG3D_REQUIRE((stat_sum == nullptr) == (stat_sq == nullptr), "stat_sum/stat_sq must come together");
  if (int rc = zero_pair(stat_sum, stat_sq, cout, accumulate, stream)) return rc;
  if (b == 0 || e == 0) return 0;
  SIG3D_REQUIRE((long)cin * e < (1L << 31) && (long)cout * e < (1L << 31),
                "cin*e and cout*e must stay below 2^31 (32-bit addressing inside the kernel)");
  const int kpad = ml_kpad(cin), ldw = kpad | 1;
  SIG3D_REQUIRE(sizeof(float) * ((size_t)32 * ldw + 2 * kpad + ML_WAVES * 2 * 32 + ML_WAVES * 16 * ML_TRLD) <= 160 * 1024,
                "input channel count too large for the LDS weight tile");
  const bool vec = (e % 4 == 0);
  auto fits = [&](int ct, size_t budget) {
    return sizeof(float) * ((size_t)ct * ldw + 2 * kpad + ML_WAVES * 2 * ct + ML_WAVES * 16 * ML_TRLD) <= budget;
  };
  // widest channel tile whose weights leave room for a second workgroup on the CU (x is then
  // re-read from L2 as rarely as possible)
  // compact lists: a sample has a few hundred live tiles, i.e. one or two per wave -- the launch then takes as long as
  // ONE wave needs for a tile, and a 64-channel tile halves that (measured at the step's shapes, 11 % and 36 % / 4 % and
  // 16 % live: 39 / 19 / 24 us against 46 / 31 / 32 with 128-channel tiles; 100 / 29 / 48 against 98 / 36 / 56)
  if (tl_n_act != nullptr && cout > 64 && fits(64, 80 * 1024))
    return layer_fwd_tile<2>(pscale != nullptr, vec, b, cin, cout, e, x, w, pscale, pshift, y, stat_sum, stat_sq, stream);
  if (cout > 64 && fits(128, 80 * 1024))
    return layer_fwd_tile<4>(pscale != nullptr, vec, b, cin, cout, e, x, w, pscale, pshift, y, stat_sum, stat_sq, stream);
  if (cout > 32 && fits(64, 80 * 1024))
    return layer_fwd_tile<2>(pscale != nullptr, vec, b, cin, cout, e, x, w, pscale, pshift, y, stat_sum, stat_sq, stream);
  if (cout > 32 && fits(64, 160 * 1024))
    return layer_fwd_tile<2>(pscale != nullptr, vec, b, cin, cout, e, x, w, pscale, pshift, y, stat_sum, stat_sq, stream);
  return layer_fwd_tile<1>(pscale != nullptr, vec, b, cin, cout, e, x, w, pscale, pshift, y, stat_sum, stat_sq, stream);
}

extern "C" int sig3d_bn_finalize(int c, double count, float eps, float momentum,
                                 const double *stat_sum, const double *stat_sq, const float *gamma,
                                 const float *beta, float *scale, float *shift, float *save_mean,
                                 float *save_invstd, float *running_mean, float *running_var,
                                 long long *num_batches_tracked, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SIG3D_REQUIRE(c >= 1 && count >= 1.0, "bad size");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(sig3d_ceil_div(c, 256)), dim3(256), 0, stream, c, count,
                     eps, momentum, stat_sum, stat_sq, gamma, beta, scale, shift, save_mean,
                     save_invstd, running_mean, running_var, num_batches_tracked);
  SIG3D_LAUNCH_CHECK("bn_finalize_kernel");
  return 0;
}

extern "C" int sig3d_bn_relu_maxpool(int b, int c, int p, int s, const float *y, const float *scale,
                                     const float *shift, float *out, int *arg, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SIG3D_REQUIRE(b >= 0 && c >= 1 && p >= 0 && s >= 1, "bad size");
  const long groups = (long)b * c * p;
  if (groups == 0) return 0;
  SIG3D_REQUIRE(groups < (1L << 27), "b*c*p must be below 2^27 (32-bit thread indexing)");
  if (s == 64)
    hipLaunchKernelGGL(bn_relu_maxpool_vec_kernel<16>, dim3((unsigned)((groups * 16 + 255) / 256)), dim3(256),
                       0, stream, (unsigned)groups, c, p, y, scale, shift, out, arg);
  else if (s == 32)
    hipLaunchKernelGGL(bn_relu_maxpool_vec_kernel<8>, dim3((unsigned)((groups * 8 + 255) / 256)), dim3(256),
                       0, stream, (unsigned)groups, c, p, y, scale, shift, out, arg);
  else if (s == 16)
    hipLaunchKernelGGL(bn_relu_maxpool_vec_kernel<4>, dim3((unsigned)((groups * 4 + 255) / 256)), dim3(256),
                       0, stream, (unsigned)groups, c, p, y, scale, shift, out, arg);
  else
    hipLaunchKernelGGL(bn_relu_maxpool_kernel, dim3((unsigned)((groups * 16 + 255) / 256)), dim3(256), 0,
                       stream, groups, c, p, s, y, scale, shift, out, arg);
  SIG3D_LAUNCH_CHECK("bn_relu_maxpool_kernel");
  return 0;
}


extern "C" int sig3d_channel_stats(int b, int c, long e, const float *y, double *stat_sum,
                                   double *stat_sq, int accumulate, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SIG3D_REQUIRE(b >= 0 && c >= 1 && e >= 0 && e < (1L << 31), "bad size");
  if (int rc = zero_pair(stat_sum, stat_sq, c, accumulate, stream)) return rc;
  if (b == 0 || e == 0) return 0;
  dim3 grid((unsigned)((e + BNB_CHUNK - 1) / BNB_CHUNK), c, b);
  hipLaunchKernelGGL(channel_stats_kernel, grid, dim3(BNB_THREADS), 0, stream, c, e, y, stat_sum, stat_sq);
  SIG3D_LAUNCH_CHECK("channel_stats_kernel");
  return 0;
}

extern "C" int sig3d_bn_relu_apply(int b, int c, long e, const float *y, const float *scale,
                                   const float *shift, float *out, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SIG3D_REQUIRE(b >= 0 && c >= 1 && e >= 0 && e < (1L << 31), "bad size");
  if (b == 0 || e == 0) return 0;
  dim3 grid((unsigned)((e + BNB_CHUNK - 1) / BNB_CHUNK), c, b);
  hipLaunchKernelGGL(bn_relu_apply_kernel, grid, dim3(BNB_THREADS), 0, stream, c, e, y, scale, shift, out);
  SIG3D_LAUNCH_CHECK("bn_relu_apply_kernel");
  return 0;
}

// ---- backward entry points -----------------------------------------------------------------

extern "C" int sig3d_bn_relu_bwd(int b, int c, long e, int s, const float *dA, const float *dOut,
                                 const int *arg, const float *y, const float *scale,
                                 const float *shift, const float *mean, const float *invstd,
                                 double *s1, double *s2, float *dY, int accumulate, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SIG3D_REQUIRE(b >= 0 && c >= 1 && e >= 0 && s >= 1, "bad size");
  SIG3D_REQUIRE((dA != nullptr) != (dOut != nullptr && arg != nullptr),
                "pass either a dense dA or the (dOut, arg) pair of the max-pool");
  SIG3D_REQUIRE(e < (1L << 31) && e % s == 0, "positions per row must be a multiple of s and < 2^31");
  if (int rc = zero_pair(s1, s2, c, accumulate, stream)) return rc;
  if (b == 0 || e == 0) return 0;
  const double count = (double)b * (double)e;
  dim3 grid((unsigned)((e + BNB_CHUNK - 1) / BNB_CHUNK), c, b);
  if (dA) {
    hipLaunchKernelGGL((bn_relu_bwd_stats_kernel<false>), grid, dim3(BNB_THREADS), 0, stream, c, e, s,
                       dA, dOut, arg, y, scale, shift, mean, invstd, s1, s2);
    hipLaunchKernelGGL((bn_relu_bwd_apply_kernel<false>), grid, dim3(BNB_THREADS), 0, stream, c, e, s,
                       count, dA, dOut, arg, y, scale, shift, mean, invstd, s1, s2, dY);
  } else {
    const int P = (int)(e / s);
    dim3 tgrid(P >= 4 * BNB_THREADS ? 4 : 1, c, b);
    if (accumulate != 2)   // 2: sig3d_bn_relu_bwd_top_from_pm has taken the statistics already
      hipLaunchKernelGGL(bn_relu_bwd_top_stats_kernel, tgrid, dim3(BNB_THREADS), 0, stream, c, P, s, dOut,
                         arg, y, scale, shift, mean, invstd, s1, s2);
    hipLaunchKernelGGL((bn_relu_bwd_apply_kernel<true>), grid, dim3(BNB_THREADS), 0, stream, c, e, s,
                       count, dA, dOut, arg, y, scale, shift, mean, invstd, s1, s2, dY);
  }
  SIG3D_LAUNCH_CHECK("bn_relu_bwd kernels");
  return 0;
}

template <bool PROLOGUE, bool VEC, bool GATHER = false>
static int launch_mlp_dw(int b, int cin, int cout, long e, const float *dY, const float *x,
                         const float *pscale, const float *pshift, float *dW, hipStream_t stream) {
  const int nblk_m = sig3d_ceil_div(cout, 64), nblk_n = sig3d_ceil_div(cin, 64);
  const long n_steps = (e + 31) / 32;
  // ONE round of resident workgroups (2 per CU: 216 VGPRs, 68 KB of LDS images): longer position
  // ranges per wave amortise the reduction epilogue and leave no tail round; >= 8 steps per wave
  long wgs_per_block_row = (2L * 256) / ((long)b * nblk_m * nblk_n);
  if (wgs_per_block_row < 1) wgs_per_block_row = 1;
  long spw = (n_steps + wgs_per_block_row * DW_WAVES - 1) / (wgs_per_block_row * DW_WAVES);
  if (spw < 8) spw = 8;
  const long wgs = (n_steps + spw * DW_WAVES - 1) / (spw * DW_WAVES);
  dim3 grid((unsigned)wgs, nblk_m * nblk_n, b);
  hipLaunchKernelGGL((mlp_dw_kernel<PROLOGUE, VEC, GATHER>), grid, dim3(DW_WAVES * 64), 0, stream, cin, cout, e,
                     (int)spw, nblk_n, dY, x, pscale, pshift, dW, tl_n_act, GATHER ? *tl_gather : MlpGather{});
  SIG3D_LAUNCH_CHECK("mlp_dw_kernel");
  return 0;
}

extern "C" int sig3d_mlp_layer_dw(int b, int cin, int cout, long e, const float *dY, const float *x,
                                  const float *pscale, const float *pshift, float *dW, int accumulate,
                                  void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SIG3D_REQUIRE(b >= 0 && cin >= 1 && cout >= 1 && e >= 0, "bad size");
  SIG3D_REQUIRE((pscale == nullptr) == (pshift == nullptr), "pscale/pshift must come together");
  if (!accumulate) SIG3D_HIP_TRY(hipMemsetAsync(dW, 0, sizeof(float) * (size_t)cout * cin, stream));
  if (b == 0 || e == 0) return 0;
  const bool vec = (e % 32 == 0);  // every 16-position run is in range and 16-byte aligned
  if (tl_gather != nullptr && pscale == nullptr)
    return vec ? launch_mlp_dw<false, true, true>(b, cin, cout, e, dY, x, pscale, pshift, dW, stream)
               : launch_mlp_dw<false, false, true>(b, cin, cout, e, dY, x, pscale, pshift, dW, stream);
  if (pscale) return vec ? launch_mlp_dw<true, true>(b, cin, cout, e, dY, x, pscale, pshift, dW, stream)
                         : launch_mlp_dw<true, false>(b, cin, cout, e, dY, x, pscale, pshift, dW, stream);
  return vec ? launch_mlp_dw<false, true>(b, cin, cout, e, dY, x, pscale, pshift, dW, stream)
             : launch_mlp_dw<false, false>(b, cin, cout, e, dY, x, pscale, pshift, dW, stream);
}

// ---- compact-mode entry points (distinct neighbours only; compact.hip) ------------------------------------
extern "C" int sig3d_mlp_layer_fwd_compact(int b, int cin, int cout, long e, const float *x, const float *w,
                                           const float *pscale, const float *pshift, float *y, double *stat_sum,
                                           double *stat_sq, int accumulate, const int *n_act, const float *mult,
                                           void *stream_) {
  SIG3D_REQUIRE(n_act != nullptr && (stat_sum == nullptr || mult != nullptr), "n_act (and mult with statistics) required");
  tl_n_act = n_act;
  tl_mult = mult;
  const int rc = sig3d_mlp_layer_fwd(b, cin, cout, e, x, w, pscale, pshift, y, stat_sum, stat_sq, accumulate, stream_);
  tl_n_act = nullptr;
  tl_mult = nullptr;
  return rc;
}

// First layer of a set-abstraction stack with the grouped operand gathered on load (MlpGather above): same
// outputs as sig3d_query_group_fused_pm / _compact followed by sig3d_mlp_layer_fwd(_compact) on its result, up
// to the order of the f32 sums.  cin = 3 + c, c a multiple of 32; idx (b, e) i32 = the ball-query lists (or the
// compact lists: then centre_of, n_act and -- with statistics -- mult are given as for the *_compact calls).
extern "C" int sig3d_mlp_layer0_gather_fwd(int b, int n, int m, int nsample, int c, int cout, int normalize_xyz,
                                           float radius, const float *xyz, const float *new_xyz,
                                           const float *features_pm, const int *idx, const float *w, float *y,
                                           double *stat_sum, double *stat_sq, int accumulate, const int *centre_of,
                                           const int *n_act, const float *mult, void *stream_) {
  SIG3D_REQUIRE(b >= 0 && n >= 1 && m >= 0 && nsample >= 0 && cout >= 1, "bad size");
  SIG3D_REQUIRE(c >= 32 && c % 32 == 0, "the gathering first layer needs a multiple of 32 feature channels");
  SIG3D_REQUIRE(xyz && new_xyz && features_pm && idx && w && y, "null operand");
  SIG3D_REQUIRE((centre_of == nullptr) == (n_act == nullptr), "centre_of and n_act come together (compact lists)");
  SIG3D_REQUIRE(n_act == nullptr || stat_sum == nullptr || mult != nullptr, "statistics over compact lists need mult");
  MlpGather ga;
  ga.xyz = xyz; ga.centre = new_xyz; ga.feat_pm = features_pm; ga.idx = idx; ga.centre_of = centre_of;
  ga.N = n; ga.P = m; ga.S = nsample; ga.C = c; ga.normalize = normalize_xyz; ga.radius = radius;
  ga.scatter = nullptr;
  tl_gather = &ga;
  tl_n_act = n_act;
  tl_mult = mult;
  // x is unused by the gathering kernel; pass a valid pointer so that generic checks hold
  const int rc = sig3d_mlp_layer_fwd(b, c + 3, cout, (long)m * nsample, features_pm, w, nullptr, nullptr, y, stat_sum,
                                     stat_sq, accumulate, stream_);
  tl_gather = nullptr;
  tl_n_act = nullptr;
  tl_mult = nullptr;
  return rc;
}

// Backward of the gathering first layer.  dw: dW (cout, 3 + c) += dY X^T with X gathered on load.
extern "C" int sig3d_mlp_layer0_gather_dw(int b, int n, int m, int nsample, int c, int cout, int normalize_xyz,
                                          float radius, const float *xyz, const float *new_xyz,
                                          const float *features_pm, const int *idx, const float *dY, float *dW,
                                          int accumulate, const int *centre_of, const int *n_act, void *stream_) {
  SIG3D_REQUIRE(b >= 0 && n >= 1 && m >= 0 && nsample >= 0 && cout >= 1 && c >= 1, "bad size");
  SIG3D_REQUIRE(xyz && new_xyz && features_pm && idx && dY && dW, "null operand");
  SIG3D_REQUIRE((centre_of == nullptr) == (n_act == nullptr), "centre_of and n_act come together (compact lists)");
  MlpGather ga;
  ga.xyz = xyz; ga.centre = new_xyz; ga.feat_pm = features_pm; ga.idx = idx; ga.centre_of = centre_of;
  ga.N = n; ga.P = m; ga.S = nsample; ga.C = c; ga.normalize = normalize_xyz; ga.radius = radius;
  ga.scatter = nullptr;
  tl_gather = &ga;
  tl_n_act = n_act;
  const int rc = sig3d_mlp_layer_dw(b, c + 3, cout, (long)m * nsample, dY, features_pm, nullptr, nullptr, dW, accumulate,
                                    stream_);
  tl_gather = nullptr;
  tl_n_act = nullptr;
  return rc;
}

// dx: grad_features_pm (b, n, c) += scatter of W^T dY over the neighbour lists (wt = W^T, (3 + c, cout));
// the caller zeroes grad_features_pm.
extern "C" int sig3d_mlp_layer0_scatter_dx(int b, int n, int m, int nsample, int c, int cout, const int *idx,
                                           const float *dY, const float *wt, float *grad_features_pm,
                                           const int *n_act, void *stream_) {
  SIG3D_REQUIRE(b >= 0 && n >= 1 && m >= 0 && nsample >= 0 && cout >= 1 && c >= 1, "bad size");
  SIG3D_REQUIRE(idx && dY && wt && grad_features_pm, "null operand");
  MlpGather ga = {};
  ga.idx = idx; ga.N = n; ga.P = m; ga.S = nsample; ga.C = c; ga.scatter = grad_features_pm;
  tl_gather = &ga;
  tl_n_act = n_act;
  // the product's output rows are the 3 + c grouped channels; y is never written in scatter mode
  const int rc = sig3d_mlp_layer_fwd(b, cout, c + 3, (long)m * nsample, dY, wt, nullptr, nullptr, grad_features_pm,
                                     nullptr, nullptr, 0, stream_);
  tl_gather = nullptr;
  tl_n_act = nullptr;
  return rc;
}

// Input gradient of a layer: dA (b, cin, e) = W^T dY with W (cout, cin) AS STORED by the forward layer (the kernel
// stages it transposed; rounds 1-3 made a W^T copy per layer per step).  n_act: compact lists, or NULL.
extern "C" int sig3d_mlp_layer_dx(int b, int cin, int cout, long e, const float *dY, const float *w, float *dA,
                                  const int *n_act, void *stream_) {
  SIG3D_REQUIRE(dY && w && dA, "null operand");
  tl_n_act = n_act;
  tl_w_t = 1;
  const int rc = sig3d_mlp_layer_fwd(b, cout, cin, e, dY, w, nullptr, nullptr, dA, nullptr, nullptr, 0, stream_);
  tl_w_t = 0;
  tl_n_act = nullptr;
  return rc;
}

// A compact level's layer in the backward pass: its weight gradient (the k-streaming product of gemm16_core.h, without its
// fold) and its input gradient (the layer kernel above with the stored weight read transposed) share only dY, and each is
// 15-40 us of mostly latency for one or two tiles per wave -- in a row they cost the step both waits.  ONE launch: workgroups
// [0, n_dw) are the product's, the rest (their first four waves) the layer kernel's (xi, yi, zi) grid.
__global__ __launch_bounds__(512, 2) void mlp_dw_dx_kernel(const gemm16::Problem p, int n_dw, int gx, int gy, int cin_k,
                                                           int cout_k, long E, int tpw, const float *__restrict__ dY,
                                                           const float *__restrict__ w, float *__restrict__ dA,
                                                           const int *__restrict__ n_act) {
  if ((int)blockIdx.x < n_dw) {
    gemm16::gemm16_body<1, 2, 4, 2, 4, gemm16::B_KC, true, true>(p, (int)blockIdx.x, n_dw);
    return;
  }
  if (threadIdx.x >= ML_WAVES * 64) return;        // (the layer kernel is four waves; a barrier counts live waves only)
  const int t = (int)blockIdx.x - n_dw;
  mlp_layer_fwd_body<2, false, true, false, false, false>(t % gx, (t / gx) % gy, t / (gx * gy), gy, cin_k, cout_k, E, tpw, dY, w,
                                                         nullptr, nullptr, dA, nullptr, nullptr, n_act, nullptr, MlpGather{}, 1);
}

// dW (cout, cin) = sum dY a^T (a = x or relu(x * pscale + pshift), slabs NOT folded: sig3d_sum_slabs_multi) and
// dA (b, cin, e) = W^T dY, compact lists (n_act).  Falls back to the two launches when the shapes take other instances.
extern "C" int sig3d_mlp_layer_dw_dx(int b, int cin, int cout, long e, const float *dY, const float *x,
                                     const float *pscale, const float *pshift, const int *n_act, const float *w,
                                     float *dW, float *work, float *dA, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SIG3D_REQUIRE(b >= 0 && cin >= 1 && cout >= 1 && e >= 0 && dY && x && w && dW && dA && n_act, "bad arguments");
  gemm16::Problem p;
  int n_dw = 0, usable = 0;
  if (int rc = sig3d_internal_dw_stream_problem(b, cin, cout, e, dY, x, pscale, pshift, n_act, dW, work, &p, &n_dw, &usable))
    return rc;
  // the input gradient runs the layer kernel with (reduction, rows) = (cout, cin): the <2, .., VEC, !RAGGED> instance only
  const int cin_k = cout, cout_k = cin;
  const int kpad = ml_kpad(cin_k), ldw = kpad | 1;
  const size_t lds_mlp = sizeof(float) * ((size_t)64 * ldw + 2 * kpad + ML_WAVES * 2 * 64 + ML_WAVES * 16 * ML_TRLD);
  const bool ragged = (cin_k % (2 * ML_KC) != 0 && kpad - cin_k >= 2) || (cout_k % 64 != 0 && cout_k % 64 <= 32);
  if (!usable || ragged || cout_k <= 32 || lds_mlp > 80 * 1024 || e % 4 != 0 ||
      (long)cin_k * e >= (1L << 31) || (long)cout_k * e >= (1L << 31)) {
    if (int rc = sig3d_mlp_layer_dw_stream_nofold(b, cin, cout, e, dY, x, pscale, pshift, n_act, dW, work, stream_)) return rc;
    return sig3d_mlp_layer_dx(b, cin, cout, e, dY, w, dA, n_act, stream_);
  }
  // the layer kernel's grid as launch_mlp_fwd_g sizes it (one round of resident workgroups)
  const long wave_tiles = (e + 31) / 32;
  const int cblocks = sig3d_ceil_div(cout_k, 64);
  int occ = (int)((160 * 1024) / lds_mlp);
  if (occ > 3) occ = 3;
  if (occ < 1) occ = 1;
  long gy = (256L * occ + (long)b * cblocks - 1) / ((long)b * cblocks);
  const long gy_max = (wave_tiles + ML_WAVES - 1) / ML_WAVES;
  if (gy > gy_max) gy = gy_max;
  if (gy < 1) gy = 1;
  const int tpw = (int)((wave_tiles + gy * ML_WAVES - 1) / (gy * ML_WAVES));
  const size_t lds_dw = gemm16::lds_bytes<1, 2, 4, 2>();
  const size_t lds = lds_mlp > lds_dw ? lds_mlp : lds_dw;
  static sig3d_once_per_device attr_done;
  if (attr_done.pending()) {
    SIG3D_HIP_TRY(hipFuncSetAttribute((const void *)mlp_dw_dx_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    attr_done.done();
  }
  const long blocks = (long)n_dw + (long)cblocks * gy * b;
  hipLaunchKernelGGL(mlp_dw_dx_kernel, dim3((unsigned)blocks), dim3(512), lds, stream, p, n_dw, cblocks, (int)gy, cin_k, cout_k,
                     e, tpw, dY, w, dA, n_act);
  SIG3D_LAUNCH_CHECK("mlp_dw_dx_kernel");
  return 0;
}

// sig3d_mlp_layer0_scatter_dx with the weight as stored, w (cout, 3 + c)
extern "C" int sig3d_mlp_layer0_scatter_dx_w(int b, int n, int m, int nsample, int c, int cout, const int *idx,
                                             const float *dY, const float *w, float *grad_features_pm,
                                             const int *n_act, void *stream_) {
  tl_w_t = 1;
  const int rc = sig3d_mlp_layer0_scatter_dx(b, n, m, nsample, c, cout, idx, dY, w, grad_features_pm, n_act, stream_);
  tl_w_t = 0;
  return rc;
}

extern "C" int sig3d_mlp_layer_dw_compact(int b, int cin, int cout, long e, const float *dY, const float *x,
                                          const float *pscale, const float *pshift, float *dW, int accumulate,
                                          const int *n_act, void *stream_) {
  SIG3D_REQUIRE(n_act != nullptr, "n_act required");
  tl_n_act = n_act;
  const int rc = sig3d_mlp_layer_dw(b, cin, cout, e, dY, x, pscale, pshift, dW, accumulate, stream_);
  tl_n_act = nullptr;
  return rc;
}

extern "C" int sig3d_bn_relu_maxpool_compact(int b, int c, int p, long e, const float *y, const float *scale,
                                             const float *shift, const int *seg_off, float *out, int *arg,
                                             void *stream_) {
  SIG3D_REQUIRE(b >= 0 && c >= 1 && p >= 0 && e >= 0 && seg_off != nullptr, "bad size");
  if (b == 0 || p == 0) return 0;
  hipLaunchKernelGGL(bn_relu_maxpool_seg_kernel, dim3(sig3d_ceil_div(p, 256), c, b), dim3(256), 0,
                     (hipStream_t)stream_, c, p, e, y, scale, shift, seg_off, out, arg);
  SIG3D_LAUNCH_CHECK("bn_relu_maxpool_seg_kernel");
  return 0;
}

extern "C" int sig3d_bn_relu_maxpool_pm(int b, int c, int p, int s, long e, const float *y, const float *scale,
                                        const float *shift, const int *seg_off, float *out, int *arg, float *out_pm,
                                        void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SIG3D_REQUIRE(b >= 0 && c >= 1 && p >= 0 && out_pm != nullptr, "bad size");
  if (b == 0 || p == 0) return 0;
  if (seg_off != nullptr) {
    SIG3D_REQUIRE(e >= 0, "bad size");
    hipLaunchKernelGGL(bn_relu_maxpool_seg_pm_kernel, dim3(sig3d_ceil_div(p, 16), sig3d_ceil_div(c, PM_TC), b), dim3(256),
                       0, stream, c, p, e, y, scale, shift, seg_off, out, arg, out_pm);
  } else if (s == 64) {
    hipLaunchKernelGGL(bn_relu_maxpool_pm_kernel<16>, dim3(sig3d_ceil_div(p, 16), sig3d_ceil_div(c, PM_TC), b), dim3(256), 0,
                       stream, c, p, y, scale, shift, out, arg, out_pm);
  } else if (s == 32) {
    hipLaunchKernelGGL(bn_relu_maxpool_pm_kernel<8>, dim3(sig3d_ceil_div(p, 32), sig3d_ceil_div(c, PM_TC), b), dim3(256), 0,
                       stream, c, p, y, scale, shift, out, arg, out_pm);
  } else if (s == 16) {
    hipLaunchKernelGGL(bn_relu_maxpool_pm_kernel<4>, dim3(sig3d_ceil_div(p, 64), sig3d_ceil_div(c, PM_TC), b), dim3(256), 0,
                       stream, c, p, y, scale, shift, out, arg, out_pm);
  } else {   // other neighbourhood sizes: the generic pooling kernel, then one transpose
    if (int rc = sig3d_bn_relu_maxpool(b, c, p, s, y, scale, shift, out, arg, stream_)) return rc;
    return sig3d_transpose_cn(b, c, p, out, out_pm, stream_);
  }
  SIG3D_LAUNCH_CHECK("bn_relu_maxpool_pm_kernel");
  return 0;
}

extern "C" int sig3d_bn_relu_bwd_top_from_pm(int b, int c, int p, int s, long e, const float *dout_pm, const int *arg,
                                             const float *y, const float *scale, const float *shift,
                                             const float *mean, const float *invstd, const int *seg_off,
                                             float *dout_cm, double *s1, double *s2, int accumulate, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SIG3D_REQUIRE(b >= 0 && c >= 1 && p >= 0 && s >= 1 && e >= 0, "bad size");
  SIG3D_REQUIRE(dout_pm && arg && y && dout_cm && s1 && s2, "null argument");
  if (int rc = zero_pair(s1, s2, c, accumulate, stream)) return rc;
  if (b == 0 || p == 0) return 0;
  hipLaunchKernelGGL(bn_relu_bwd_top_from_pm_kernel, dim3(sig3d_ceil_div(p, 64), sig3d_ceil_div(c, PM_TC), b), dim3(256), 0,
                     stream, c, p, s, e, dout_pm, arg, y, scale, shift, mean, invstd, seg_off, dout_cm, s1, s2);
  SIG3D_LAUNCH_CHECK("bn_relu_bwd_top_from_pm_kernel");
  return 0;
}

extern "C" int sig3d_bn_relu_bwd_compact(int b, int c, long e, int p, const float *dA, const float *dOut,
                                         const int *arg, const float *y, const float *scale, const float *shift,
                                         const float *mean, const float *invstd, double *s1, double *s2,
                                         float *dY, int accumulate, const int *n_act, const float *mult,
                                         const int *centre_of, const int *seg_off, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SIG3D_REQUIRE(b >= 0 && c >= 1 && e >= 0 && p >= 1, "bad size");
  SIG3D_REQUIRE((dA != nullptr) != (dOut != nullptr && arg != nullptr),
                "pass either a dense dA or the (dOut, arg) pair of the max-pool");
  SIG3D_REQUIRE(n_act && mult && centre_of && seg_off, "compact lists missing");
  SIG3D_REQUIRE(e < (1L << 31), "positions per row must stay below 2^31");
  if (int rc = zero_pair(s1, s2, c, accumulate, stream)) return rc;
  if (b == 0 || e == 0) return 0;
  const double count = (double)b * (double)e;  // the statistics are over ALL columns, padded ones included
  // compact rows are mostly empty behind n_act[b]: BNB_CGRID workgroups per row, each walking chunks (kernels above)
  const unsigned chunks = (unsigned)((e + BNB_CHUNK - 1) / BNB_CHUNK);
  dim3 grid(chunks < (unsigned)BNB_CGRID ? chunks : (unsigned)BNB_CGRID, c, b);
  if (dA) {
    hipLaunchKernelGGL((bn_relu_bwd_stats_c_kernel<false>), grid, dim3(BNB_THREADS), 0, stream, c, e, p, dA, dOut, arg,
                       y, scale, shift, mean, invstd, n_act, centre_of, seg_off, s1, s2);
    hipLaunchKernelGGL((bn_relu_bwd_apply_c_kernel<false>), grid, dim3(BNB_THREADS), 0, stream, c, e, p, count, dA,
                       dOut, arg, y, scale, shift, mean, invstd, n_act, mult, centre_of, seg_off, s1, s2, dY);
  } else {
    dim3 ggrid(sig3d_ceil_div(p, BNB_THREADS), c, b);
    if (accumulate != 2)   // 2: sig3d_bn_relu_bwd_top_from_pm has taken the statistics already
      hipLaunchKernelGGL(bn_relu_bwd_top_stats_c_kernel, dim3(p >= 4 * BNB_THREADS ? 4 : 1, c, b), dim3(BNB_THREADS), 0,
                         stream, c, e, p, dOut, arg, y, scale, shift, mean, invstd, seg_off, s1, s2);
    hipLaunchKernelGGL(bn_relu_bwd_top_sweep_c_kernel, grid, dim3(BNB_THREADS), 0, stream, c, e, count, y, scale, mean,
                       invstd, n_act, mult, s1, s2, dY);
    hipLaunchKernelGGL(bn_relu_bwd_top_fix_c_kernel, ggrid, dim3(BNB_THREADS), 0, stream, c, e, p, dOut, arg, y, scale,
                       shift, seg_off, dY);
  }
  SIG3D_LAUNCH_CHECK("bn_relu_bwd compact kernels");
  return 0;
}
