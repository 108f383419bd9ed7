// shared_mlp_fwd.h -- the SharedMLP layer kernel (forward product, and the input gradient as the same product over
// the transposed weight) with its launch.  Included by shared_mlp.hip, whose mlp_dw_dx_kernel embeds the kernel's body,
// and by shared_mlp_fwd1/2/4/4r.hip, which hold the kernel's instances of one channel-tile width each (the widest split
// again into its ragged and full-tile halves): in one translation unit the 48 instances took 110 s to compile and alone
// set the wall time of the build.
#pragma once
#include <cstdlib>

#include "sig3d_common.h"

struct MlpGather {
  const float *xyz, *centre, *feat_pm;
  const int *idx, *centre_of;   // centre_of: compact lists only (else centre = e / S)
  int N, P, S, C, normalize;
  float radius;
  float *scatter;               // input-gradient use: add the tile into this point-major (B, N, C) gradient
};

// what the *_compact / gather / dx entry points add to a plain layer call
struct MlpFwdCall {
  const int *n_act;          // compact lists: live columns per sample, or NULL
  const float *mult;         // compact lists: multiplicity of a column in the statistics
  const MlpGather *gather;   // first layer gathered on load / input gradient scattered, or NULL
  int w_t;                   // the weight operand is given transposed (sig3d_mlp_layer_dx)
};

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int ML_WAVES = 4;  // waves per workgroup, one 32-position tile each per sweep
constexpr int ML_KC = 16;    // K-steps (of 2 input channels) per software-pipeline chunk
constexpr int ML_TRLD = 36;  // row stride of the epilogue transpose buffer (16-byte aligned rows)

// Phase timing for tools/mlp_timing.py (only with -DSIG3D_MLP_TIMING): wave 0 of workgroup
// (0,0,0) appends the 100 MHz real-time counter at every mark.
#ifdef SIG3D_MLP_TIMING
__device__ unsigned long long g_ml_marks[64];
__device__ unsigned long long g_ml_cycles[64];   // shader-clock counter at the same marks
__device__ int g_ml_nmarks;
#define ML_MARK(id)                                                                              \
  do {                                                                                           \
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) {             \
      const int n_ = g_ml_nmarks;                                                                \
      if (n_ < 64) { g_ml_marks[n_] = ((unsigned long long)(id) << 56) | __builtin_amdgcn_s_memrealtime(); g_ml_cycles[n_] = __builtin_readcyclecounter(); g_ml_nmarks = n_ + 1; } \
    }                                                                                            \
  } while (0)
#else
#define ML_MARK(id) do { } while (0)
#endif

__device__ __forceinline__ int mrow(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

__device__ __forceinline__ float row32_sum(float v) {
  // sum over the 32 lanes that share (lane >> 5): two DPP rows of 16, then the neighbour row
  v = row_allreduce_sum_f32(v);
  v += __shfl_xor(v, 16);
  return v;
}

__host__ __device__ inline int ml_kpad(int cin) { return (cin + 2 * ML_KC - 1) / (2 * ML_KC) * (2 * ML_KC); }

// ---- forward layer -------------------------------------------------------------------------
// grid (cout / (32*NT), position chunks, B); dynamic LDS: W tile [32*NT][ldw] + pscale[kpad] +
// pshift[kpad] + cross-wave stat scratch.
// MFMA orientation: D[i = position][j = channel] = sum_k a[k][position] * W[channel][k]
// (A operand = activations, B operand = weights).  With channels on LANES, the per-channel batch
// statistics are plain in-lane sums over the 16 accumulator registers (2 registers per 32-channel
// tile, instead of a 64-register lane-local image), and four consecutive accumulator registers
// are four consecutive positions of one channel row -> 16-byte stores.
// The A operand (one activation value per lane per K-step, a coalesced 128-byte row segment per
// half-wave) is software-pipelined in chunks of ML_KC steps: the loads of chunk c+1 are in flight
// while the MFMAs of chunk c issue.
// GATHER (first layer of a set-abstraction stack): the activation operand is not a
// stored (B, 3 + C, npoint, nsample) tensor but is gathered while it is loaded -- position e of batch b is the
// neighbour idx[b][e]: C feature channels from its row of the POINT-MAJOR feature copy (B, N, C) (16 floats
// per lane per chunk, four 16-byte loads from one row) and 3 channels xyz[idx] - centre (/ radius), exactly
// QueryAndGroup's arithmetic (pointnet2_utils.py:348-359, csrc/group_points.hip).  The reduction index is
// permuted so that those loads are contiguous: LDS weight column c' holds the weight of feature channel c'
// (column 3 + c' of the layer's weight) for c' < C and of xyz channel c' - C behind them; K-step i of chunk c
// pairs channels 32c + i and 32c + 16 + i.  Sums are re-associated relative to the stored-tensor kernel
// (f32, far inside 1e-4); nothing (B, 3 + C, P, S)-shaped is written or read.

// (the body takes its workgroup's coordinates and the grid's y extent as arguments: mlp_dw_dx_kernel below runs it as a
// workgroup range of a launch that also carries the layer's weight-gradient product)
template <int NT, bool PROLOGUE, bool VEC, bool RAGGED, bool GATHER, bool SCATTER>
__device__ __forceinline__ void mlp_layer_fwd_body(
    const int blk_x, const int blk_y, const int blk_z, const int grid_y,
    int cin, int cout, long E, int tiles_per_wave, const float *__restrict__ x,
    const float *__restrict__ w, const float *__restrict__ pscale, const float *__restrict__ pshift,
    float *__restrict__ y, double *__restrict__ stat_sum, double *__restrict__ stat_sq,
    const int *__restrict__ n_act, const float *__restrict__ mult, const MlpGather &ga, int w_t) {
  // Compact mode (n_act given, compact.hip): only the first n_act[b] positions of every row exist -- the
  // distinct neighbours -- and position u stands for mult[b][u] equal columns: the statistics are weighted.
  // E stays the row stride; En is the number of positions.
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int CT = 32 * NT;
  const int kpad = ml_kpad(cin);
  const int ldw = kpad | 1;  // odd stride: one weight row per lane -> distinct banks
  float *s_w = smem;                // [CT][ldw]
  float *s_ps = s_w + CT * ldw;     // [kpad]
  float *s_pb = s_ps + kpad;        // [kpad]
  float *s_red = s_pb + kpad;       // [ML_WAVES][2][CT]

  const int lane = lane_id(), l31 = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int co0 = blk_x * CT;
  const int bi = blk_z;
  const int nt_act = min(NT, (cout - co0 + 31) / 32);  // 32-channel tiles of this block that hold channels
  float *s_tr = s_red + ML_WAVES * 2 * CT + wave * (16 * ML_TRLD);  // [ML_WAVES][16][ML_TRLD], wave-private

  // compact mode: the grid is sized for the dense row; workgroups without a tile leave before staging weights
  if (n_act != nullptr && (long)blk_y * ML_WAVES * 32 >= (long)n_act[bi]) return;
  ML_MARK(0);
  // weight tile -> LDS: a wave takes 8 rows at a time and issues their (clamped, unconditional)
  // loads together; one load -> wait -> ds_write per element took 20 us per workgroup at 128x128
  // w_t: the operand is the TRANSPOSE of the stored matrix -- an input-gradient product dA = W^T dY reads the forward
  // layer's weight W (cin x cout here, row length cout) as it lies, with the lanes along its rows (no W^T copy per step)
  if (w_t) {
    for (int c0 = wave * 8; c0 < kpad; c0 += ML_WAVES * 8) {
      for (int r = lane; r < CT; r += 64) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
          v[j] = w[(size_t)min(c0 + j, cin - 1) * cout + min(co0 + r, cout - 1)];
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (c0 + j < kpad) s_w[r * ldw + c0 + j] = (c0 + j < cin && co0 + r < cout) ? v[j] : 0.f;
      }
    }
  } else
  for (int r0 = wave * 8; r0 < CT; r0 += ML_WAVES * 8) {
    for (int c = lane; c < kpad; c += 64) {
      float v[8];
      // GATHER: LDS column c <- weight column of the channel that reduction slot c stands for
      const int wc = GATHER ? (c < ga.C ? c + 3 : min(c - ga.C, 2)) : min(c, cin - 1);
#pragma unroll
      for (int j = 0; j < 8; ++j)
        v[j] = w[(size_t)min(co0 + r0 + j, cout - 1) * cin + wc];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        s_w[(r0 + j) * ldw + c] = (c < cin && co0 + r0 + j < cout) ? v[j] : 0.f;
    }
  }
  for (int i = threadIdx.x; i < kpad; i += ML_WAVES * 64) {
    s_ps[i] = (PROLOGUE && i < cin) ? pscale[i] : 1.f;
    s_pb[i] = (PROLOGUE && i < cin) ? pshift[i] : 0.f;
  }
  __syncthreads();
  ML_MARK(1);
  const float *xb = x + (size_t)bi * cin * E;
  float *yb = y + (size_t)bi * cout * E;
  const long En = n_act ? (long)n_act[bi] : E;
  const float *mb = mult ? mult + (size_t)bi * E : nullptr;
  float s1[NT], s2[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) { s1[nt] = 0.f; s2[nt] = 0.f; }

  const int nchunks = kpad / (2 * ML_KC);
  // Tiles of 32 positions are dealt round-robin over all waves of the (batch, channel-block) row
  // of the grid: wave-global index wg = blockIdx.y*ML_WAVES + wave owns tiles wg, wg + stride, ...
  // The launcher sizes gridDim.y so that the grid is ONE full round of resident workgroups
  // (weights staged once per workgroup, no tail round, neighbouring waves stream neighbouring tiles).
  const long n_tiles = (En + 31) / 32;
  const long tile0 = (long)blk_y * ML_WAVES + wave;
  const long tile_stride = (long)grid_y * ML_WAVES;
  int my_tiles = 0;
  for (int t = 0; t < tiles_per_wave; ++t)
    if (tile0 + (long)t * tile_stride < n_tiles) my_tiles = t + 1;
  auto e0_of = [&](int t) { return (tile0 + (long)t * tile_stride) * 32; };
  // The wave's work is ONE stream of (tile, K-chunk) items g = tile*nchunks + chunk with the
  // operand loads running TWO items ahead of the MFMAs, across tile boundaries: a chunk is only
  // ~0.9-1.7 us of MFMA work while a load takes ~2 us to return, and the first chunk of every
  // tile used to be fetched with nothing to overlap it (MFMA pipe < 50 % busy at SA2).
  const int total = my_tiles * nchunks;
  f32x16 acc[NT];

  auto issue = [&](float (&buf)[ML_KC], int g) {
    const int t = g / nchunks, c = g - t * nchunks;
    // 32-bit offsets from the uniform batch base (launcher: cin*E, cout*E < 2^31): one VGPR and
    // one multiply-add per address instead of a 64-bit pair
    const long e = e0_of(t) + l31;
    const unsigned eo = (unsigned)(e < En ? e : En - 1);
    if constexpr (GATHER) {
      const size_t pos = (size_t)bi * E + eo;
      const int gi = ga.idx[pos];
      if (c * 32 < ga.C) {   // feature chunk (uniform): 16 consecutive channels of the neighbour's row
        const float4 *p4 = reinterpret_cast<const float4 *>(ga.feat_pm + ((size_t)bi * ga.N + gi) * ga.C + c * 32 + 16 * half);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 v4 = p4[q];
          buf[4 * q + 0] = v4.x; buf[4 * q + 1] = v4.y; buf[4 * q + 2] = v4.z; buf[4 * q + 3] = v4.w;
        }
      } else {               // the xyz chunk: slots 0..2 of the lower half-wave
        const int ctr = ga.centre_of ? ga.centre_of[pos] : (int)(eo / (unsigned)ga.S);
        const float *pt = ga.xyz + ((size_t)bi * ga.N + gi) * 3;
        const float *cc = ga.centre + ((size_t)bi * ga.P + ctr) * 3;
#pragma unroll
        for (int i = 0; i < ML_KC; ++i) buf[i] = 0.f;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          float d = __fsub_rn(pt[j], cc[j]);
          if (ga.normalize) d = __fdiv_rn(d, ga.radius);
          buf[j] = half == 0 ? d : 0.f;
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < ML_KC; ++i) {
        const int k = (c * ML_KC + i) * 2 + half;
        buf[i] = xb[(unsigned)min(k, cin - 1) * (unsigned)E + eo];
      }
    }
  };
  auto consume = [&](const float (&buf)[ML_KC], int g) {
    const int t = g / nchunks, c = g - t * nchunks;
    ML_MARK(2);
    if (c == 0) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x16){0};
    }
    // SCATTER: the 16 neighbour indices of this lane's accumulator positions are requested before the last
    // chunk's MFMAs, so that the atomics of the epilogue do not start with a memory round trip per group
    int sgi[16];
    if constexpr (SCATTER) {
      if (c == nchunks - 1) {
        const int *ip = ga.idx + (size_t)bi * E;
        const long e0s = e0_of(t);
#pragma unroll
        for (int r = 0; r < 16; ++r) sgi[r] = ip[min(e0s + 8 * (r >> 2) + 4 * half + (r & 3), En - 1)];
      }
    }
    // Two bodies.  The common one is branch-free straight-line code (any per-step branch cuts the
    // chunk into basic blocks and exposes the LDS operand latency of every step: +25 % measured).
    // The ragged one -- last K chunk with padding steps (cin = 131 pads to 160, 259 to 288) or a
    // channel block with empty 32-channel tiles (cout = 131: the second block holds 3 channels) --
    // skips the padding MFMAs under uniform predicates.
    // (RAGGED is a template parameter: compiled into the common kernel, the second body costs the
    // NT = 4 variant its third wave per SIMD.)
    const bool ragged = RAGGED && (nt_act < NT || (c + 1) * ML_KC * 2 > cin + 1);
    if (!ragged) {
#pragma unroll
      for (int i = 0; i < ML_KC; ++i) {
        const int k = GATHER ? c * 32 + 16 * half + i : (c * ML_KC + i) * 2 + half;
        float a = buf[i];
        if (PROLOGUE) a = fmaxf(0.f, a * s_ps[k] + s_pb[k]);  // BN(prev) + ReLU on load
        a = (k < cin) ? a : 0.f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, s_w[(nt * 32 + l31) * ldw + k], acc[nt], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < ML_KC; ++i) {
        if (GATHER ? (c * 32 + i < cin) : ((c * ML_KC + i) * 2 < cin)) {
          const int k = GATHER ? c * 32 + 16 * half + i : (c * ML_KC + i) * 2 + half;
          float a = buf[i];
          if (PROLOGUE) a = fmaxf(0.f, a * s_ps[k] + s_pb[k]);
          a = (k < cin) ? a : 0.f;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            if (nt < nt_act)
              acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, s_w[(nt * 32 + l31) * ldw + k], acc[nt], 0, 0, 0);
        }
      }
    }
    ML_MARK(3);
    if (c != nchunks - 1) return;
    // epilogue: acc[nt][4g..4g+3] = positions e0 + 8g + 4*half + (0..3) of channel co0+32nt+l31
    const long e0 = e0_of(t);
    const bool full = e0 + 32 <= En;
    if constexpr (SCATTER) {
      // Input gradient of a gathering first layer: row co of this product is d(loss)/d(grouped channel co);
      // channel co >= 3 is feature co - 3 of neighbour idx[e], so the tile is ADDED into the point-major
      // feature gradient right here (32 lanes = 32 consecutive floats of one row per atomic instruction) and
      // the (B, 3 + C, P, S) gradient tensor never exists.  The xyz rows (co < 3) carry no gradient
      // (QueryAndGroup's xyz is not differentiated: pointnet2_utils.py:334).
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int co = co0 + nt * 32 + l31;
        if (co >= 3 && co < cout) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (e0 + 8 * (r >> 2) + 4 * half + (r & 3) < En)
              unsafeAtomicAdd(ga.scatter + ((size_t)bi * ga.N + sgi[r]) * ga.C + (co - 3), acc[nt][r]);
        }
      }
      return;
    }
    // (multiplicities are fetched where they are used, compact mode only: a 16-register array here cost the
    // dense NT = 4 instances their third wave per SIMD)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int co = co0 + nt * 32 + l31;
      if (co < cout) {
        float *yrow = yb + (size_t)co * E + e0 + 4 * half;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const float v0 = acc[nt][4 * g4], v1 = acc[nt][4 * g4 + 1], v2 = acc[nt][4 * g4 + 2], v3 = acc[nt][4 * g4 + 3];
          if (VEC && full) {
            if (mb) {
              const float4 m4 = *reinterpret_cast<const float4 *>(mb + e0 + 8 * g4 + 4 * half);
              const float w0 = m4.x, w1 = m4.y, w2 = m4.z, w3 = m4.w;
              s1[nt] += (w0 * v0 + w1 * v1) + (w2 * v2 + w3 * v3);
              s2[nt] += (w0 * v0 * v0 + w1 * v1 * v1) + (w2 * v2 * v2 + w3 * v3 * v3);
            } else {
              s1[nt] += (v0 + v1) + (v2 + v3);
              s2[nt] += (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
            }
          } else {
            const float vv[4] = {v0, v1, v2, v3};
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (e0 + 8 * g4 + 4 * half + q < En) {
                const float wq = mb ? mb[e0 + 8 * g4 + 4 * half + q] : 1.f;
                yrow[8 * g4 + q] = vv[q];
                s1[nt] += wq * vv[q];
                s2[nt] += wq * vv[q] * vv[q];
              }
          }
        }
      }
    }
    if (VEC && full) {
      // Stores through a wave-private LDS transpose.  In the accumulator layout a lane owns ONE
      // channel row, so a direct store instruction scatters 64 separate 16-byte pieces over 64
      // rows and every 128-byte line of y is assembled from 8 partial writes (2.9 TB/s at SA1
      // L3, epilogues of up to 12 us).  Transposed, 8 lanes write one complete 128-byte row
      // segment and an instruction covers 8 full lines.
      const int trow = lane >> 3, tcol = (lane & 7) * 4;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        if (RAGGED && nt >= nt_act) continue;
#pragma unroll
        for (int hrow = 0; hrow < 2; ++hrow) {  // 16 channel rows per pass
          if ((l31 >> 4) == hrow) {
            // registers 4g..4g+3 are four consecutive positions 8g + 4*half + (0..3): 16-byte LDS writes
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
              *reinterpret_cast<float4 *>(s_tr + (l31 & 15) * ML_TRLD + 8 * g4 + 4 * half) =
                  make_float4(acc[nt][4 * g4], acc[nt][4 * g4 + 1], acc[nt][4 * g4 + 2], acc[nt][4 * g4 + 3]);
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int row = trow + 8 * j;
            const int co = co0 + nt * 32 + hrow * 16 + row;
            const float4 v = *reinterpret_cast<const float4 *>(s_tr + row * ML_TRLD + tcol);
            if (co < cout) *reinterpret_cast<float4 *>(yb + ((unsigned)co * (unsigned)E + (unsigned)e0 + tcol)) = v;
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
      }
    }
    ML_MARK(4);
  };

  if (NT >= 4) {
    // NT = 4: a chunk is 64 MFMAs (~4.5 us with the pipe shared), one chunk of lookahead covers the
    // load latency, and the 16 registers of a third buffer are what separates 2 from 3 waves per SIMD
    float b0[ML_KC], b1[ML_KC];
    if (total > 0) issue(b0, 0);
    for (int g = 0; g < total; g += 2) {
      if (g + 1 < total) issue(b1, g + 1);
      consume(b0, g);
      if (g + 1 < total) {
        if (g + 2 < total) issue(b0, g + 2);
        consume(b1, g + 1);
      }
    }
  } else {
    float b0[ML_KC], b1[ML_KC], b2[ML_KC];
    if (total > 0) issue(b0, 0);
    if (total > 1) issue(b1, 1);
    for (int g = 0; g < total; g += 3) {
      if (g + 2 < total) issue(b2, g + 2);
      consume(b0, g);
      if (g + 1 < total) {
        if (g + 3 < total) issue(b0, g + 3);
        consume(b1, g + 1);
      }
      if (g + 2 < total) {
        if (g + 4 < total) issue(b1, g + 4);
        consume(b2, g + 2);
      }
    }
  }

  if (stat_sum == nullptr) return;  // input-gradient use: no statistics wanted (uniform branch)
  // per-channel partial sums: the two half-waves hold disjoint position sets of the same channel
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const float a = s1[nt] + __shfl_xor(s1[nt], 32);
    const float q = s2[nt] + __shfl_xor(s2[nt], 32);
    if (half == 0) {
      s_red[(wave * 2 + 0) * CT + nt * 32 + l31] = a;
      s_red[(wave * 2 + 1) * CT + nt * 32 + l31] = q;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * CT; i += ML_WAVES * 64) {
    const int which = i / CT, co = i % CT;
    if (co0 + co < cout) {
      double tot = 0.0;
#pragma unroll
      for (int wv = 0; wv < ML_WAVES; ++wv) tot += (double)s_red[(wv * 2 + which) * CT + co];
      unsafeAtomicAdd((which ? stat_sq : stat_sum) + co0 + co, tot);
    }
  }
}

template <int NT, bool PROLOGUE, bool VEC, bool RAGGED, bool GATHER, bool SCATTER>
__global__ __launch_bounds__(ML_WAVES * 64, 2) void mlp_layer_fwd_kernel(
    int cin, int cout, long E, int tiles_per_wave, const float *__restrict__ x,
    const float *__restrict__ w, const float *__restrict__ pscale, const float *__restrict__ pshift,
    float *__restrict__ y, double *__restrict__ stat_sum, double *__restrict__ stat_sq,
    const int *__restrict__ n_act, const float *__restrict__ mult, MlpGather ga, int w_t) {
  mlp_layer_fwd_body<NT, PROLOGUE, VEC, RAGGED, GATHER, SCATTER>((int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z,
                                                                 (int)gridDim.y, cin, cout, E, tiles_per_wave, x, w, pscale,
                                                                 pshift, y, stat_sum, stat_sq, n_act, mult, ga, w_t);
}

}  // namespace

template <int NT, bool PROLOGUE, bool VEC, bool RAGGED, bool GATHER, bool SCATTER = false>
static int launch_mlp_fwd_g(const MlpFwdCall &call, int b, int cin, int cout, long e, const float *x, const float *w,
                            const float *pscale, const float *pshift, float *y, double *stat_sum,
                            double *stat_sq, hipStream_t stream) {
  constexpr int CT = 32 * NT;
  const int kpad = ml_kpad(cin), ldw = kpad | 1;
  const size_t lds = sizeof(float) * ((size_t)CT * ldw + 2 * kpad + ML_WAVES * 2 * CT + ML_WAVES * 16 * ML_TRLD);
  static sig3d_once_per_device attr_done;  // per template instance
  if (attr_done.pending()) {
    SIG3D_HIP_TRY(hipFuncSetAttribute((const void *)mlp_layer_fwd_kernel<NT, PROLOGUE, VEC, RAGGED, GATHER, SCATTER>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done.done();
  }
  // one full round of resident workgroups: occupancy per CU from LDS and registers (168 / 134 /
  // 120 VGPRs for NT = 4 / 2 / 1 -> 3 / 3 / 4 waves per SIMD), 256 CUs
  const long wave_tiles = (e + 31) / 32;
  const int cblocks = sig3d_ceil_div(cout, CT);
  int occ = (int)((160 * 1024) / lds);
  const int occ_regs = NT == 1 ? 4 : 3;
  if (occ > occ_regs) occ = occ_regs;
  if (occ < 1) occ = 1;
  long gy = (256L * occ + (long)b * cblocks - 1) / ((long)b * cblocks);
  const long gy_max = (wave_tiles + ML_WAVES - 1) / ML_WAVES;  // at least one tile per wave
  if (gy > gy_max) gy = gy_max;
  if (gy < 1) gy = 1;
  const int tpw = (int)((wave_tiles + gy * ML_WAVES - 1) / (gy * ML_WAVES));
  dim3 grid(cblocks, (unsigned)gy, b);
  hipLaunchKernelGGL((mlp_layer_fwd_kernel<NT, PROLOGUE, VEC, RAGGED, GATHER, SCATTER>), grid, dim3(ML_WAVES * 64), lds, stream,
                     cin, cout, e, tpw, x, w, pscale, pshift, y, stat_sum, stat_sq, call.n_act, call.mult,
                     call.gather ? *call.gather : MlpGather{}, (GATHER ? 0 : call.w_t));
  SIG3D_LAUNCH_CHECK("mlp_layer_fwd_kernel");
  return 0;
}

template <int NT, bool PROLOGUE, bool VEC, bool RAGGED>
static int launch_mlp_fwd(const MlpFwdCall &call, int b, int cin, int cout, long e, const float *x, const float *w,
                          const float *pscale, const float *pshift, float *y, double *stat_sum,
                          double *stat_sq, hipStream_t stream) {
  if constexpr (!PROLOGUE) {   // the gathering operand load exists for first layers only
    if (call.gather != nullptr && call.gather->scatter == nullptr)
      return launch_mlp_fwd_g<NT, PROLOGUE, VEC, RAGGED, true>(call, b, cin, cout, e, x, w, pscale, pshift, y, stat_sum,
                                                              stat_sq, stream);
    if (call.gather != nullptr)
      return launch_mlp_fwd_g<NT, PROLOGUE, VEC, RAGGED, false, true>(call, b, cin, cout, e, x, w, pscale, pshift, y,
                                                                     stat_sum, stat_sq, stream);
  }
  return launch_mlp_fwd_g<NT, PROLOGUE, VEC, RAGGED, false>(call, b, cin, cout, e, x, w, pscale, pshift, y, stat_sum,
                                                            stat_sq, stream);
}

template <int NT, bool RAGGED>
static int dispatch_mlp_fwd2(const MlpFwdCall &call, bool prologue, bool vec, int b, int cin, int cout, long e, const float *x,
                             const float *w, const float *pscale, const float *pshift, float *y,
                             double *stat_sum, double *stat_sq, hipStream_t stream) {
  if (prologue)
    return vec ? launch_mlp_fwd<NT, true, true, RAGGED>(call, b, cin, cout, e, x, w, pscale, pshift, y, stat_sum, stat_sq, stream)
               : launch_mlp_fwd<NT, true, false, RAGGED>(call, b, cin, cout, e, x, w, pscale, pshift, y, stat_sum, stat_sq, stream);
  return vec ? launch_mlp_fwd<NT, false, true, RAGGED>(call, b, cin, cout, e, x, w, pscale, pshift, y, stat_sum, stat_sq, stream)
             : launch_mlp_fwd<NT, false, false, RAGGED>(call, b, cin, cout, e, x, w, pscale, pshift, y, stat_sum, stat_sq, stream);
}

// padding K-steps (cin not a multiple of 32) or empty 32-channel tiles in the last channel block: the RAGGED instances
template <int NT>
static bool mlp_fwd_ragged(int cin, int cout) {
  return (cin % (2 * ML_KC) != 0 && ml_kpad(cin) - cin >= 2) || (cout % (32 * NT) != 0 && cout % (32 * NT) <= 32 * (NT - 1));
}

// the instances of one channel-tile width (32 * NT output channels per workgroup), ragged or not, behind a plain function
#define SIG3D_MLP_FWD_NAME(NT, R) sig3d_internal_mlp_fwd_nt##NT##_r##R
#define SIG3D_MLP_FWD_ARGS                                                                                       \
  const MlpFwdCall *call, int prologue, int vec, int b, int cin, int cout, long e, const float *x, const float *w, \
      const float *pscale, const float *pshift, float *y, double *stat_sum, double *stat_sq, void *stream
#define SIG3D_MLP_FWD_INSTANCES(NT, R)                                                                           \
  extern "C" int SIG3D_MLP_FWD_NAME(NT, R)(SIG3D_MLP_FWD_ARGS) {                                                 \
    return dispatch_mlp_fwd2<NT, (R) != 0>(*call, prologue != 0, vec != 0, b, cin, cout, e, x, w, pscale, pshift, y, \
                                           stat_sum, stat_sq, (hipStream_t)stream);                              \
  }
