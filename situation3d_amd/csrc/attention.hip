// attention.hip -- Q-Former attention core (forward + backward) on exact-f32 MFMA for gfx950.
//
// Replaces the score/softmax/context part of BertSelfAttention.forward
// (3DLLM_BLIP2-base/lavis/models/blip2_models/Qformer.py:185-227):
//   context = softmax(Q K^T / sqrt(d) + mask) V,   d = 64, few queries (32 / 52), many keys.
// The reference materialises the (B,12,Nq,Nk) score tensor in HBM three times (scores, probs,
// dropped probs); here scores never leave registers (streaming softmax over 32-key tiles).
//
// Numerics: v_mfma_f32_32x32x2_f32 is an exact f32 fma chain (no bf16/tf32 rounding), which is
// what keeps the 1e-4 activation bar of the north star.  Peak for this instruction is the f32
// matrix rate (157 TFLOP/s), not the bf16 one.
//
// Layout trick (CDNA MFMA C/D map: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)):
// the forward pass computes the TRANSPOSED tiles S^T = K Q^T and O^T = V^T P^T, so a query row
// is always a lane (col) and keys/features run over registers.  Row max / row sum / rescale
// are then in-lane VALU work plus one swap with lane^32 -- no LDS, no cross-lane reduction
// trees -- and the exponentiated S^T accumulator registers ARE the B operand of the PV MFMA
// (contraction over keys == contraction over the accumulator's row index).
// The reduction index of every MFMA is permuted (k-step s covers d = s and d = 32+s): sums are
// reassociated relative to a sequential dot product, well inside the f32 tolerance.
#include "sig3d_common.h"
#include <stdlib.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// head size D is a template parameter: 64 (Qformer.py:112: 768 / 12) and 96 (the MCAN blocks of the
// native SIG3D head, mcan_sqa_module.py:113-126: 768 / 8).  A lane owns half a row (D/2 features), the
// D features of the PV / dK / dV / dQ products are D/32 MFMA blocks of 32.
constexpr int AT_WAVES = 4;    // waves per workgroup, each streams every 4th key tile
constexpr int AT_NQ_MAX = 128; // backward keeps per-row statistics / dQ of one query chunk in LDS

// Phase timing for tools/attn_timing.py (compiled in only with -DSIG3D_ATTN_TIMING): workgroup
// (0,0,0), lane 0 of the wave that passes a mark stores the 100 MHz real-time counter.
#ifdef SIG3D_ATTN_TIMING
__device__ unsigned long long g_at_marks[2][4][16];  // [fwd/bwd][wave][mark]
#define AT_MARK(kind, id)                                                                   \
  do {                                                                                      \
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && (threadIdx.x & 63) == 0)   \
      g_at_marks[kind][threadIdx.x >> 6][id] = __builtin_amdgcn_s_memrealtime();            \
  } while (0)
#else
#define AT_MARK(kind, id) do { } while (0)
#endif

__device__ __forceinline__ int mfma_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// lane (row = l31, half) <- DH = D/2 contiguous floats base[row*stride + half*DH ...]; zero when !valid
template <int DH>
__device__ __forceinline__ void load_half_row(float (&f)[DH], const float *base, unsigned row,
                                              unsigned stride, int half, bool valid) {
  if (valid) {
    const float4 *p = reinterpret_cast<const float4 *>(base + row * stride + half * DH);
#pragma unroll
    for (int i = 0; i < DH / 4; ++i) {
      const float4 t = p[i];
      f[4 * i + 0] = t.x; f[4 * i + 1] = t.y; f[4 * i + 2] = t.z; f[4 * i + 3] = t.w;
    }
  } else {
#pragma unroll
    for (int i = 0; i < DH; ++i) f[i] = 0.f;
  }
}

// Token (batch bi, position i) -> storage row of a token-major operand.
//   plain layout   (seg == n): row = bi*n + i                      -- a dense (b, n, ...) tensor
//   two segments   (seg <  n): the first `seg` tokens of EVERY batch element are stored together
//                  (rows [0, b*seg)), the remaining n-seg tokens of every batch element from row
//                  `base2` on (base2 = b*seg when the segments are packed back to back).
// The Q-Former keeps [all query tokens | all text tokens] in that order (qformer.py): the query /
// text split of BertLayer.forward (Qformer.py:375-405) is then a pair of contiguous row ranges
// instead of strided slices that must be copied for the feed-forward GEMMs.
// 32-bit on purpose (the launchers check rows * stride < 2^31): the kernels form ~60 operand
// addresses per tile, and 64-bit row * stride products made address arithmetic cost as much as the
// memory round trip itself (2 us per tile phase, tools/attn_timing.py).
__device__ __forceinline__ unsigned tok_row(int i, int bi, int n, int seg, int base2) {
  return (unsigned)(i < seg ? bi * seg + i : base2 + bi * (n - seg) + (i - seg));
}

// Storage rows that hold no token -- the gap between the segments and the tail up to `rows` -- get
// zeros in columns [hi*D, hi*D+D) of an output tensor (row stride ld): padded layouts (qformer.py
// pads both segments to a common row count so that per-segment GEMMs batch) must never leak
// uninitialised memory into the row-wise kernels and weight-gradient GEMMs that follow.
template <int D>
__device__ __forceinline__ void zero_pad_rows(float *base, unsigned ld, int hi, int nb, int n, int seg,
                                              int base2, int rows) {
  if (rows <= 0) return;
  const bool two = seg > 0 && seg < n;
  const int end1 = two ? nb * seg : nb * n;
  const int start2 = two ? base2 : end1, end2 = two ? base2 + nb * (n - seg) : end1;
  const int npad = (start2 - end1) + max(rows - end2, 0);
  constexpr int PER = D / 4;  // float4 pieces per row
  for (int i = threadIdx.x; i < npad * PER; i += blockDim.x) {
    int r = i / PER;
    const int c4 = i - r * PER;
    r = r < start2 - end1 ? end1 + r : end2 + (r - (start2 - end1));
    *reinterpret_cast<float4 *>(base + (unsigned)r * ld + hi * D + c4 * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

// Attention-probability dropout (Qformer.py:219, nn.Dropout on the softmax output): the keep bit
// of element (b, head, query, key) is a pure hash of (device counter, call id, that index), so the
// forward and backward kernels regenerate identical masks in their different tile layouts and
// nothing (B,H,Nq,Nk)-shaped is ever stored.  Dropout acts on the NORMALISED probabilities:
// O = sum_k keep_k p_k V_k / ((1-p) l) with l = sum_k p_k over ALL keys, i.e. only the PV numerator
// is masked; in the backward pass dP is masked the same way and D = rowsum(dO*O) is unchanged.
struct AttnDropout {
  unsigned seed;
  unsigned thresh;   // keep iff the element's 16 hash bits >= thresh (p in units of 2^-16)
  float inv_keep;    // 1 / (1 - p)
  int on;
};

__device__ __forceinline__ unsigned at_mix(unsigned x) {
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  return x;
}

__device__ __forceinline__ AttnDropout make_dropout(float p_drop, unsigned call_id,
                                                    const unsigned *rng_counter) {
  AttnDropout d;
  d.on = p_drop > 0.f;
  d.seed = at_mix((rng_counter ? *rng_counter : 0u) * 0x9E3779B9u + call_id);
  d.thresh = (unsigned)((double)p_drop * 65536.0 + 0.5);
  d.inv_keep = 1.f / (1.f - p_drop);
  return d;
}

// One 32-bit hash serves the TWO keys 2i, 2i+1 of a query row (16 bits each: the drop probability is
// quantised to 2^-16, a relative error of the keep rate below 2e-5): the forward kernel, where a lane owns
// consecutive keys of one query, pays half the integer multiplies (quarter-rate on the VALU) per element.
__device__ __forceinline__ unsigned at_pair_hash(const AttnDropout &d, unsigned row_base, int key) {
  // row_base = ((b*h + head)*nq + query) * ceil(nk / 2)  (wraps mod 2^32 for huge shapes: still a fixed map)
  return at_mix(d.seed + (row_base + ((unsigned)key >> 1)) * 0x9E3779B9u);
}
__device__ __forceinline__ bool at_keep(const AttnDropout &d, unsigned row_base, int key) {
  const unsigned hsh = at_pair_hash(d, row_base, key);
  return ((key & 1) ? (hsh >> 16) : (hsh & 0xFFFFu)) >= d.thresh;
}

// ------------------------------------------------------------------------------------------
// forward: grid (q_tiles, h, b), 256 threads.
#ifndef SIG3D_ATTN_NT
#define SIG3D_ATTN_NT 1   // K / V of the long-key (rotating) forward are read exactly once, coalesced: non-temporal (+2 %)
#endif
template <int D, bool PIPE = false>
__global__ __launch_bounds__(AT_WAVES * 64, 2) void attention_fwd_kernel(
    int h, int nq, int nk, int q_seg, int k_seg, int q_base2, int k_base2, int q_rows, int ldq, int ldk,
    int ldv, float scale, float p_drop,
    unsigned call_id, const unsigned *__restrict__ rng_counter, const float *__restrict__ q,
    const float *__restrict__ k, const float *__restrict__ v, const float *__restrict__ mask,
    float *__restrict__ out, float *__restrict__ lse, int key_splits, float *__restrict__ part) {
  constexpr int DH = D / 2, NB = D / 32;
  __shared__ float s_o[AT_WAVES][D][32];
  __shared__ float s_m[AT_WAVES][32];
  __shared__ float s_l[AT_WAVES][32];
  __shared__ __attribute__((aligned(16))) float s_k[PIPE ? AT_WAVES : 1][PIPE ? 32 : 1][D + 4];   // PIPE: K staging tiles

  const int lane = lane_id(), l31 = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // blockIdx.y = query tile * key_splits + key split: with few queries and many keys (3D-LLM: 32 x
  // 5000..80000) the key range is cut into key_splits pieces so that the grid covers the chip; the
  // pieces leave (max, sum, un-normalised O) partials that attention_combine_kernel folds.
  // heads are the FASTEST grid dimension: the h workgroups that read the h 256-byte slices of the same K / V
  // rows are dispatched together, so a 3 KB row is fetched from HBM while its DRAM page is open
  const int q0 = (blockIdx.y / key_splits) * 32, split = blockIdx.y % key_splits;
  const int hi = blockIdx.x, bi = blockIdx.z;
  if (blockIdx.y == 0 && bi == 0) zero_pad_rows<D>(out, (unsigned)(h * D), hi, gridDim.z, nq, q_seg, q_base2, q_rows);
  // token-major operands: storage row r of head hi starts at base + r*ld + hi*D, where ld is the
  // row stride in floats (h*D for a dense (b, n, h*d) tensor, 3*h*D for a slice of a fused QKV
  // projection output) and r = tok_row(token, batch)
  const float *Q = q + hi * D;
  const float *K = k + hi * D;
  const float *V = v + hi * D;
  // no mask: the (unconditional) mask loads read any valid address -- the K base -- and are dropped
  const float *Mz = mask ? mask + (size_t)bi * nk : k;
  const bool has_mask = mask != nullptr;

  float qf[DH];
  load_half_row<DH>(qf, Q, tok_row(min(q0 + l31, nq - 1), bi, nq, q_seg, q_base2), ldq, half, q0 + l31 < nq);
  const AttnDropout drop = make_dropout(p_drop, call_id, rng_counter);
  const unsigned row_base = ((unsigned)((bi * h + hi) * nq + q0 + l31)) * (unsigned)((nk + 1) >> 1);

  float m_run = -INFINITY, l_run = 0.f;
  f32x16 o[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[j][r] = 0.f;
  const int ntiles = (nk + 31) / 32;
  const int tps = (ntiles + key_splits - 1) / key_splits;
  const int t_begin = split * tps, t_end = min(ntiles, t_begin + tps);
  // Every global operand of a key tile -- K rows, the 16 additive mask values and the 32 V^T operands of
  // the PV product -- is requested in one batch before the first MFMA, and the scheduler is fenced so that
  // it cannot sink the loads down to their uses (it did: 8 + 16 + 16 dependent round trips per tile).
  // The kernel fits TWO waves per SIMD (177 + 32 registers) and that is what hides the load latency and
  // the softmax VALU work behind the other wave's MFMAs: a one-tile register lookahead (second operand
  // set, 327 registers, one wave per SIMD) was measured SLOWER at every size (Nk = 80 000: 46.5 vs 52.2
  // TFLOP/s; 5000: 31.8 vs 37.5; 256: 13.1 vs 11.7 us); a K-only lookahead that keeps two waves per SIMD
  // changes nothing at full occupancy (53.8 vs 52.1 / 36.9 vs 38.4), and neither does giving every (batch,
  // head) contiguous 256-byte rows (tools/attn_bench.py --fold: 54.5) -- so DRAM locality is not the limit.
  auto load_tile = [&](float (&kf)[DH], float (&mk)[16], float (&va)[NB][16], int t) {
    const int key0 = t * 32;
    load_half_row<DH>(kf, K, tok_row(min(key0 + l31, nk - 1), bi, nk, k_seg, k_base2), ldk, half, true);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = min(key0 + mfma_row(r, half), nk - 1);
      mk[r] = Mz[key];
      const unsigned voff = tok_row(key, bi, nk, k_seg, k_base2) * (unsigned)ldv + l31;
#pragma unroll
      for (int j = 0; j < NB; ++j) va[j][r] = V[voff + 32 * j];
    }
  };
  auto compute_tile = [&](const float (&kf)[DH], const float (&mk)[16], const float (&va)[NB][16], int t) {
    const int key0 = t * 32;
    f32x16 st = {0};
#pragma unroll
    for (int s = 0; s < DH; ++s) st = mfma32(kf[s], qf[s], st);  // S^T[key][q]
    float p[16];
    float tmax = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = key0 + mfma_row(r, half);
      float sv = st[r] * scale;                     // Qformer.py:207 (/ sqrt(d))
      sv += has_mask ? mk[r] : 0.f;                 // Qformer.py:210
      sv = key < nk ? sv : -INFINITY;
      p[r] = sv;
      tmax = fmaxf(tmax, sv);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
    const float m_new = fmaxf(m_run, tmax);
    const float alpha = __expf(m_run - m_new);
    float rs = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      p[r] = __expf(p[r] - m_new);
      rs += p[r];  // the normaliser sees every key
    }
    if (drop.on) {  // ... the PV numerator only the kept ones; registers 2i, 2i+1 are keys 2k, 2k+1: one hash
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const unsigned hsh = at_pair_hash(drop, row_base, key0 + mfma_row(r, half));
        p[r] = (hsh & 0xFFFFu) >= drop.thresh ? p[r] * drop.inv_keep : 0.f;
        p[r + 1] = (hsh >> 16) >= drop.thresh ? p[r + 1] * drop.inv_keep : 0.f;
      }
    }
    rs += __shfl_xor(rs, 32);
    l_run = l_run * alpha + rs;
    m_run = m_new;
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[j][r] *= alpha;   // (skipping this under a uniform "no row raised its maximum" branch was slower: it splits the block)
#pragma unroll
    for (int s = 0; s < 16; ++s) {  // O^T[d][q] += V^T[d][key] P^T[key][q]
#pragma unroll
      for (int j = 0; j < NB; ++j) o[j] = mfma32(va[j][s], p[s], o[j]);
    }
  };
  if constexpr (!PIPE) {
    for (int t = t_begin + wave; t < t_end; t += AT_WAVES) {
      float kf[DH], mk[16], va[NB][16];
      load_tile(kf, mk, va, t);
      __builtin_amdgcn_sched_barrier(0);
      compute_tile(kf, mk, va, t);
    }
  } else if constexpr (D == 64) {
   if (t_begin + wave < t_end) {
    // Long key ranges (3D-LLM shapes): ROTATING prefetch without a second operand set.  The K rows and mask values
    // of a tile are dead once S^T and the masked scores exist, the V^T operands once the PV product has consumed
    // them -- so the next tile's K (+ mask) is requested right after the scores, under the softmax and the PV
    // MFMAs, and the next tile's V right after PV, under the next tile's S MFMAs: every MFMA phase of a wave has
    // 8 KB of its own loads in flight, in the same 177 + 32 registers (two waves per SIMD).
    float kf[DH], mk[16], va[NB][16];
    const int last = t_begin + wave + ((t_end - 1 - t_begin - wave) / AT_WAVES) * AT_WAVES;   // this wave's last tile
    auto load_k = [&](int t) {
      const int key0 = t * 32;
      // COALESCED: 16 lanes read one key row of this head (256 contiguous bytes), four rows per instruction, so
      // every cache line is requested once.  (A lane that owns a whole row -- the MFMA operand layout, as in
      // load_half_row -- touches its two lines in eight separate instructions: 8x the L1 requests, and the
      // non-temporal form of THAT pattern runs at 0.66x because it defeats the L1 hits it lives on.)  The rows
      // reach the operand layout through a wave-private LDS tile right before S^T (k_to_operands).
      static_assert(D == 64, "coalesced K staging is written for 64-wide heads");
      constexpr bool NTLOAD = SIG3D_ATTN_NT != 0;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const unsigned row = tok_row(min(key0 + 4 * i + (lane >> 4), nk - 1), bi, nk, k_seg, k_base2);
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4 tv = NTLOAD ? __builtin_nontemporal_load(reinterpret_cast<const f4 *>(K + row * (unsigned)ldk + 4 * (lane & 15)))
                             : *reinterpret_cast<const f4 *>(K + row * (unsigned)ldk + 4 * (lane & 15));
        kf[4 * i + 0] = tv.x; kf[4 * i + 1] = tv.y; kf[4 * i + 2] = tv.z; kf[4 * i + 3] = tv.w;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) mk[r] = Mz[min(key0 + mfma_row(r, half), nk - 1)];
    };
    auto load_v = [&](int t) {
      constexpr bool NTLOAD = SIG3D_ATTN_NT != 0;
      const int key0 = t * 32;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const unsigned voff = tok_row(min(key0 + mfma_row(r, half), nk - 1), bi, nk, k_seg, k_base2) * (unsigned)ldv + l31;
#pragma unroll
        for (int j = 0; j < NB; ++j) va[j][r] = NTLOAD ? __builtin_nontemporal_load(V + voff + 32 * j) : V[voff + 32 * j];
      }
    };
    float(*sk)[D + 4] = s_k[wave];
    auto k_to_operands = [&]() {   // staged rows -> lane (key l31, half) holds its 32 floats
#pragma unroll
      for (int i = 0; i < 8; ++i)
        *reinterpret_cast<float4 *>(&sk[4 * i + (lane >> 4)][4 * (lane & 15)]) =
            make_float4(kf[4 * i + 0], kf[4 * i + 1], kf[4 * i + 2], kf[4 * i + 3]);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float4 tv = *reinterpret_cast<const float4 *>(&sk[l31][half * DH + 4 * i]);
        kf[4 * i + 0] = tv.x; kf[4 * i + 1] = tv.y; kf[4 * i + 2] = tv.z; kf[4 * i + 3] = tv.w;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();   // the tile is free for the next staging only after every lane has read
    };
    load_k(t_begin + wave);
    load_v(t_begin + wave);
    for (int t = t_begin + wave; t < t_end; t += AT_WAVES) {
      const int key0 = t * 32;
      const int tn = min(t + AT_WAVES, last);   // past the end: the last tile again (an L2 hit that nobody reads)
      __builtin_amdgcn_sched_barrier(0);
      k_to_operands();
      f32x16 st = {0};
#pragma unroll
      for (int s2 = 0; s2 < DH; ++s2) st = mfma32(kf[s2], qf[s2], st);  // S^T[key][q]
      float p[16];
      float tmax = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = key0 + mfma_row(r, half);
        float sv = st[r] * scale;
        sv += has_mask ? mk[r] : 0.f;
        sv = key < nk ? sv : -INFINITY;
        p[r] = sv;
        tmax = fmaxf(tmax, sv);
      }
      __builtin_amdgcn_sched_barrier(0);
      load_k(tn);
      __builtin_amdgcn_sched_barrier(0);
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
      const float m_new = fmaxf(m_run, tmax);
      const float alpha = __expf(m_run - m_new);
      float rs = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        p[r] = __expf(p[r] - m_new);
        rs += p[r];
      }
      if (drop.on) {
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const unsigned hsh = at_pair_hash(drop, row_base, key0 + mfma_row(r, half));
          p[r] = (hsh & 0xFFFFu) >= drop.thresh ? p[r] * drop.inv_keep : 0.f;
          p[r + 1] = (hsh >> 16) >= drop.thresh ? p[r + 1] * drop.inv_keep : 0.f;
        }
      }
      rs += __shfl_xor(rs, 32);
      l_run = l_run * alpha + rs;
      m_run = m_new;
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[j][r] *= alpha;
#pragma unroll
      for (int s2 = 0; s2 < 16; ++s2) {  // O^T[d][q] += V^T[d][key] P^T[key][q]
#pragma unroll
        for (int j = 0; j < NB; ++j) o[j] = mfma32(va[j][s2], p[s2], o[j]);
      }
      __builtin_amdgcn_sched_barrier(0);
      load_v(tn);
    }
   }
  }

  // combine the AT_WAVES partial (m, l, O^T) triples
#pragma unroll
  for (int j = 0; j < NB; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) s_o[wave][32 * j + mfma_row(r, half)][l31] = o[j][r];
  if (half == 0) {
    s_m[wave][l31] = m_run;
    s_l[wave][l31] = l_run;
  }
  __syncthreads();
  {
    constexpr int FG = D / 8;                                // features per thread
    const int qq = threadIdx.x & 31, dg = threadIdx.x >> 5;  // 8 d-groups of FG features
    float mt = -INFINITY;
#pragma unroll
    for (int w = 0; w < AT_WAVES; ++w) mt = fmaxf(mt, s_m[w][qq]);
    float wgt[AT_WAVES], lt = 0.f;
#pragma unroll
    for (int w = 0; w < AT_WAVES; ++w) {
      // waves without a tile: exp(-inf) = 0; a workgroup without ANY tile (mt = -inf) must give 0, not exp(nan)
      wgt[w] = mt == -INFINITY ? 0.f : __expf(s_m[w][qq] - mt);
      lt += s_l[w][qq] * wgt[w];
    }
    if (key_splits > 1) {
      // partial of this key split: part[row][split] = {m, l, O[D] relative to m}, row = (b*h + head)*nq_pad + q
      if (q0 + qq < nq) {
        const int nq_pad = (nq + 31) / 32 * 32;
        float *pp = part + (((size_t)(bi * h + hi) * nq_pad + q0 + qq) * key_splits + split) * (D + 2);
        if (dg == 0) { pp[0] = mt; pp[1] = lt; }
#pragma unroll
        for (int i = 0; i < FG; ++i) {
          float acc = 0.f;
#pragma unroll
          for (int w = 0; w < AT_WAVES; ++w) acc += s_o[w][dg * FG + i][qq] * wgt[w];
          pp[2 + dg * FG + i] = acc;
        }
      }
      return;
    }
    const float inv = 1.f / lt;
    if (q0 + qq < nq) {
      float res[FG];
#pragma unroll
      for (int i = 0; i < FG; ++i) {
        float acc = 0.f;
#pragma unroll
        for (int w = 0; w < AT_WAVES; ++w) acc += s_o[w][dg * FG + i][qq] * wgt[w];
        res[i] = acc * inv;
      }
      // context_layer.permute(0,2,1,3).view(B, Nq, 768)  (Qformer.py:225-227)
      float *op = out + tok_row(q0 + qq, bi, nq, q_seg, q_base2) * (unsigned)(h * D) + hi * D + dg * FG;
#pragma unroll
      for (int i = 0; i < FG; i += 4)
        *reinterpret_cast<float4 *>(op + i) = make_float4(res[i], res[i + 1], res[i + 2], res[i + 3]);
      if (lse && dg == 0) lse[(size_t)(bi * h + hi) * nq + q0 + qq] = mt + __logf(lt);
    }
  }
}

// fold the key-split partials of one (batch, head, query) row: one wave per row, lane = feature
template <int D>
__global__ __launch_bounds__(256) void attention_combine_kernel(int h, int nq, int q_seg, int q_base2,
                                                                int key_splits, int nb,
                                                                const float *__restrict__ part,
                                                                float *__restrict__ out,
                                                                float *__restrict__ lse) {
  const int lane = lane_id();
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);   // (b*h + head)*nq + q
  if (row >= (long)nb * h * nq) return;
  const int qq = (int)(row % nq), bh = (int)(row / nq), hi = bh % h, bi = bh / h;
  const int nq_pad = (nq + 31) / 32 * 32;
  const float *pp = part + ((size_t)bh * nq_pad + qq) * key_splits * (D + 2);
  float mt = -INFINITY;
  for (int s2 = 0; s2 < key_splits; ++s2) mt = fmaxf(mt, pp[(size_t)s2 * (D + 2)]);
  constexpr int FL = (D + 63) / 64;  // features per lane: lane, lane + 64
  float lt = 0.f, acc[FL];
#pragma unroll
  for (int f = 0; f < FL; ++f) acc[f] = 0.f;
  for (int s2 = 0; s2 < key_splits; ++s2) {
    const float *ps = pp + (size_t)s2 * (D + 2);
    const float wgt = __expf(ps[0] - mt);   // splits without keys: exp(-inf) = 0
    lt += ps[1] * wgt;
#pragma unroll
    for (int f = 0; f < FL; ++f) acc[f] += ps[2 + min(lane + 64 * f, D - 1)] * wgt;
  }
#pragma unroll
  for (int f = 0; f < FL; ++f)
    if (lane + 64 * f < D)
      out[tok_row(qq, bi, nq, q_seg, q_base2) * (unsigned)(h * D) + hi * D + lane + 64 * f] = acc[f] / lt;
  if (lse && lane == 0) lse[row] = mt + __logf(lt);
}

// ------------------------------------------------------------------------------------------
// backward: grid (key_splits * query_chunks, h, b), 256 threads.  A workgroup owns one chunk of at most
// `qchunk` query rows (their statistics and dQ images live in LDS) and one range of key tiles; with more
// than one chunk dK / dV are accumulated with atomics.  Work items are (32-key tile, 32-query tile)
// pairs.  With >= 4 key tiles per workgroup a wave owns whole key tiles (dK / dV are plain stores
// from its accumulators); with fewer (self-attention: 52 keys = 2 tiles) the query tiles of a key
// tile are spread over the otherwise idle waves and their dK / dV partials meet in LDS.
// dQ: every wave accumulates into its OWN LDS image (plain read-modify-write around the dQ MFMAs,
// no zero fill: first visit starts from 0) and the images are summed once at the end --
// ds_add_f32 from four waves onto one image cost 9 us of a 27 us launch (tools/attn_timing.py).
template <int D, bool ONEQT>
__global__ __launch_bounds__(AT_WAVES * 64, 1) void attention_bwd_kernel(
    int h, int nq, int nk, int q_seg, int k_seg, int q_base2, int k_base2, int q_rows, int k_rows,
    int ldq, int ldk, int ldv, float scale, int tiles_per_split, int nsplits, int qchunk, int atomic_dq,
    int atomic_dkv, int qsplit_max, float p_drop,
    unsigned call_id, const unsigned *__restrict__ rng_counter,
    const float *__restrict__ q, const float *__restrict__ k, const float *__restrict__ v,
    const float *__restrict__ mask, const float *__restrict__ out, const float *__restrict__ lse,
    const float *__restrict__ grad_out, float *__restrict__ dq, float *__restrict__ dk,
    float *__restrict__ dv) {
  __shared__ float s_D[AT_NQ_MAX];
  __shared__ float s_lse[AT_NQ_MAX];
  __shared__ float s_T[AT_WAVES][32][33];
  constexpr int DH = D / 2, NB = D / 32;
  // dynamic: dQ images [AT_WAVES][qchunk][D + 1] (rows local to the chunk), then (qsplit_max == 2 only)
  // the dK/dV partials of the q-split mode [2 key slots][2*NB*16 regs][64 lanes]
  extern __shared__ __attribute__((aligned(16))) float s_dq[];
  auto dq_img = [&](int w, int ql, int d) -> float & { return s_dq[((size_t)w * qchunk + ql) * (D + 1) + d]; };
  float *s_red = s_dq + (size_t)AT_WAVES * qchunk * (D + 1);
  auto red = [&](int sl, int r, int ln) -> float & { return s_red[(sl * (2 * NB * 16) + r) * 64 + ln]; };

  const int lane = lane_id(), l31 = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int hi = blockIdx.x, bi = blockIdx.z;   // heads fastest, see the forward kernel
  const int ksplit = blockIdx.y % nsplits, chunk = blockIdx.y / nsplits;
  const int qc0 = chunk * qchunk, q_end = min(nq, qc0 + qchunk), nql = q_end - qc0;  // this chunk's queries
  if (blockIdx.y == 0 && bi == 0) {
    zero_pad_rows<D>(dq, (unsigned)ldq, hi, gridDim.z, nq, q_seg, q_base2, q_rows);
    zero_pad_rows<D>(dk, (unsigned)ldk, hi, gridDim.z, nk, k_seg, k_base2, k_rows);
    zero_pad_rows<D>(dv, (unsigned)ldv, hi, gridDim.z, nk, k_seg, k_base2, k_rows);
  }
  const size_t bh = (size_t)(bi * h + hi);
  const unsigned ostride = (unsigned)h * D;  // out / grad_out are dense rows of h*d floats
  const float *Q = q + hi * D;          // storage row r at Q + r*ldq, r = tok_row(token, batch)
  const float *K = k + hi * D;
  const float *V = v + hi * D;
  const float *M = mask ? mask + (size_t)bi * nk : nullptr;
  const float *O = out + hi * D;
  const float *dO = grad_out + hi * D;
  auto qrow = [&](int i) { return tok_row(i, bi, nq, q_seg, q_base2); };
  auto krow_of = [&](int j) { return tok_row(j, bi, nk, k_seg, k_base2); };

  AT_MARK(1, 0);
  // D[q] = sum_d dO[q][d] * O[q][d]; two threads per (chunk-local) row
  {
    const int ql = threadIdx.x >> 1, hh = threadIdx.x & 1;
    float part = 0.f;
    if (ql < nql) {
      const float4 *a = reinterpret_cast<const float4 *>(dO + qrow(qc0 + ql) * ostride + hh * DH);
      const float4 *c = reinterpret_cast<const float4 *>(O + qrow(qc0 + ql) * ostride + hh * DH);
#pragma unroll
      for (int i = 0; i < DH / 4; ++i) {
        const float4 x = a[i], y = c[i];
        part += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
      }
    }
    part += __shfl_xor(part, 1);
    if (hh == 0 && ql < AT_NQ_MAX) {
      s_D[ql] = ql < nql ? part : 0.f;
      s_lse[ql] = ql < nql ? lse[bh * nq + qc0 + ql] : 0.f;
    }
  }
  __syncthreads();
  AT_MARK(1, 1);

  const AttnDropout drop = make_dropout(p_drop, call_id, rng_counter);
  const int ntiles = (nk + 31) / 32;
  const int t_begin = ksplit * tiles_per_split;
  const int t_end = min(ntiles, t_begin + tiles_per_split);
  const int nqt = (nql + 31) / 32;
  // work assignment (uniform per wave): key slot = wave % nslots, query part = wave / nslots
  const int nt = t_end - t_begin;
  const int nslots = nt >= AT_WAVES ? AT_WAVES : max(nt, 1);
  const int qsplit = nt >= AT_WAVES ? 1 : min(nqt, min(AT_WAVES / nslots, qsplit_max));
  const int slot = wave % nslots, qpart = wave / nslots;
  const bool active = qpart < qsplit;
  unsigned visited = 0;  // bit qt: this wave's dQ image holds q-tile qt
  f32x16 dvt[NB], dkt[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) dvt[j][r] = dkt[j][r] = 0.f;

  if constexpr (ONEQT) {
    // ---- one query tile (nq <= 32: the 32 learned queries against any number of scene tokens) ----------
    // Everything that depends on the queries only -- Q and dO rows for S and dP, their transposed operands
    // for dK and dV -- is loaded ONCE and stays in registers (the generic loop re-fetches 96 values per
    // lane per key tile), dQ accumulates in registers across the key tiles (no LDS read-modify-write per
    // tile), and the K / V rows of the next tile are in flight while the 160 MFMAs of the current one
    // issue.  One wave per SIMD, ~400 of its 512 registers.  At Nk = 80 000 the generic loop ran at
    // 40 TFLOP/s with the matrix pipe waiting on two dependent load round trips per tile.
    const int q0 = qc0;
    // dO / Q rows of the tile as the transposed operands of dV / dK: one LDS copy for the four waves (row
    // stride 72 floats: the two half-waves read rows r and r + 4, 4 * 72 = 32 banks apart); in registers
    // they cost 64 VGPRs and the kernel spilled
    constexpr int GX_LD = 72;
    __shared__ float s_g[32][GX_LD], s_x[32][GX_LD];
    {
      const int row = threadIdx.x >> 3, c8 = (threadIdx.x & 7) * 8;
      const unsigned rr = qrow(min(q0 + row, nq - 1));
      const bool ok = q0 + row < q_end;
      static_assert(D == 64, "the one-query-tile path is built for head size 64");
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float4 a = *reinterpret_cast<const float4 *>(dO + rr * ostride + c8 + 4 * i);
        const float4 c = *reinterpret_cast<const float4 *>(Q + rr * (unsigned)ldq + c8 + 4 * i);
        *reinterpret_cast<float4 *>(&s_g[row][c8 + 4 * i]) = ok ? a : make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4 *>(&s_x[row][c8 + 4 * i]) = ok ? c : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    float fr[DH], fdo[DH];
    {
      const unsigned my_qrow = qrow(min(q0 + l31, nq - 1));
      load_half_row<DH>(fr, Q, my_qrow, ldq, half, q0 + l31 < q_end);
      load_half_row<DH>(fdo, dO, my_qrow, ostride, half, q0 + l31 < q_end);
    }
    __syncthreads();
    f32x16 dqt[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) dqt[j][r] = 0.f;
    const float *Mz = M ? M : K;  // no mask: the unconditional load reads any valid address and is dropped
    auto load_kv = [&](float (&kf)[DH], float (&vf)[DH], float &mk, int t) {
      const unsigned kr = krow_of(min(t * 32 + l31, nk - 1));
      load_half_row<DH>(kf, K, kr, ldk, half, true);
      load_half_row<DH>(vf, V, kr, ldv, half, true);
      mk = Mz[min(t * 32 + l31, nk - 1)];
    };
    auto load_kop = [&](float (&kop)[NB][16], int t) {
#pragma unroll
      for (int s = 0; s < 16; ++s) {  // K^T operands of the dQ product
        const unsigned koff = krow_of(min(t * 32 + mfma_row(s, half), nk - 1)) * (unsigned)ldk + l31;
#pragma unroll
        for (int j = 0; j < NB; ++j) kop[j][s] = K[koff + 32 * j];
      }
    };
    auto tile = [&](const float (&kf)[DH], const float (&vf)[DH], float mk_in, const float (&kop)[NB][16], int t) {
      const int key0 = t * 32;
      const bool key_ok = key0 + l31 < nk;
      const float mk = M ? mk_in : 0.f;
      f32x16 sacc = {0};
#pragma unroll
      for (int s = 0; s < DH; ++s) sacc = mfma32(fr[s], kf[s], sacc);  // S[q][key]
      float p[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qq = q0 + mfma_row(r, half);
        const float row_lse = s_lse[min(qq - qc0, AT_NQ_MAX - 1)];   // fully fill-masked rows: see the generic loop
        const float e = row_lse < -1e8f ? 1.f / (float)nk : __expf(sacc[r] * scale + mk - row_lse);
        p[r] = (key_ok && qq < q_end) ? e : 0.f;
      }
      f32x16 dpacc = {0};
#pragma unroll
      for (int s = 0; s < DH; ++s) dpacc = mfma32(fdo[s], vf[s], dpacc);  // dP[q][key]
      float ds[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qq = q0 + mfma_row(r, half);
        float dp = dpacc[r];
        if (drop.on) {
          const unsigned row_base = ((unsigned)((bi * h + hi) * nq + qq)) * (unsigned)((nk + 1) >> 1);
          const bool keep = at_keep(drop, row_base, key0 + l31);
          dp = keep ? dp * drop.inv_keep : 0.f;
          ds[r] = p[r] * (dp - s_D[min(qq - qc0, AT_NQ_MAX - 1)]);
          p[r] = keep ? p[r] * drop.inv_keep : 0.f;
        } else {
          ds[r] = p[r] * (dp - s_D[min(qq - qc0, AT_NQ_MAX - 1)]);
        }
        if (s_lse[min(qq - qc0, AT_NQ_MAX - 1)] < -1e8f) ds[r] = 0.f;
      }
      f32x16 dvt1[NB], dkt1[NB];
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) dvt1[j][r] = dkt1[j][r] = 0.f;
#pragma unroll
      for (int s = 0; s < 16; ++s) {
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          dvt1[j] = mfma32(s_g[mfma_row(s, half)][l31 + 32 * j], p[s], dvt1[j]);
          dkt1[j] = mfma32(s_x[mfma_row(s, half)][l31 + 32 * j], ds[s], dkt1[j]);
        }
      }
      // dQ^T[d][q] += K^T[d][key] dS^T[key][q]: the dS tile is transposed through wave-private LDS
#pragma unroll
      for (int r = 0; r < 16; ++r) s_T[wave][l31][mfma_row(r, half)] = ds[r];
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const float bq = s_T[wave][mfma_row(s, half)][l31];
#pragma unroll
        for (int j = 0; j < NB; ++j) dqt[j] = mfma32(kop[j][s], bq, dqt[j]);
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
      __builtin_amdgcn_wave_barrier();
      if (key_ok) {
        float *dvp = dv + krow_of(key0 + l31) * (unsigned)ldv + hi * D;  // grads mirror the inputs
        float *dkp = dk + krow_of(key0 + l31) * (unsigned)ldk + hi * D;
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const int d = 32 * j + 8 * g4 + 4 * half;
            const float a4[4] = {dvt1[j][4 * g4], dvt1[j][4 * g4 + 1], dvt1[j][4 * g4 + 2], dvt1[j][4 * g4 + 3]};
            const float c4[4] = {dkt1[j][4 * g4] * scale, dkt1[j][4 * g4 + 1] * scale, dkt1[j][4 * g4 + 2] * scale,
                                 dkt1[j][4 * g4 + 3] * scale};
            if (atomic_dkv) {
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                unsafeAtomicAdd(dvp + d + i, a4[i]);
                unsafeAtomicAdd(dkp + d + i, c4[i]);
              }
            } else {
              *reinterpret_cast<float4 *>(dvp + d) = make_float4(a4[0], a4[1], a4[2], a4[3]);
              *reinterpret_cast<float4 *>(dkp + d) = make_float4(c4[0], c4[1], c4[2], c4[3]);
            }
          }
      }
    };
    {
      float kfA[DH], vfA[DH], mkA, kfB[DH], vfB[DH], mkB, kop[NB][16];
      // one query tile: qsplit == 1, so a workgroup with fewer than AT_WAVES key tiles leaves waves idle
      int t = active ? t_begin + slot : t_end;
      if (t < t_end) load_kv(kfA, vfA, mkA, t);
      for (; t < t_end; t += 2 * nslots) {
        const int t1 = t + nslots, t2 = t + 2 * nslots;
        load_kop(kop, t);
        load_kv(kfB, vfB, mkB, min(t1, t_end - 1));   // unconditional lookahead (clamped at the tail)
        __builtin_amdgcn_sched_barrier(0);
        tile(kfA, vfA, mkA, kop, t);
        if (t1 < t_end) {
          load_kop(kop, t1);
          load_kv(kfA, vfA, mkA, min(t2, t_end - 1));
          __builtin_amdgcn_sched_barrier(0);
          tile(kfB, vfB, mkB, kop, t1);
        }
      }
    }
    // the wave's dQ goes to its LDS image once; the common tail below sums the four images
    if (active && t_begin + slot < t_end) {
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq_img(wave, l31, 32 * j + mfma_row(r, half)) = dqt[j][r];
    }
  } else {
  for (int t = t_begin + slot; active && t < t_end; t += nslots) {
    const int key0 = t * 32;
    const int krow = min(key0 + l31, nk - 1);
    const bool key_ok = key0 + l31 < nk;
    float kf[DH], vf[DH], kop[NB][16];
    load_half_row<DH>(kf, K, krow_of(krow), ldk, half, true);
    load_half_row<DH>(vf, V, krow_of(krow), ldv, half, true);
    const float mk = M ? M[krow] : 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) {  // K^T operands of the dQ product: fixed for the whole key tile
      const unsigned koff = krow_of(min(key0 + mfma_row(s, half), nk - 1)) * (unsigned)ldk + l31;
#pragma unroll
      for (int j = 0; j < NB; ++j) kop[j][s] = K[koff + 32 * j];
    }
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) dvt[j][r] = dkt[j][r] = 0.f;
    AT_MARK(1, 2);

    for (int qt = qpart; qt < nqt; qt += qsplit) {
      const int q0 = qc0 + qt * 32;
      // all global operands of this (key tile, query tile) pair are requested before the first MFMA
      // and the scheduler is fenced (see the forward kernel): Q and dO rows for S and dP, and the
      // transposed dO / Q operands of the dV / dK products
      float fr[DH], fdo[DH], g[NB][16], x[NB][16];
      const unsigned my_qrow = qrow(min(q0 + l31, nq - 1));
      load_half_row<DH>(fr, Q, my_qrow, ldq, half, q0 + l31 < q_end);
      load_half_row<DH>(fdo, dO, my_qrow, ostride, half, q0 + l31 < q_end);
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const unsigned qq = qrow(min(q0 + mfma_row(s, half), nq - 1));
        const unsigned goff = qq * ostride + l31, xoff = qq * (unsigned)ldq + l31;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          g[j][s] = dO[goff + 32 * j];
          x[j][s] = Q[xoff + 32 * j];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      AT_MARK(1, 3);
      f32x16 sacc = {0};
#pragma unroll
      for (int s = 0; s < DH; ++s) sacc = mfma32(fr[s], kf[s], sacc);  // S[q][key]
      AT_MARK(1, 4);
      float p[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qq = q0 + mfma_row(r, half);
        // a row whose every key carries a fill mask (-1e9, the MCAN blocks' masked_fill) has all scores equal
        // to the fill and log-sum-exp = fill + log(nk), which float32 cannot hold next to 1e9: such a row
        // attends uniformly, p = 1/nk -- and, the scores having been REPLACED by a constant, passes no
        // gradient to q / k (dS = 0 below); dV still sees p
        const float row_lse = s_lse[min(qq - qc0, AT_NQ_MAX - 1)];
        const float e = row_lse < -1e8f ? 1.f / (float)nk : __expf(sacc[r] * scale + mk - row_lse);
        p[r] = (key_ok && qq < q_end) ? e : 0.f;
      }
      f32x16 dpacc = {0};
#pragma unroll
      for (int s = 0; s < DH; ++s) dpacc = mfma32(fdo[s], vf[s], dpacc);  // dP[q][key]
      float ds[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qq = q0 + mfma_row(r, half);
        float dp = dpacc[r];
        if (drop.on) {  // same keep bit as the forward pass: element (b, head, qq, key0 + l31)
          const unsigned row_base = ((unsigned)((bi * h + hi) * nq + qq)) * (unsigned)((nk + 1) >> 1);
          const bool keep = at_keep(drop, row_base, key0 + l31);
          dp = keep ? dp * drop.inv_keep : 0.f;          // d(P_dropped)/dP
          ds[r] = p[r] * (dp - s_D[min(qq - qc0, AT_NQ_MAX - 1)]);
          p[r] = keep ? p[r] * drop.inv_keep : 0.f;      // dV uses the dropped probabilities
        } else {
          ds[r] = p[r] * (dp - s_D[min(qq - qc0, AT_NQ_MAX - 1)]);
        }
        if (s_lse[min(qq - qc0, AT_NQ_MAX - 1)] < -1e8f) ds[r] = 0.f;  // fully fill-masked row, see above
      }
      AT_MARK(1, 5);
      // dV^T[d][key] += dO^T[d][q] P[q][key];  dK^T[d][key] += Q^T[d][q] dS[q][key]
#pragma unroll
      for (int s = 0; s < 16; ++s) {
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          dvt[j] = mfma32(g[j][s], p[s], dvt[j]);
          dkt[j] = mfma32(x[j][s], ds[s], dkt[j]);
        }
      }
      AT_MARK(1, 6);
      // dQ^T[d][q] = K^T[d][key] dS^T[key][q]: transpose the dS tile through wave-private LDS
#pragma unroll
      for (int r = 0; r < 16; ++r) s_T[wave][l31][mfma_row(r, half)] = ds[r];
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
      __builtin_amdgcn_wave_barrier();
      const int ql0 = qt * 32;  // chunk-local row of the tile
      f32x16 dqt[NB];
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) dqt[j][r] = 0.f;
      if ((visited >> qt) & 1u) {  // continue this wave's running dQ of the q-tile
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) dqt[j][r] = dq_img(wave, ql0 + l31, 32 * j + mfma_row(r, half));
      }
      visited |= 1u << qt;
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const float bq = s_T[wave][mfma_row(s, half)][l31];
#pragma unroll
        for (int j = 0; j < NB; ++j) dqt[j] = mfma32(kop[j][s], bq, dqt[j]);
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
      __builtin_amdgcn_wave_barrier();
      AT_MARK(1, 7);
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq_img(wave, ql0 + l31, 32 * j + mfma_row(r, half)) = dqt[j][r];
      AT_MARK(1, 8);
    }
    if (key_ok && qsplit == 1) {
      float *dvp = dv + krow_of(key0 + l31) * (unsigned)ldv + hi * D;  // grads mirror the inputs
      float *dkp = dk + krow_of(key0 + l31) * (unsigned)ldk + hi * D;
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {  // regs 4g..4g+3 are four consecutive feature rows
          const int d = 32 * j + 8 * g4 + 4 * half;
          const float a4[4] = {dvt[j][4 * g4], dvt[j][4 * g4 + 1], dvt[j][4 * g4 + 2], dvt[j][4 * g4 + 3]};
          const float c4[4] = {dkt[j][4 * g4] * scale, dkt[j][4 * g4 + 1] * scale, dkt[j][4 * g4 + 2] * scale,
                               dkt[j][4 * g4 + 3] * scale};
          if (atomic_dkv) {  // several query chunks meet in dK / dV
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              unsafeAtomicAdd(dvp + d + i, a4[i]);
              unsafeAtomicAdd(dkp + d + i, c4[i]);
            }
          } else {
            *reinterpret_cast<float4 *>(dvp + d) = make_float4(a4[0], a4[1], a4[2], a4[3]);
            *reinterpret_cast<float4 *>(dkp + d) = make_float4(c4[0], c4[1], c4[2], c4[3]);
          }
        }
    }
  }
  }
  AT_MARK(1, 9);
  if (qsplit > 1) {
    // q-split mode (at most one key tile per wave): the part-1 wave of a key slot parks its dK / dV
    // accumulators in LDS, the part-0 wave adds them and stores
    if (active && qpart == 1) {
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          red(slot, 16 * j + r, lane) = dvt[j][r];
          red(slot, 16 * (NB + j) + r, lane) = dkt[j][r];
        }
    }
    __syncthreads();
    const int key0 = (t_begin + slot) * 32;
    if (active && qpart == 0 && t_begin + slot < t_end && key0 + l31 < nk) {
      float *dvp = dv + krow_of(key0 + l31) * (unsigned)ldv + hi * D;
      float *dkp = dk + krow_of(key0 + l31) * (unsigned)ldk + hi * D;
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int d = 32 * j + 8 * g4 + 4 * half;
          float a4[4], c4[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int r = 4 * g4 + i;
            a4[i] = dvt[j][r] + red(slot, 16 * j + r, lane);
            c4[i] = (dkt[j][r] + red(slot, 16 * (NB + j) + r, lane)) * scale;
          }
          if (atomic_dkv) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              unsafeAtomicAdd(dvp + d + i, a4[i]);
              unsafeAtomicAdd(dkp + d + i, c4[i]);
            }
          } else {
            *reinterpret_cast<float4 *>(dvp + d) = make_float4(a4[0], a4[1], a4[2], a4[3]);
            *reinterpret_cast<float4 *>(dkp + d) = make_float4(c4[0], c4[1], c4[2], c4[3]);
          }
        }
    }
  }
  __syncthreads();
  AT_MARK(1, 10);
  // sum the waves' images: 4 consecutive features per thread, all LDS reads issued before the adds
  for (int it = threadIdx.x; it < nql * (D / 4); it += AT_WAVES * 64) {
    const int qq = it / (D / 4), d = (it - qq * (D / 4)) * 4;  // qq: chunk-local row
    const int qt = qq >> 5;
    float part[AT_WAVES][4];
#pragma unroll
    for (int w = 0; w < AT_WAVES; ++w) {  // waves whose image holds this q-tile (same rule as above)
      const int wslot = w % nslots, wpart = w / nslots;
      const bool has = wpart < qsplit && t_begin + wslot < t_end && (qt % qsplit) == wpart;
#pragma unroll
      for (int i = 0; i < 4; ++i) part[w][i] = has ? dq_img(w, qq, d + i) : 0.f;
    }
    float val[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) val[i] = ((part[0][i] + part[1][i]) + (part[2][i] + part[3][i])) * scale;
    float *dst = dq + qrow(qc0 + qq) * (unsigned)ldq + hi * D + d;
    if (atomic_dq) {
#pragma unroll
      for (int i = 0; i < 4; ++i) unsafeAtomicAdd(dst + i, val[i]);
    } else {
      *reinterpret_cast<float4 *>(dst) = make_float4(val[0], val[1], val[2], val[3]);
    }
  }
  AT_MARK(1, 11);
}


// ------------------------------------------------------------------------------------------
// backward, SMALL problems (round 5): the 52 x 52 self-attention (32 queries + 20 question tokens) and the 32 x 256
// cross-attention of the SQA3D shape, 18 launches per step.  The generic kernel above runs them as one wave per SIMD
// on 96 / 192 workgroups whose waves re-read Q, dO and K from memory in the transposed operand layout (~224 load
// instructions per wave: 21-25 us per launch, 3-6 % of the matrix pipe: profiles/r03_pmc_attention_mfma.md).  Here a
// workgroup of EIGHT waves owns one (batch, head) -- and, for many keys, one chunk of 32 NKB keys -- and everything
// it needs sits in LDS once: Q, dO (32 NQB rows), K, V (32 NKB rows), read from memory in 256-byte rows.
//   phase 1: the 2 NQB NKB blocks of S = Q K^T and dP = dO V^T, one per wave (NQB NKB = 4);
//   phase 2: P = exp(S scale + mask - lse), dropout bits regenerated, dS = P (dP' - D), element by element in LDS;
//   phase 3: dV = P'^T dO, dK = dS^T Q, dQ = dS K as 32 x 32 blocks dealt over the waves (dQ of the many-key shape in
//            four key slices per block, summed through LDS), stored from the accumulators as 128-byte runs.
// Operand reads are conflict-free by construction: a block's A / B operand is either 32 consecutive floats of one LDS
// row, or one 8-byte pair per row with row strides of 66 / 32 NKB + 2 floats (2 banks per row, 64 banks).
template <int NQB, int NKB>
__global__ __launch_bounds__(512, 1) void attention_bwd_small_kernel(
    int h, int nq, int nk, int q_seg, int k_seg, int q_base2, int k_base2, int q_rows, int k_rows, int ldq, int ldk,
    int ldv, float scale, int atomic_dq, float p_drop, unsigned call_id, const unsigned *__restrict__ rng_counter,
    const float *__restrict__ q, const float *__restrict__ k, const float *__restrict__ v,
    const float *__restrict__ mask, const float *__restrict__ out, const float *__restrict__ lse,
    const float *__restrict__ grad_out, float *__restrict__ dq, float *__restrict__ dk, float *__restrict__ dv) {
  static_assert(NQB * NKB == 4, "eight waves: one S block and one dP block each");
  constexpr int D = 64, RQ = 32 * NQB, RK = 32 * NKB, LD = 66, LP = RK + 2;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float *sQ = sm, *sdO = sQ + RQ * LD, *sK = sdO + RQ * LD, *sV = sK + RK * LD;
  float *sP = sV + RK * LD, *sdP = sP + RQ * LP, *sD = sdP + RQ * LP, *sL = sD + RQ, *sM = sL + RQ;
  float *sRed = sM + RK;   // NQB == 1: [8 waves][16][64] partial dQ blocks

  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = blockIdx.x, kc = blockIdx.y, bi = blockIdx.z;
  const int key0 = kc * RK, nkl = min(RK, nk - key0);
  if (kc == 0 && bi == 0) {
    zero_pad_rows<D>(dq, (unsigned)ldq, hi, gridDim.z, nq, q_seg, q_base2, q_rows);
    zero_pad_rows<D>(dk, (unsigned)ldk, hi, gridDim.z, nk, k_seg, k_base2, k_rows);
    zero_pad_rows<D>(dv, (unsigned)ldv, hi, gridDim.z, nk, k_seg, k_base2, k_rows);
  }
  const size_t bh = (size_t)(bi * h + hi);
  const unsigned ostride = (unsigned)h * D;
  const float *Q = q + hi * D, *K = k + hi * D, *V = v + hi * D, *O = out + hi * D, *dO = grad_out + hi * D;
  auto qrow = [&](int i) { return tok_row(i, bi, nq, q_seg, q_base2); };
  auto krow_of = [&](int j) { return tok_row(j, bi, nk, k_seg, k_base2); };

  // ---- phase 0: everything into LDS.  Sixteen lanes per 256-byte row; every request is issued before the first store
  // (rows beyond the problem are clamped and zeroed at store time); D = rowsum(dO O) falls out of the same loads.
  {
    constexpr int QI = RQ * 16 / 512, KI = RK * 16 / 512;
    float4 aq[QI], ag[QI], ao[QI], ak[KI], av[KI];
#pragma unroll
    for (int j = 0; j < QI; ++j) {
      const int i = tid + 512 * j, row = i >> 4, c4 = (i & 15) * 4;
      const unsigned rr = qrow(min(row, nq - 1));
      aq[j] = *reinterpret_cast<const float4 *>(Q + rr * (unsigned)ldq + c4);
      ag[j] = *reinterpret_cast<const float4 *>(dO + rr * ostride + c4);
      ao[j] = *reinterpret_cast<const float4 *>(O + rr * ostride + c4);
    }
#pragma unroll
    for (int j = 0; j < KI; ++j) {
      const int i = tid + 512 * j, row = i >> 4, c4 = (i & 15) * 4;
      const unsigned rr = krow_of(min(key0 + row, nk - 1));
      ak[j] = *reinterpret_cast<const float4 *>(K + rr * (unsigned)ldk + c4);
      av[j] = *reinterpret_cast<const float4 *>(V + rr * (unsigned)ldv + c4);
    }
    if (tid < RQ) sL[tid] = tid < nq ? lse[bh * nq + tid] : 0.f;
    if (tid < RK) sM[tid] = (mask && tid < nkl) ? mask[(size_t)bi * nk + key0 + tid] : 0.f;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < QI; ++j) {
      const int i = tid + 512 * j, row = i >> 4, c4 = (i & 15) * 4;
      const bool ok = row < nq;
      const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 x = ok ? aq[j] : z, g = ok ? ag[j] : z, o = ok ? ao[j] : z;
      *reinterpret_cast<float2 *>(sQ + row * LD + c4) = make_float2(x.x, x.y);
      *reinterpret_cast<float2 *>(sQ + row * LD + c4 + 2) = make_float2(x.z, x.w);
      *reinterpret_cast<float2 *>(sdO + row * LD + c4) = make_float2(g.x, g.y);
      *reinterpret_cast<float2 *>(sdO + row * LD + c4 + 2) = make_float2(g.z, g.w);
      const float part = row_allreduce_sum_f32(g.x * o.x + g.y * o.y + g.z * o.z + g.w * o.w);   // 16 lanes = one row
      if ((i & 15) == 0) sD[row] = part;
    }
#pragma unroll
    for (int j = 0; j < KI; ++j) {
      const int i = tid + 512 * j, row = i >> 4, c4 = (i & 15) * 4;
      const bool ok = row < nkl;
      const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 x = ok ? ak[j] : z, y = ok ? av[j] : z;
      *reinterpret_cast<float2 *>(sK + row * LD + c4) = make_float2(x.x, x.y);
      *reinterpret_cast<float2 *>(sK + row * LD + c4 + 2) = make_float2(x.z, x.w);
      *reinterpret_cast<float2 *>(sV + row * LD + c4) = make_float2(y.x, y.y);
      *reinterpret_cast<float2 *>(sV + row * LD + c4 + 2) = make_float2(y.z, y.w);
    }
  }
  __syncthreads();

  // ---- phase 1: wave w < 4: S block (qb, kb) -> P; wave w >= 4: dP block.  Reduction index of step (j, e):
  // d = 4 j + 2 half + e -- the same map for both operands, read as 8-byte pairs.
  {
    const int blk = wave & 3, qb = blk / NKB, kb = blk % NKB;
    const bool is_s = wave < 4;
    const float *A = (is_s ? sQ : sdO) + (32 * qb + l31) * LD + 2 * half;
    const float *B = (is_s ? sK : sV) + (32 * kb + l31) * LD + 2 * half;
    f32x16 acc = {0};
#pragma unroll
    for (int j = 0; j < D / 4; ++j) {
      const float2 a2 = *reinterpret_cast<const float2 *>(A + 4 * j), b2 = *reinterpret_cast<const float2 *>(B + 4 * j);
      acc = mfma32(a2.x, b2.x, acc);
      acc = mfma32(a2.y, b2.y, acc);
    }
    const int key = 32 * kb + l31;
    const bool key_ok = key < nkl;
    const float mk = sM[key];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qq = 32 * qb + mfma_row(r, half);
      float val = acc[r];
      if (is_s) {
        const float row_lse = sL[qq];   // fully fill-masked rows attend uniformly (see the generic kernel)
        const float e = row_lse < -1e8f ? 1.f / (float)nk : __expf(val * scale + mk - row_lse);
        val = (key_ok && qq < nq) ? e : 0.f;
      }
      (is_s ? sP : sdP)[qq * LP + key] = val;
    }
  }
  __syncthreads();

  // ---- phase 2: P' = P keep / (1 - p) (for dV), dS = P (dP keep / (1 - p) - D), in place
  {
    const AttnDropout drop = make_dropout(p_drop, call_id, rng_counter);
#pragma unroll
    for (int j = 0; j < RQ * RK / 512; ++j) {
      const int e = tid + 512 * j, qq = e / RK, key = e - qq * RK;
      const float p = sP[qq * LP + key];
      float dp = sdP[qq * LP + key], pd = p;
      if (drop.on) {
        const unsigned row_base = ((unsigned)((bi * h + hi) * nq + qq)) * (unsigned)((nk + 1) >> 1);
        const bool keep = at_keep(drop, row_base, key0 + key);
        dp = keep ? dp * drop.inv_keep : 0.f;
        pd = keep ? p * drop.inv_keep : 0.f;
      }
      float ds = p * (dp - sD[qq]);
      if (sL[qq] < -1e8f) ds = 0.f;
      sP[qq * LP + key] = pd;
      sdP[qq * LP + key] = ds;
    }
  }
  __syncthreads();

  // ---- phase 3.  dV / dK block (kb, db): rows = keys, A = P' / dS column slice (32 consecutive keys of one query row),
  // B = dO / Q row slice; dQ block (qb, db): rows = queries, A = dS read as 8-byte pairs along the keys, B = K rows.
  auto keys_block = [&](const float *A_, const float *B_, int kb, int db, float f, float *dst, int ld_) {
    f32x16 acc = {0};
    const float *a = A_ + 32 * kb + l31, *b = B_ + 32 * db + l31;
#pragma unroll
    for (int s = 0; s < RQ / 2; ++s) acc = mfma32(a[(2 * s + half) * LP], b[(2 * s + half) * LD], acc);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = 32 * kb + mfma_row(r, half);
      if (key < nkl) dst[krow_of(key0 + key) * (unsigned)ld_ + hi * D + 32 * db + l31] = acc[r] * f;
    }
  };
  auto dq_slice = [&](int qb, int db, int k_lo, int k_n) {
    f32x16 acc = {0};
    const float *a = sdP + (32 * qb + l31) * LP + k_lo + 2 * half, *b = sK + (k_lo + 2 * half) * LD + 32 * db + l31;
    for (int j = 0; j < k_n / 4; ++j) {
      const float2 a2 = *reinterpret_cast<const float2 *>(a + 4 * j);
      acc = mfma32(a2.x, b[(4 * j) * LD], acc);
      acc = mfma32(a2.y, b[(4 * j + 1) * LD], acc);
    }
    return acc;
  };
  auto dq_store = [&](const f32x16 &acc, int qb, int db) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qq = 32 * qb + mfma_row(r, half);
      if (qq < nq) {
        float *dst = dq + qrow(qq) * (unsigned)ldq + hi * D + 32 * db + l31;
        if (atomic_dq) unsafeAtomicAdd(dst, acc[r] * scale);
        else *dst = acc[r] * scale;
      }
    }
  };
  if constexpr (NQB == 2) {        // 12 blocks over 8 waves: dV 0-3, dK 4-7, dQ 8-11
    for (int it = wave; it < 12; it += 8) {
      const int kind = it >> 2, b2 = it & 3;
      if (kind == 0) keys_block(sP, sdO, b2 >> 1, b2 & 1, 1.f, dv, ldv);
      else if (kind == 1) keys_block(sdP, sQ, b2 >> 1, b2 & 1, scale, dk, ldk);
      else dq_store(dq_slice(b2 >> 1, b2 & 1, 0, RK), b2 >> 1, b2 & 1);
    }
  } else {                         // one query block, four key blocks: dV and dK block w each, a quarter of a dQ block
    keys_block(sP, sdO, wave >> 1, wave & 1, 1.f, dv, ldv);
    keys_block(sdP, sQ, wave >> 1, wave & 1, scale, dk, ldk);
    const f32x16 part = dq_slice(0, wave & 1, 32 * (wave >> 1), 32);
#pragma unroll
    for (int r = 0; r < 16; ++r) sRed[(wave * 16 + r) * 64 + lane] = part[r];
    __syncthreads();
    if (wave < 2) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        acc[r] = (sRed[((wave + 0) * 16 + r) * 64 + lane] + sRed[((wave + 2) * 16 + r) * 64 + lane]) +
                 (sRed[((wave + 4) * 16 + r) * 64 + lane] + sRed[((wave + 6) * 16 + r) * 64 + lane]);
      dq_store(acc, 0, wave);
    }
  }
}

template <int NQB, int NKB>
constexpr size_t attention_bwd_small_lds() {
  constexpr int RQ = 32 * NQB, RK = 32 * NKB;
  return sizeof(float) * (2 * RQ * 66 + 2 * RK * 66 + 2 * RQ * (RK + 2) + 2 * RQ + RK + (NQB == 1 ? 8 * 16 * 64 : 0));
}

// read per call (host side, once per launch or capture): tools/ab_step.py env:SIG3D_ATTN_BWD_SMALL 0 1 builds both steps
inline bool attn_bwd_small_enabled() {
  const char *e = getenv("SIG3D_ATTN_BWD_SMALL");
  return e == nullptr || atoi(e) != 0;
}

}  // namespace

#ifdef SIG3D_ATTN_TIMING
extern "C" int sig3d_debug_attention_marks(unsigned long long *host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_at_marks), sizeof(unsigned long long) * 2 * 4 * 16);
}
#endif

extern "C" int sig3d_attention_fwd(int b, int h, int nq, int nk, int d, int q_seg, int k_seg, int q_base2,
                                   int k_base2, int q_rows, int k_rows, int ldq, int ldk, int ldv,
                                   float scale, const float *q, const float *k, const float *v,
                                   const float *mask, float *out, float *lse, float p_drop,
                                   unsigned call_id, const unsigned *rng_counter, int key_splits,
                                   float *workspace, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SIG3D_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "dropout probability must be in [0, 1)");
  SIG3D_REQUIRE(b >= 0 && h >= 0 && nq >= 0 && nk >= 0, "negative size");
  SIG3D_REQUIRE(d == 64 || d == 96, "attention head size must be 64 or 96");
  SIG3D_REQUIRE(q_seg >= 0 && q_seg <= nq && k_seg >= 0 && k_seg <= nk, "segment sizes must be in [0, n]");
  SIG3D_REQUIRE((q_seg == 0 || q_seg == nq || q_base2 >= b * q_seg) && (k_seg == 0 || k_seg == nk || k_base2 >= b * k_seg),
                "the second segment must start at or after the end of the first");
  SIG3D_REQUIRE(ldq >= h * d && ldk >= h * d && ldv >= h * d && ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0,
                "row strides must be >= h*d and multiples of 4 floats");
  if (b == 0 || h == 0 || nq == 0) return 0;
  SIG3D_REQUIRE(nk >= 1, "attention needs at least one key");
  SIG3D_REQUIRE((long)(b * nq + q_base2 + q_rows) * ldq < (1L << 31) && (long)(b * nk + k_base2 + k_rows) * ldk < (1L << 31) &&
                    (long)(b * nk + k_base2 + k_rows) * ldv < (1L << 31),
                "operand extents must stay below 2^31 floats (32-bit addressing inside the kernel)");
  if (key_splits < 1) key_splits = 1;
  const int ntiles_fwd = (nk + 31) / 32;
  if (key_splits > ntiles_fwd) key_splits = ntiles_fwd;
  if (key_splits > 1) {   // no split may be left without a tile: ceil(tiles / splits) per split covers the range early
    const int tps = (ntiles_fwd + key_splits - 1) / key_splits;
    key_splits = (ntiles_fwd + tps - 1) / tps;
  }
  SIG3D_REQUIRE(key_splits == 1 || workspace != nullptr,
                "key_splits > 1 needs a workspace of b*h*roundup32(nq)*key_splits*(d+2) floats");
  dim3 grid(h, ((nq + 31) / 32) * key_splits, b);
  (void)k_rows;
  const long rows = (long)b * h * nq;
  // rotating K / V prefetch when a wave streams many key tiles (3D-LLM shapes)
  const int tiles_per_wave = ntiles_fwd / (key_splits * AT_WAVES);
  // (head size 64 only: at 96 the rotating form spills)
  const bool pipe = d == 64 && tiles_per_wave >= 4;
#define SIG3D_ATT_FWD(DD)                                                                                         \
  if (pipe)                                                                                                       \
    hipLaunchKernelGGL((attention_fwd_kernel<64, true>), grid, dim3(AT_WAVES * 64), 0, stream, h, nq, nk, q_seg,   \
                       k_seg, q_base2, k_base2, q_rows, ldq, ldk, ldv, scale, p_drop, call_id, rng_counter, q, k,  \
                       v, mask, out, lse, key_splits, workspace);                                                  \
  else                                                                                                            \
    hipLaunchKernelGGL((attention_fwd_kernel<DD, false>), grid, dim3(AT_WAVES * 64), 0, stream, h, nq, nk, q_seg,  \
                       k_seg, q_base2, k_base2, q_rows, ldq, ldk, ldv, scale, p_drop, call_id, rng_counter, q, k,  \
                       v, mask, out, lse, key_splits, workspace)
  if (d == 64) {
    SIG3D_ATT_FWD(64);
    if (key_splits > 1)
      hipLaunchKernelGGL(attention_combine_kernel<64>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, h, nq,
                         q_seg, q_base2, key_splits, b, workspace, out, lse);
  } else {
    SIG3D_ATT_FWD(96);
    if (key_splits > 1)
      hipLaunchKernelGGL(attention_combine_kernel<96>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, h, nq,
                         q_seg, q_base2, key_splits, b, workspace, out, lse);
  }
#undef SIG3D_ATT_FWD
  SIG3D_LAUNCH_CHECK("attention_fwd_kernel");
  return 0;
}

// dq_zeroed: the caller hands over dq rows that are zero already (sig3d_attention_bwd_z)
static int attention_bwd_impl(int b, int h, int nq, int nk, int d, int q_seg, int k_seg, int q_base2,
                              int k_base2, int q_rows, int k_rows, int ldq, int ldk, int ldv,
                              float scale, const float *q, const float *k, const float *v,
                              const float *mask, const float *out, const float *lse,
                              const float *grad_out, float *dq, float *dk, float *dv,
                              float p_drop, unsigned call_id, const unsigned *rng_counter,
                              bool dq_zeroed, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SIG3D_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "dropout probability must be in [0, 1)");
  SIG3D_REQUIRE(b >= 0 && h >= 0 && nq >= 0 && nk >= 0, "negative size");
  SIG3D_REQUIRE(d == 64 || d == 96, "attention head size must be 64 or 96");
  SIG3D_REQUIRE(q_seg >= 0 && q_seg <= nq && k_seg >= 0 && k_seg <= nk, "segment sizes must be in [0, n]");
  SIG3D_REQUIRE((q_seg == 0 || q_seg == nq || q_base2 >= b * q_seg) && (k_seg == 0 || k_seg == nk || k_base2 >= b * k_seg),
                "the second segment must start at or after the end of the first");
  if (b == 0 || h == 0 || nq == 0) return 0;
  SIG3D_REQUIRE(nk >= 1, "attention needs at least one key");
  SIG3D_REQUIRE((long)(b * nq + q_base2 + q_rows) * ldq < (1L << 31) && (long)(b * nk + k_base2 + k_rows) * ldk < (1L << 31) &&
                    (long)(b * nk + k_base2 + k_rows) * ldv < (1L << 31),
                "operand extents must stay below 2^31 floats (32-bit addressing inside the kernel)");
  SIG3D_REQUIRE(ldq >= h * d && ldk >= h * d && ldv >= h * d && ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0,
                "row strides must be >= h*d and multiples of 4 floats");
  // small problems: one eight-wave workgroup per (batch, head[, chunk of 128 keys]) with everything in LDS
  if (attn_bwd_small_enabled() && d == 64 && ((nq <= 64 && nk <= 64) || (nq <= 32 && nk <= 1024))) {
    const bool self_like = nq > 32 || nk <= 64;
    const int chunks = self_like ? 1 : (nk + 127) / 128;
    if (chunks > 1 && !dq_zeroed) {  // dq is accumulated with atomics across key chunks
      const int q_extent = (q_seg > 0 && q_seg < nq) ? q_base2 + b * (nq - q_seg) : b * nq;
      SIG3D_HIP_TRY(hipMemset2DAsync(dq, sizeof(float) * ldq, 0, sizeof(float) * h * d,
                                     (size_t)(q_rows > q_extent ? q_rows : q_extent), stream));
    }
    constexpr size_t lds22 = attention_bwd_small_lds<2, 2>(), lds14 = attention_bwd_small_lds<1, 4>();
    static sig3d_once_per_device small_attr;
    if (small_attr.pending()) {
      SIG3D_HIP_TRY(hipFuncSetAttribute((const void *)attention_bwd_small_kernel<2, 2>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds22));
      SIG3D_HIP_TRY(hipFuncSetAttribute((const void *)attention_bwd_small_kernel<1, 4>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds14));
      small_attr.done();
    }
    if (self_like)
      hipLaunchKernelGGL((attention_bwd_small_kernel<2, 2>), dim3(h, 1, b), dim3(512), lds22, stream,
                         h, nq, nk, q_seg, k_seg, q_base2, k_base2, q_rows, k_rows, ldq, ldk, ldv, scale, 0, p_drop, call_id,
                         rng_counter, q, k, v, mask, out, lse, grad_out, dq, dk, dv);
    else
      hipLaunchKernelGGL((attention_bwd_small_kernel<1, 4>), dim3(h, chunks, b), dim3(512), lds14,
                         stream, h, nq, nk, q_seg, k_seg, q_base2, k_base2, q_rows, k_rows, ldq, ldk, ldv, scale,
                         chunks > 1 ? 1 : 0, p_drop, call_id, rng_counter, q, k, v, mask, out, lse, grad_out, dq, dk, dv);
    SIG3D_LAUNCH_CHECK("attention_bwd_small_kernel");
    return 0;
  }
  const int ntiles = (nk + 31) / 32;
  // enough workgroups to cover the chip, but at least AT_WAVES tiles per workgroup
  int splits = 1;
  const long bh = (long)b * h;
  if (bh < 512) {
    splits = (int)((1024 + bh - 1) / bh);
    const int max_splits = (ntiles + AT_WAVES - 1) / AT_WAVES;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
  }
  const int tiles_per_split = (ntiles + splits - 1) / splits;
  splits = (ntiles + tiles_per_split - 1) / tiles_per_split;
  SIG3D_REQUIRE(ldq >= h * d && ldk >= h * d && ldv >= h * d && ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0,
                "row strides must be >= h*d and multiples of 4 floats");
  if (splits > 1 && !dq_zeroed)  // dq is accumulated with atomics across key splits: zero its (possibly strided) rows
  {
    const int q_extent = (q_seg > 0 && q_seg < nq) ? q_base2 + b * (nq - q_seg) : b * nq;
    SIG3D_HIP_TRY(hipMemset2DAsync(dq, sizeof(float) * ldq, 0, sizeof(float) * h * d,
                                   (size_t)(q_rows > q_extent ? q_rows : q_extent), stream));
  }
  // query chunks: the per-row statistics and the four dQ images of a chunk must fit in LDS
  const int chunk_max = d == 64 ? AT_NQ_MAX : 64;
  const int nq_pad = (nq + 31) / 32 * 32;
  const int nchunks = (nq + chunk_max - 1) / chunk_max;
  const int qchunk = nchunks == 1 ? nq_pad : chunk_max;
  if (nchunks > 1) {  // dk / dv are accumulated with atomics across query chunks
    const int k_extent = (k_seg > 0 && k_seg < nk) ? k_base2 + b * (nk - k_seg) : b * nk;
    const size_t krows = (size_t)(k_rows > k_extent ? k_rows : k_extent);
    SIG3D_HIP_TRY(hipMemset2DAsync(dk, sizeof(float) * ldk, 0, sizeof(float) * h * d, krows, stream));
    SIG3D_HIP_TRY(hipMemset2DAsync(dv, sizeof(float) * ldv, 0, sizeof(float) * h * d, krows, stream));
  }
  dim3 grid(h, splits * nchunks, b);
  const size_t img = sizeof(float) * AT_WAVES * (size_t)qchunk * (d + 1);
  const size_t red_bytes = sizeof(float) * 2 * (2 * (d / 32) * 16) * 64;  // dK/dV partials of the q-split mode
  const int qsplit_max = (qchunk <= 64 && img + red_bytes <= 136 * 1024) ? 2 : 1;
  const size_t lds = img + (qsplit_max == 2 ? red_bytes : 0);
  SIG3D_REQUIRE(lds <= 136 * 1024, "attention backward: LDS budget exceeded");
  static sig3d_once_per_device attr_done;
  if (attr_done.pending()) {
    SIG3D_HIP_TRY(hipFuncSetAttribute((const void *)attention_bwd_kernel<64, false>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 136 * 1024));
    SIG3D_HIP_TRY(hipFuncSetAttribute((const void *)attention_bwd_kernel<64, true>,   // + 36 KB of static LDS
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    SIG3D_HIP_TRY(hipFuncSetAttribute((const void *)attention_bwd_kernel<96, false>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 136 * 1024));
    attr_done.done();
  }
  // one query tile and whole key tiles per wave: the register-resident path (see the kernel)
  const bool oneqt = d == 64 && nq <= 32;
#define SIG3D_ATT_BWD(DD, ONE)                                                                                    \
  hipLaunchKernelGGL((attention_bwd_kernel<DD, ONE>), grid, dim3(AT_WAVES * 64), lds, stream, h, nq, nk, q_seg,    \
                     k_seg, q_base2, k_base2, q_rows, k_rows, ldq, ldk, ldv, scale, tiles_per_split, splits,      \
                     qchunk, splits > 1 ? 1 : 0, nchunks > 1 ? 1 : 0, qsplit_max, p_drop, call_id, rng_counter, q, \
                     k, v, mask, out, lse, grad_out, dq, dk, dv)
  if (d == 64 && oneqt) SIG3D_ATT_BWD(64, true);
  else if (d == 64) SIG3D_ATT_BWD(64, false);
  else SIG3D_ATT_BWD(96, false);
#undef SIG3D_ATT_BWD
  SIG3D_LAUNCH_CHECK("attention_bwd_kernel");
  return 0;
}

extern "C" int sig3d_attention_bwd(int b, int h, int nq, int nk, int d, int q_seg, int k_seg, int q_base2,
                                   int k_base2, int q_rows, int k_rows, int ldq, int ldk, int ldv,
                                   float scale, const float *q, const float *k, const float *v,
                                   const float *mask, const float *out, const float *lse,
                                   const float *grad_out, float *dq, float *dk, float *dv,
                                   float p_drop, unsigned call_id, const unsigned *rng_counter,
                                   void *stream_) {
  return attention_bwd_impl(b, h, nq, nk, d, q_seg, k_seg, q_base2, k_base2, q_rows, k_rows, ldq, ldk, ldv, scale, q, k, v,
                            mask, out, lse, grad_out, dq, dk, dv, p_drop, call_id, rng_counter, false, stream_);
}

// Same, for a dq whose rows the caller has ZEROED (one fill per step for all such buffers: scratch.py): when the keys
// are split over workgroups dq is accumulated with atomics and this entry point does not clear it first.
extern "C" int sig3d_attention_bwd_z(int b, int h, int nq, int nk, int d, int q_seg, int k_seg, int q_base2,
                                     int k_base2, int q_rows, int k_rows, int ldq, int ldk, int ldv,
                                     float scale, const float *q, const float *k, const float *v,
                                     const float *mask, const float *out, const float *lse,
                                     const float *grad_out, float *dq, float *dk, float *dv,
                                     float p_drop, unsigned call_id, const unsigned *rng_counter,
                                     void *stream_) {
  return attention_bwd_impl(b, h, nq, nk, d, q_seg, k_seg, q_base2, k_base2, q_rows, k_rows, ldq, ldk, ldv, scale, q, k, v,
                            mask, out, lse, grad_out, dq, dk, dv, p_drop, call_id, rng_counter, true, stream_);
}
