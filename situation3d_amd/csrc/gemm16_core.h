// gemm16_core.h -- exact-f32 MFMA GEMM core for row counts of a few hundred (gfx950).
//
// Round 4 rebuild of the dense-layer core behind the Q-Former's nn.Linear layers
//   3DLLM_BLIP2-base/lavis/models/blip2_models/Qformer.py:116-118 (query / key / value), :238, :305, :320
// and their input-gradient products.  Design rules, each from a measurement:
//   * the chip has 1024 SIMDs and a 416-row problem has ~1000-1500 16x16 output blocks per 768 columns: the
//     decomposition decides everything.  v_mfma_f32_16x16x4_f32 (32 cycles per instruction per SIMD, 40
//     dependent) lets a wave own AB x BB blocks of 16 x 16 -- 32x32, 32x48, 16x48 ... -- so that every product of
//     the step has a tiling that fills the chip once or twice over with equal pieces;
//   * where the tile grid alone cannot do that the reduction is split over workgroups and every split writes
//     its own SLAB of C (plain coalesced stores, no atomics, no zero-fill); the consumer -- a LayerNorm tail, the
//     attention backward -- adds the slabs while it loads them;
//   * LDS tiles are [row][32 k] with k contiguous (what nn.Linear's operands are in memory): k-contiguous
//     operands are copied with 16-byte loads and stores, operands with the row index contiguous (the weight of
//     an input-gradient product) are turned by their 4-byte LDS stores; every operand read is ONE ds_read_b128
//     per 16 x 16 block per 16 k (four MFMAs), conflict-free through a 16-byte-slot XOR swizzle (PMC: zero bank
//     conflicts, 1.3 % LDS issue stalls);
//   * one workgroup has one wave per SIMD, so nothing hides a latency for it but its own instruction stream: PF
//     chunks are in flight from global memory (a ~2 us HBM round trip is four chunks of MFMA work), three LDS
//     stages let the operand reads of chunk i+1 be issued before the second half of chunk i's MFMAs, addresses
//     are computed once (a first version that recomputed them per chunk spent 20 % of its cycles issuing VALU
//     instructions and 29 % in s_waitcnt / s_barrier, matrix pipe 52 % busy).
// Arithmetic: every product and sum in f32 (bitwise an fmaf chain per k slice, guide section 3).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <type_traits>

namespace gemm16 {

// hipFuncAttributeMaxDynamicSharedMemorySize is set per kernel AND per device: one of these per launch site
struct OncePerDevice {
  std::atomic<unsigned long long> mask{0};
  static unsigned long long current() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) d = 0;
    return 1ull << (d & 63);
  }
  bool pending() const { return !(mask.load(std::memory_order_acquire) & current()); }
  void done() { mask.fetch_or(current(), std::memory_order_release); }
};

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 32;   // k per LDS chunk: 8 slots of 16 bytes per row

enum { B_KC = 0, B_MC = 1 };   // B element (n, k) at B[n*ldb + k]  /  B[k*ldb + n]

struct Problem {
  const float *A;        // (M, K) rows, k contiguous, row stride lda
  const float *B;        // B_KC: (N, K) rows;  B_MC: (K, N) rows
  float *C;              // (M, N) rows, row stride ldc: split 0
  float *Cs;             // splits z >= 1 write Cs + (z - 1) * slab (same row stride and batch stride)
  const float *bias;     // [N] added by split 0 (or null)
  const float *addend;   // same layout as C, added by split 0 (or null); may alias C
  float *aux;            // act 1: gelu input kept here (or null); act 2: gelu input read from here
  int M, N, K;
  int lda, ldb, ldc;
  long sA, sB, sC, sBias;   // batch strides in elements
  long slab;                // split stride in elements
  int batch, splits;
  int act;                  // 0 none, 1 erf-GELU (splits == 1), 2 times gelu'(aux) (splits == 1)
  int ntm, ntn;
  // weight-gradient use (launch<..., DW = true>, sig3d_mlp_layer_dw_stream): the reduction length of batch element i is
  // k_dev[i] (device memory; K is then the row capacity), B rows pass through relu(v * b_scale[n] + b_shift[n]) on their
  // way to LDS, and every (batch, split) pair writes its own slab: pair 0 to C, pair q to Cs + (q - 1) * slab
  const int *k_dev;
  const float *b_scale, *b_shift;
#ifdef GEMM16_TIMING
  unsigned long long *dbg;  // tools/micro/gemm16_bench.hip: cycle stamps of workgroup 0, wave 0
#endif
};

__device__ __forceinline__ float gelu(float u) { return 0.5f * u * (1.f + erff(u * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad(float u) {
  const float cdf = 0.5f * (1.f + erff(u * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * u * u);
  return cdf + u * pdf;
}

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>());
    static_for<I + 1, N>(f);
  }
}

// physical float offset of (row, 16-byte slot q) inside a [rows][32] tile
__device__ __forceinline__ int slot_off(int row, int q) { return row * BK + ((q ^ ((row >> 1) & 7)) << 2); }

// AB x BB blocks of 16 x 16 per wave, WGM x WGN waves per workgroup, PF >= 3 chunks in flight (register ring,
// statically indexed: the loop is unrolled PF times), OCC workgroups per CU the register budget is sized for.
// The kernel's body as a device function of (problem, this workgroup's number, workgroups of the product): a launch may
// carry another kernel's workgroups behind these (shared_mlp.hip: a layer's weight gradient and its input gradient as
// two workgroup ranges of one launch).
template <int AB, int BB, int WGM, int WGN, int PF, int BMODE, bool KEDGE, bool DW>
__device__ __forceinline__ void gemm16_body(const Problem &p, const int block_id, const int T) {
  static_assert(!DW || BMODE == B_KC, "the weight-gradient form streams two k-contiguous operands");
  static_assert(PF >= 3, "chunk c + 2 is stored while chunk c + PF is requested into chunk c's slot");
  constexpr int NW = WGM * WGN, NT = 64 * NW;
  constexpr int TM = 16 * AB * WGM, TN = 16 * BB * WGN;
  constexpr int STAGE = (TM + TN) * BK;         // floats of one chunk of A and B
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [3][TM + TN][BK]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wv / WGN, wn = wv - wm * WGN;

  // ---- which tile.  Workgroups of one XCD (blockIdx % 8) take a contiguous run of work ids; ids are ordered
  // (batch, n tile, split, m tile) with the m tile fastest: neighbours share their weight tile in the XCD's L2.
  const int xcd = block_id & 7;
  int base = 0;
  for (int y = 0; y < xcd; ++y) base += (T - y + 7) >> 3;
  int w = base + (block_id >> 3);
  const int tm = w % p.ntm; w /= p.ntm;
  const int z = w % p.splits; w /= p.splits;
  const int tn = w % p.ntn;
  const int batch = w / p.ntn;
  const int m0 = tm * TM, n0 = tn * TN;

  const float *__restrict__ A = p.A + (size_t)batch * p.sA;
  const float *__restrict__ B = p.B + (size_t)batch * p.sB;
  const int M = p.M, N = p.N, K = (DW && p.k_dev) ? __builtin_amdgcn_readfirstlane(p.k_dev[batch]) : p.K;
  const int nchunks_all = (K + BK - 1) / BK;
  const int c_lo = (int)((long)nchunks_all * z / p.splits), c_hi = (int)((long)nchunks_all * (z + 1) / p.splits);
  const int nchunks = c_hi - c_lo;

  // ---- global -> register -> LDS staging; every address is computed ONCE.  Rows beyond M / N are clamped (their
  // products land in rows / columns that are never stored); k beyond K is zeroed at store time (KEDGE instances).
  constexpr int A_UNITS = TM * 8, A_PER = (A_UNITS + NT - 1) / NT;          // float4 units of a chunk
  constexpr int BKC_UNITS = TN * 8, BKC_PER = (BKC_UNITS + NT - 1) / NT;
  constexpr int BMC_WI = 2 * (TN / 16), BMC_PER = (BMC_WI + NW - 1) / NW;   // wave-instructions (16 k x 16 n)
  constexpr int B_PER = BMODE == B_KC ? BKC_PER : BMC_PER;
  // Buffer loads: a per-lane byte offset computed once + a per-chunk SCALAR offset.  (With flat 64-bit addresses
  // hipcc rebuilt every address per chunk in VGPR pairs that aliased pending load destinations, and waited for
  // nearly the whole queue -- vmcnt(2) of 16 -- once per trip round the ring.)  num_records is the operand's real
  // extent: whatever a clamped row or a k >= K element would read beyond it comes back as zero.
  // The loads are inline asm, i.e. hidden from hipcc's s_waitcnt bookkeeping, and counted by hand (guide 5.7 item 1,
  // form ii): with compiler-counted loads the copy of the loop body at the loop header waited for one chunk more
  // than it needed (vmcnt(7..4) where 11..8 were enough: the header merges the prologue's state), i.e. for a
  // request only ONE iteration old, once per trip round the ring -- 25 % of all wave cycles in s_waitcnt.
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  auto descriptor = [](const float *base, size_t bytes) {
    const unsigned long long a = (unsigned long long)base;
    i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));     // stride 0
    r[2] = __builtin_amdgcn_readfirstlane((int)(unsigned)bytes);
    r[3] = 0x00020000;
    return r;
  };
  const i32x4 rsA = descriptor(A, ((size_t)(M - 1) * p.lda + K) * 4);
  const i32x4 rsB = descriptor(B, (BMODE == B_KC ? (size_t)(N - 1) * p.ldb + K : (size_t)(K - 1) * p.ldb + N) * 4);
  int va[A_PER], vb[B_PER];              // byte offsets of the unit in chunk 0 of this split
  int sa[A_PER], sb[B_PER];              // LDS float offsets inside a stage (B_MC: of the first of four rows)
  int ka[A_PER], kb[B_PER];              // k inside the chunk of the unit's first element (KEDGE)
#pragma unroll
  for (int i = 0; i < A_PER; ++i) {
    const int u = min(tid + NT * i, A_UNITS - 1), row = u >> 3, q = u & 7;
    va[i] = (min(m0 + row, M - 1) * p.lda + c_lo * BK + 4 * q) * 4;
    sa[i] = slot_off(row, q);
    ka[i] = 4 * q;
  }
#pragma unroll
  for (int i = 0; i < B_PER; ++i) {
    if (BMODE == B_KC) {
      const int u = min(tid + NT * i, BKC_UNITS - 1), row = u >> 3, q = u & 7;
      vb[i] = (min(n0 + row, N - 1) * p.ldb + c_lo * BK + 4 * q) * 4;
      sb[i] = TM * BK + slot_off(row, q);
      kb[i] = 4 * q;
    } else {
      // a lane takes four consecutive n at one k: lanes & 3 -> n quad, lanes >> 2 -> k (16 k x 16 n per instruction)
      const int u = min(wv + NW * i, BMC_WI - 1), kh = u & 1, nb = u >> 1;
      const int k = 16 * kh + (lane >> 2), nl = 16 * nb + 4 * (lane & 3);
      vb[i] = ((c_lo * BK + k) * p.ldb + min(n0 + nl, N - 4)) * 4;
      sb[i] = TM * BK + slot_off(nl, k >> 2) + (k & 3);
      kb[i] = k;
    }
  }
  const int a_step = BK * 4, b_step = (BMODE == B_KC ? BK : BK * p.ldb) * 4;   // bytes per chunk
  float bsc[B_PER], bsh[B_PER];          // DW: the BatchNorm + ReLU of the previous layer, applied to B rows on load
  const bool b_pro = DW && p.b_scale != nullptr;
#pragma unroll
  for (int i = 0; i < B_PER; ++i) {
    const int u = min(tid + NT * i, BKC_UNITS - 1), row = min(n0 + (u >> 3), N - 1);
    bsc[i] = b_pro ? p.b_scale[row] : 1.f;
    bsh[i] = b_pro ? p.b_shift[row] : 0.f;
  }

  f32x4 ring_a[PF][A_PER], ring_b[PF][B_PER];
#ifndef GEMM16_KO
#define GEMM16_KO 0   // measurement builds: 1 no global loads, 2 no LDS stores, 4 no LDS reads, 8 no barrier (wrong results)
#endif
  constexpr int NLOAD = A_PER + B_PER;
  constexpr int NDSW = A_PER + (BMODE == B_KC ? B_PER : 4 * B_PER);   // LDS store instructions per chunk
  constexpr int NF = AB + BB, H = 4 * AB * BB;                        // fragment reads / MFMAs per half chunk
  static_assert((PF - 1) * NLOAD <= 63, "vmcnt is a 6-bit counter");

  // One request of chunk c_ (unit i: A units first).  Requests are issued UNCONDITIONALLY, the hand count below
  // relies on it: beyond the last chunk the scalar offset points past the operand, the request stays in the count,
  // touches no memory and returns zeros at once.
  auto load_unit = [&](int c_, auto i_, f32x4 (&ra)[A_PER], f32x4 (&rb)[B_PER]) {
    constexpr int i = decltype(i_)::value;
    if (GEMM16_KO & 1) return;
    (void)va; (void)vb; (void)rsA; (void)rsB;   // clang does not capture what only an asm operand names
    const bool live = c_ < nchunks;
    if constexpr (i < A_PER) {
      const int so = live ? c_ * a_step : rsA[2];
      asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(ra[i]) : "v"(va[i]), "s"(rsA), "s"(so) : "memory");
    } else {
      const int so = live ? c_ * b_step : rsB[2];
      asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(rb[i - A_PER]) : "v"(vb[i - A_PER]), "s"(rsB), "s"(so) : "memory");
    }
  };
  // before the first use of a ring slot: at most `newer` younger requests may still be in flight
  auto wait_chunk = [&](auto newer, f32x4 (&ra)[A_PER], f32x4 (&rb)[B_PER]) {
    if (GEMM16_KO & 1) return;
    asm volatile("s_waitcnt vmcnt(%0)" : : "i"(decltype(newer)::value) : "memory");
#pragma unroll
    for (int i = 0; i < A_PER; ++i) asm volatile("" : "+v"(ra[i]));
#pragma unroll
    for (int i = 0; i < B_PER; ++i) asm volatile("" : "+v"(rb[i]));
  };
  // One LDS store instruction of chunk c (index i over the A units, then the B units; a B_MC unit is four)
  auto store_unit = [&](float *__restrict__ st, int c, auto i_, const f32x4 (&ra)[A_PER], const f32x4 (&rb)[B_PER]) {
    constexpr int i = decltype(i_)::value;
    if (GEMM16_KO & 2) return;
    if constexpr (i < A_PER) {
      if (A_UNITS % NT != 0 && tid + NT * i >= A_UNITS) return;
      f32x4 v = ra[i];
      if (KEDGE) {
        const int k = (c_lo + c) * BK + ka[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = k + e < K ? v[e] : 0.f;
      }
      *reinterpret_cast<f32x4 *>(st + sa[i]) = v;
    } else if constexpr (BMODE == B_KC) {
      constexpr int u = i - A_PER;
      if (BKC_UNITS % NT != 0 && tid + NT * u >= BKC_UNITS) return;
      f32x4 v = rb[u];
      if (DW && b_pro) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(0.f, v[e] * bsc[u] + bsh[u]);
      }
      if (KEDGE) {
        const int k = (c_lo + c) * BK + kb[u];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = k + e < K ? v[e] : 0.f;
      }
      *reinterpret_cast<f32x4 *>(st + sb[u]) = v;
    } else {
      constexpr int u = (i - A_PER) / 4, e = (i - A_PER) % 4;
      if (BMC_WI % NW != 0 && wv + NW * u >= BMC_WI) return;
      const bool ok = !KEDGE || (c_lo + c) * BK + kb[u] < K;
      // rows nl + e, e = 0 .. 3, nl a multiple of 4: the swizzle term (row >> 1) & 7 is even for e = 0, 1 and the
      // next (odd) value for e = 2, 3, i.e. the slot index flips its lowest bit: 4 floats up or down
      const int flip = e >= 2 ? 4 - 8 * ((sb[u] >> 2) & 1) : 0;
      st[sb[u] + e * BK + flip] = ok ? rb[u][e] : 0.f;
    }
  };

  f32x4 acc[AB][BB];
#pragma unroll
  for (int a = 0; a < AB; ++a)
#pragma unroll
    for (int b = 0; b < BB; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // operand read offsets of this lane: row (lane & 15) of a block, slots (lane >> 4) [half 0] and (lane >> 4) + 4 [1]
  const int lr = lane & 15, lg = lane >> 4;
  const int offA = (wm * 16 * AB + lr) * BK, offB = TM * BK + (wn * 16 * BB + lr) * BK;
  const int swz[2] = {((lg) ^ (lr >> 1)) << 2, ((lg + 4) ^ (lr >> 1)) << 2};
  f32x4 fa[2][AB], fb[2][BB];   // fragments of half 0 / half 1
  if (GEMM16_KO) {
    for (int h = 0; h < 2; ++h) {
      for (int a = 0; a < AB; ++a) fa[h][a] = f32x4{1.f * lane, 2.f, 3.f, 4.f};
      for (int b = 0; b < BB; ++b) fb[h][b] = f32x4{1.f, 2.f * lane, 3.f, 4.f};
    }
    for (int d = 0; d < PF; ++d) {
      for (int i = 0; i < A_PER; ++i) ring_a[d][i] = f32x4{1.f * tid, 0.f, 0.f, 0.f};
      for (int i = 0; i < B_PER; ++i) ring_b[d][i] = f32x4{2.f * tid, 0.f, 0.f, 0.f};
    }
  }
  // fragment i of half hh (A blocks first): ONE ds_read_b128
  auto read_frag = [&](const float *st, auto hh_, auto i_) {
    constexpr int hh = decltype(hh_)::value, i = decltype(i_)::value;
    if (GEMM16_KO & 4) return;
    if constexpr (i < AB) fa[hh][i] = *reinterpret_cast<const f32x4 *>(st + offA + i * 16 * BK + swz[hh]);
    else fb[hh][i - AB] = *reinterpret_cast<const f32x4 *>(st + offB + (i - AB) * 16 * BK + swz[hh]);
  };
  // MFMA m of half hh: k step j = m / (AB BB), blocks a, b -- consecutive MFMAs go to different accumulators
  auto mfma_one = [&](auto hh_, auto m_) {
    constexpr int hh = decltype(hh_)::value, m = decltype(m_)::value;
    constexpr int j = m / (AB * BB), a = (m / BB) % AB, b = m % BB;
    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[hh][a][j], fb[hh][b][j], acc[a][b], 0, 0, 0);
  };
  typedef std::integral_constant<int, 0> I0;
  typedef std::integral_constant<int, 1> I1;

  // ---- prologue: PF chunks requested, chunks 0 and 1 staged, half 0 of chunk 0 in registers
#ifdef GEMM16_TIMING
  int stamp_n = 0;
#define GEMM16_STAMP() do { if (block_id == 0 && tid == 0 && stamp_n < 60) p.dbg[stamp_n] = __builtin_readcyclecounter(); ++stamp_n; } while (0)
#else
#define GEMM16_STAMP() do { } while (0)
#endif
  GEMM16_STAMP();
  static_for<0, PF>([&](auto d_) {
    constexpr int d = decltype(d_)::value;
    static_for<0, NLOAD>([&](auto i_) { load_unit(d, i_, ring_a[d], ring_b[d]); });
  });
  wait_chunk(std::integral_constant<int, (PF - 1) * NLOAD>(), ring_a[0], ring_b[0]);
  static_for<0, NDSW>([&](auto i_) { store_unit(smem, 0, i_, ring_a[0], ring_b[0]); });
  wait_chunk(std::integral_constant<int, (PF - 2) * NLOAD>(), ring_a[1], ring_b[1]);
  static_for<0, NDSW>([&](auto i_) { store_unit(smem + STAGE, 1, i_, ring_a[1], ring_b[1]); });
  __syncthreads();
  static_for<0, NF>([&](auto i_) { read_frag(smem, I0(), i_); });
  GEMM16_STAMP();

  // Iteration c: stage c % 3 holds chunk c and stage (c + 1) % 3 chunk c + 1 (published by the last barrier); ring
  // slot c % PF is free (chunk c went to LDS two iterations ago) and takes the requests for chunk c + PF; chunk c + 2
  // (in flight: chunks c + 2 .. c + PF) goes to LDS.  An MFMA holds the matrix pipe for 32 cycles and ONE other
  // instruction (a request, an LDS read or store: ~20-25 cycles of issue each, measured by knocking them out) hides
  // behind it; two in a row do not.  So the body is written slot by slot -- MFMA, at most one filler, pinned by a
  // scheduling barrier (hipcc otherwise bunches the fillers: every LDS read next to its first use, the requests
  // where it likes) -- half 0: the fragments of half 1, then the requests; half 1: the stores, then the next
  // chunk's fragments of half 0.
  constexpr int NFILL0 = NF + NLOAD, NFILL1 = NDSW + NF;
  // fillers go behind the first H - TAIL MFMAs of a half: the last ones cover the latency of the last LDS read
  constexpr int TAIL = H >= 12 ? 4 : 1, HF = H - TAIL;
  int s0 = 0;   // c % 3
  for (int c0 = 0; c0 < nchunks; c0 += PF) {
    static_for<0, PF>([&](auto d_) {
      constexpr int d = decltype(d_)::value, d2 = (d + 2) % PF;
      const int c = c0 + d;
      if (c >= nchunks) return;
      const int s1 = s0 == 2 ? 0 : s0 + 1, s2 = s1 == 2 ? 0 : s1 + 1;
      const float *st0 = smem + s0 * STAGE, *st1 = smem + s1 * STAGE;
      float *st2 = smem + s2 * STAGE;
      static_for<0, H>([&](auto m_) {
        constexpr int m = decltype(m_)::value;
        mfma_one(I0(), m_);
        static_for<(m < HF ? m * NFILL0 / HF : NFILL0), (m < HF ? (m + 1) * NFILL0 / HF : NFILL0)>([&](auto k_) {
          constexpr int k = decltype(k_)::value;
          if constexpr (k < NF) read_frag(st0, I1(), k_);
          else load_unit(c + PF, std::integral_constant<int, k - NF>(), ring_a[d], ring_b[d]);
        });
        __builtin_amdgcn_sched_barrier(0);
      });
      wait_chunk(std::integral_constant<int, (PF - 2) * NLOAD>(), ring_a[d2], ring_b[d2]);
      static_for<0, H>([&](auto m_) {
        constexpr int m = decltype(m_)::value;
        mfma_one(I1(), m_);
        static_for<(m < HF ? m * NFILL1 / HF : NFILL1), (m < HF ? (m + 1) * NFILL1 / HF : NFILL1)>([&](auto k_) {
          constexpr int k = decltype(k_)::value;
          if constexpr (k < NDSW) store_unit(st2, c + 2, k_, ring_a[d2], ring_b[d2]);
          else read_frag(st1, I0(), std::integral_constant<int, k - NDSW>());
        });
        __builtin_amdgcn_sched_barrier(0);
      });
      if (!(GEMM16_KO & 8)) __syncthreads();
      GEMM16_STAMP();
      s0 = s1;
    });
  }

  // requests hipcc does not know about may still be in flight (the out-of-range ones of the last iterations): their
  // destination registers are free game for the epilogue unless they have landed
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  GEMM16_STAMP();

  // ---- epilogue.  C/D map of the 16x16 MFMA: column = lane & 15, row = 4 (lane >> 4) + register.
  const int pair = batch * p.splits + z;    // DW: every (batch, split) pair has a slab of its own, summed by the caller
  float *__restrict__ C = DW ? (pair == 0 ? p.C : p.Cs + (size_t)(pair - 1) * p.slab)
                             : (z == 0 ? p.C : p.Cs + (size_t)(z - 1) * p.slab) + (size_t)batch * p.sC;
  const float *bias = (p.bias && z == 0) ? p.bias + (size_t)batch * p.sBias : nullptr;
  const float *addend = (p.addend && z == 0) ? p.addend + (size_t)batch * p.sC : nullptr;
  float *aux = p.aux ? p.aux + (size_t)batch * p.sC : nullptr;
  const int act = p.act;
#pragma unroll
  for (int a = 0; a < AB; ++a) {
    const int row0 = m0 + (wm * AB + a) * 16 + 4 * lg;
#pragma unroll
    for (int b = 0; b < BB; ++b) {
      const int col = n0 + (wn * BB + b) * 16 + lr;
      const bool col_ok = col < N;
      const float bv = (bias && col_ok) ? bias[col] : 0.f;
      float cin[4] = {0.f, 0.f, 0.f, 0.f}, xin[4] = {0.f, 0.f, 0.f, 0.f};
      if (addend) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (col_ok && row0 + r < M) cin[r] = addend[(size_t)(row0 + r) * p.ldc + col];
      }
      if (act == 2) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (col_ok && row0 + r < M) xin[r] = aux[(size_t)(row0 + r) * p.ldc + col];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (!(col_ok && row0 + r < M)) continue;
        float v = acc[a][b][r] + bv;
        if (act == 1) {
          if (aux) aux[(size_t)(row0 + r) * p.ldc + col] = v;
          v = gelu(v);
        } else if (act == 2) {
          v *= gelu_grad(xin[r]);
        }
        C[(size_t)(row0 + r) * p.ldc + col] = v + cin[r];
      }
    }
  }
  GEMM16_STAMP();
#undef GEMM16_STAMP
}

template <int AB, int BB, int WGM, int WGN, int PF, int OCC, int BMODE, bool KEDGE, bool DW = false>
__global__ __launch_bounds__(64 * WGM * WGN, OCC) void gemm16_kernel(const Problem p) {
  gemm16_body<AB, BB, WGM, WGN, PF, BMODE, KEDGE, DW>(p, (int)blockIdx.x, (int)gridDim.x);
}

template <int AB, int BB, int WGM, int WGN>
constexpr size_t lds_bytes() {
  return (size_t)3 * (16 * AB * WGM + 16 * BB * WGN) * BK * sizeof(float);
}

// Launch one configuration.  Returns hipSuccess or the launch error.
template <int AB, int BB, int WGM, int WGN, int PF, int OCC, bool DW = false>
hipError_t launch(Problem p, int bmode, hipStream_t stream) {
  constexpr int TM = 16 * AB * WGM, TN = 16 * BB * WGN;
  constexpr size_t lds = lds_bytes<AB, BB, WGM, WGN>();
  p.ntm = (p.M + TM - 1) / TM;
  p.ntn = (p.N + TN - 1) / TN;
  const unsigned grid = (unsigned)(p.ntm * p.ntn * p.splits * p.batch);
  if (grid == 0) return hipSuccess;
  const bool kedge = (p.K % BK) != 0;
#define GEMM16_GO(BM, KE)                                                                                   \
  do {                                                                                                      \
    auto kern = gemm16_kernel<AB, BB, WGM, WGN, PF, OCC, BM, KE, DW>;                                       \
    static gemm16::OncePerDevice attr_done;                                                                 \
    if (lds > 64 * 1024 && attr_done.pending()) {                                                           \
      hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,    \
                                         (int)lds);                                                         \
      if (e != hipSuccess) return e;                                                                        \
      attr_done.done();                                                                                     \
    }                                                                                                       \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WGM * WGN), lds, stream, p);                             \
  } while (0)
  if constexpr (DW) {
    // lengths from device memory (compact lists) are arbitrary: masked chunks; whole 32-deep chunks need no masking
    if (kedge || p.k_dev) GEMM16_GO(B_KC, true); else GEMM16_GO(B_KC, false);
  } else {
    if (bmode == B_KC) { if (kedge) GEMM16_GO(B_KC, true); else GEMM16_GO(B_KC, false); }
    else { if (kedge) GEMM16_GO(B_MC, true); else GEMM16_GO(B_MC, false); }
  }
#undef GEMM16_GO
  return hipGetLastError();
}

}  // namespace gemm16
