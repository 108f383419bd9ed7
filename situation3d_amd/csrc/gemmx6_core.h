// gemmx6_core.h -- the f32 GEMM of gemm16_core.h on the bf16 matrix cores: three-term split, six products (gfx950).
//
// Same products (Qformer.py:116-118, :238, :305, :320 and their input gradients), same Problem / epilogues / slabs.
// An f32 number is the EXACT sum of three bf16 numbers (a = a1 + a2 + a3: round to nearest, subtract, repeat: 8 + 8 + 8
// significand bits); of the nine cross products of two such sums the six with i + j <= 4 are kept -- the dropped ones are
// below 2^-27 of |a b| -- each of them is exact in f32 (8 x 8 bits) and they are accumulated in f32 by the matrix core,
// smallest terms first.  Measured on the step's shapes the result is 2-3 x closer to the float64 product than an f32
// GEMM's: this is f32 arithmetic, not a reduced-precision mode.
// Why: v_mfma_f32_32x32x16_bf16 does 16 x the multiply-adds of v_mfma_f32_16x16x4_f32 per cycle; six of them per f32
// product leave 2.7 x.  tools/micro/bf16x6_rate.hip: the read + MFMA stream of a 32 x 64 wave tile runs at 340
// f32-equivalent TFLOP/s with ONE wave per SIMD (the f32 stream of gemm16: 112-130), 285 for two waves of 32 x 32.
// What it costs: the split is made while a chunk travels from the register ring to LDS -- 11 VALU instructions per pair
// of elements (v_cvt_pk_bf16_f32 x 3, two unpacks and two subtractions x 2) -- and LDS holds 6 bytes per element.
//
// LDS image of a chunk (32 k): [plane 3][k octet 4][row TM + TN][8 bf16 = 16 bytes], the octet stride padded by 16 bytes.
// A lane of v_mfma_f32_32x32x16_bf16 holds 8 consecutive k of row (lane & 31): ONE ds_read_b128 per plane, block and
// K-step, consecutive lanes at consecutive 16-byte slots (conflict-free without a swizzle).  k-contiguous operands are
// stored by (row, octet) units: two 16-byte requests, one ds_write_b128 per plane; n-contiguous weights (the input-
// gradient products) by (n, octet) units: eight 4-byte requests down the k rows -- consecutive lanes at consecutive n, so
// requests and stores are dense -- the transposition costs nothing but narrower requests.
#pragma once
#include "gemm16_core.h"

#ifndef GEMMX6_KO
#define GEMMX6_KO 0   // measurement builds: 2 no split / LDS stores, 4 no LDS reads, 8 no barrier, 16 no MFMAs (wrong results)
#endif

namespace gemmx6 {

using gemm16::B_KC;
using gemm16::B_MC;
using gemm16::BK;
using gemm16::Problem;
using gemm16::static_for;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// (x0, x1) -> three dwords of packed bf16 pairs (low half = x0's term): x = t1 + t2 + t3 exactly
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned &t1, unsigned &t2, unsigned &t3) {
  auto pk = [](float a, float b) {
    const bf16x2 h = __builtin_convertvector(f32x2{a, b}, bf16x2);   // v_cvt_pk_bf16_f32, round to nearest even
    return __builtin_bit_cast(unsigned, h);
  };
  t1 = pk(x0, x1);
  const float r0 = x0 - __builtin_bit_cast(float, t1 << 16), r1 = x1 - __builtin_bit_cast(float, t1 & 0xffff0000u);
  t2 = pk(r0, r1);
  const float s0 = r0 - __builtin_bit_cast(float, t2 << 16), s1 = r1 - __builtin_bit_cast(float, t2 & 0xffff0000u);
  t3 = pk(s0, s1);
}

// MB x NB blocks of 32 x 32 per wave, WGM x WGN waves per workgroup, PF >= 3 chunks in flight.
template <int MB, int NB, int WGM, int WGN, int PF, int BMODE, bool KEDGE>
__global__ __launch_bounds__(64 * WGM * WGN, 1) void gemmx6_kernel(const Problem p) {
  static_assert(PF >= 3, "chunk c + 2 is stored while chunk c + PF is requested into chunk c's slot");
  constexpr int NW = WGM * WGN, NT = 64 * NW;
  constexpr int TM = 32 * MB * WGM, TN = 32 * NB * WGN, TR = TM + TN;
  constexpr int PS = TR * 16 + 16;              // bytes of one k octet of one plane (padded)
  constexpr int PLANE = 4 * PS, STAGE = 3 * PLANE;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [3 stages][3 planes][4 octets][TR][16 bytes]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wv / WGN, wn = wv - wm * WGN;

  // ---- which tile (as gemm16_kernel: contiguous runs of work ids per XCD, m tile fastest)
  const int T = gridDim.x;
  const int xcd = blockIdx.x & 7;
  int base = 0;
  for (int y = 0; y < xcd; ++y) base += (T - y + 7) >> 3;
  int w = base + (blockIdx.x >> 3);
  const int tm = w % p.ntm; w /= p.ntm;
  const int z = w % p.splits; w /= p.splits;
  const int tn = w % p.ntn;
  const int batch = w / p.ntn;
  const int m0 = tm * TM, n0 = tn * TN;

  const float *__restrict__ A = p.A + (size_t)batch * p.sA;
  const float *__restrict__ B = p.B + (size_t)batch * p.sB;
  const int M = p.M, N = p.N, K = p.K;
  const int nchunks_all = (K + BK - 1) / BK;
  const int c_lo = (int)((long)nchunks_all * z / p.splits), c_hi = (int)((long)nchunks_all * (z + 1) / p.splits);
  const int nchunks = c_hi - c_lo;

  // ---- global -> register ring -> (split) -> LDS.  A unit = 8 consecutive k of one row (one k octet).
  constexpr int A_UNITS = TM * 4, A_PER = (A_UNITS + NT - 1) / NT;
  constexpr int B_UNITS = TN * 4, B_PER = (B_UNITS + NT - 1) / NT;
  constexpr int BREQ = BMODE == B_KC ? 2 : 8;          // requests per B unit (16 bytes / 4 bytes each)
  auto descriptor = [](const float *base_, size_t bytes) {
    const unsigned long long a = (unsigned long long)base_;
    i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));
    r[2] = __builtin_amdgcn_readfirstlane((int)(unsigned)bytes);
    r[3] = 0x00020000;
    return r;
  };
  const i32x4 rsA = descriptor(A, ((size_t)(M - 1) * p.lda + K) * 4);
  const i32x4 rsB = descriptor(B, (BMODE == B_KC ? (size_t)(N - 1) * p.ldb + K : (size_t)(K - 1) * p.ldb + N) * 4);
  int va[A_PER], vb[B_PER];      // byte offsets of the unit's first request in chunk 0 of this split
  int sa[A_PER], sb[B_PER];      // LDS byte offsets inside a plane
  int ka[A_PER], kb[B_PER];      // k of the unit's first element inside the chunk
#pragma unroll
  for (int i = 0; i < A_PER; ++i) {
    const int u = min(tid + NT * i, A_UNITS - 1), row = u >> 2, ko = u & 3;
    va[i] = (min(m0 + row, M - 1) * p.lda + c_lo * BK + 8 * ko) * 4;
    sa[i] = ko * PS + row * 16;
    ka[i] = 8 * ko;
  }
  const int ldb4 = p.ldb * 4;
#pragma unroll
  for (int i = 0; i < B_PER; ++i) {
    const int u = min(tid + NT * i, B_UNITS - 1);
    if (BMODE == B_KC) {
      const int row = u >> 2, ko = u & 3;
      vb[i] = (min(n0 + row, N - 1) * p.ldb + c_lo * BK + 8 * ko) * 4;
      sb[i] = ko * PS + (TM + row) * 16;
      kb[i] = 8 * ko;
    } else {
      const int n = u % TN, ko = u / TN;     // consecutive lanes: consecutive n
      vb[i] = ((c_lo * BK + 8 * ko) * p.ldb + min(n0 + n, N - 1)) * 4;
      sb[i] = ko * PS + (TM + n) * 16;
      kb[i] = 8 * ko;
    }
  }
  const int a_step = BK * 4, b_step = (BMODE == B_KC ? BK : BK * p.ldb) * 4;   // bytes per chunk

  f32x4 ring_a[PF][A_PER][2];
  f32x4 ring_b[PF][B_PER][2];
  float ring_m[PF][B_PER][8];    // B_MC: the eight dwords of a unit, k order (each its own request: scalars)
  constexpr int NLOAD = 2 * A_PER + BREQ * B_PER;
  static_assert((PF - 1) * NLOAD <= 63, "vmcnt is a 6-bit counter");

  // request i of chunk c_ (A requests first); issued UNCONDITIONALLY (hand-counted vmcnt): beyond the last chunk the
  // scalar offset points past the operand and the request returns zeros at once
  auto load_unit = [&](int c_, auto i_, f32x4 (&ra)[A_PER][2], f32x4 (&rb)[B_PER][2], float (&rm)[B_PER][8]) {
    constexpr int i = decltype(i_)::value;
    (void)va; (void)vb; (void)rsA; (void)rsB; (void)ldb4;
    const bool live = c_ < nchunks;
    if constexpr (i < 2 * A_PER) {
      const int so = live ? c_ * a_step : rsA[2];
      if constexpr (i % 2 == 0)
        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(ra[i / 2][0]) : "v"(va[i / 2]), "s"(rsA), "s"(so) : "memory");
      else
        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:16" : "=v"(ra[i / 2][1]) : "v"(va[i / 2]), "s"(rsA), "s"(so) : "memory");
    } else if constexpr (BMODE == B_KC) {
      constexpr int j = i - 2 * A_PER;
      const int so = live ? c_ * b_step : rsB[2];
      if constexpr (j % 2 == 0)
        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(rb[j / 2][0]) : "v"(vb[j / 2]), "s"(rsB), "s"(so) : "memory");
      else
        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:16" : "=v"(rb[j / 2][1]) : "v"(vb[j / 2]), "s"(rsB), "s"(so) : "memory");
    } else {
      constexpr int j = i - 2 * A_PER, u = j / 8, e = j % 8;
      const int so = live ? c_ * b_step + e * ldb4 : rsB[2];    // row k + e of the weight
      asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(rm[u][e]) : "v"(vb[u]), "s"(rsB), "s"(so) : "memory");
    }
  };
  auto wait_chunk = [&](auto newer, f32x4 (&ra)[A_PER][2], f32x4 (&rb)[B_PER][2], float (&rm)[B_PER][8]) {
    asm volatile("s_waitcnt vmcnt(%0)" : : "i"(decltype(newer)::value) : "memory");
#pragma unroll
    for (int i = 0; i < A_PER; ++i) { asm volatile("" : "+v"(ra[i][0])); asm volatile("" : "+v"(ra[i][1])); }
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
      if (BMODE == B_KC) { asm volatile("" : "+v"(rb[i][0])); asm volatile("" : "+v"(rb[i][1])); }
      else {
#pragma unroll
        for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(rm[i][e]));
      }
    }
  };
  // split unit i of chunk c (A units first) into its three planes and store them: 44 VALU + 3 ds_write_b128
  auto store_unit = [&](char *__restrict__ st, int c, auto i_, const f32x4 (&ra)[A_PER][2], const f32x4 (&rb)[B_PER][2],
                        const float (&rm)[B_PER][8]) {
    constexpr int i = decltype(i_)::value;
    if (GEMMX6_KO & 2) return;
    f32x4 lo, hi;
    int off, k0;
    if constexpr (i < A_PER) {
      if (A_UNITS % NT != 0 && tid + NT * i >= A_UNITS) return;
      lo = ra[i][0]; hi = ra[i][1]; off = sa[i]; k0 = ka[i];
    } else {
      constexpr int u = i - A_PER;
      if (B_UNITS % NT != 0 && tid + NT * u >= B_UNITS) return;
      if constexpr (BMODE == B_KC) { lo = rb[u][0]; hi = rb[u][1]; }
      else { lo = f32x4{rm[u][0], rm[u][1], rm[u][2], rm[u][3]}; hi = f32x4{rm[u][4], rm[u][5], rm[u][6], rm[u][7]}; }
      off = sb[u]; k0 = kb[u];
    }
    if (KEDGE && (BMODE == B_KC || i < A_PER)) {   // (n-contiguous weights: rows beyond K are out of the buffer: zeros)
      const int k = (c_lo + c) * BK + k0;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        lo[e] = k + e < K ? lo[e] : 0.f;
        hi[e] = k + 4 + e < K ? hi[e] : 0.f;
      }
    }
    unsigned a1[4], a2[4], a3[4];
    split_pair(lo[0], lo[1], a1[0], a2[0], a3[0]);
    split_pair(lo[2], lo[3], a1[1], a2[1], a3[1]);
    split_pair(hi[0], hi[1], a1[2], a2[2], a3[2]);
    split_pair(hi[2], hi[3], a1[3], a2[3], a3[3]);
    const u32x4 t1 = {a1[0], a1[1], a1[2], a1[3]}, t2 = {a2[0], a2[1], a2[2], a2[3]}, t3 = {a3[0], a3[1], a3[2], a3[3]};
    *reinterpret_cast<u32x4 *>(st + off) = t1;
    *reinterpret_cast<u32x4 *>(st + PLANE + off) = t2;
    *reinterpret_cast<u32x4 *>(st + 2 * PLANE + off) = t3;
  };

  f32x16 acc[MB][NB];
#pragma unroll
  for (int a = 0; a < MB; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  // operand reads of this lane: row (lane & 31) of a block, k octet 2 ks + (lane >> 5) in K-step ks
  const int lr = lane & 31, lg = lane >> 5;
  const int offA = lg * PS + (wm * 32 * MB + lr) * 16, offB = lg * PS + (TM + wn * 32 * NB + lr) * 16;
  bf16x8 fa[2][3][MB], fb[2][3][NB];   // [K-step][plane][block]
  if (GEMMX6_KO & 4) {
    for (int h = 0; h < 2; ++h)
      for (int pl = 0; pl < 3; ++pl) {
        for (int a = 0; a < MB; ++a) for (int e = 0; e < 8; ++e) fa[h][pl][a][e] = (__bf16)(float)(lane & 3);
        for (int b = 0; b < NB; ++b) for (int e = 0; e < 8; ++e) fb[h][pl][b][e] = (__bf16)1.f;
      }
  }
  constexpr int NF = 3 * (MB + NB), H = 6 * MB * NB;   // fragment reads / MFMAs per K-step
  auto read_frag = [&](const char *st, auto ks_, auto i_) {
    constexpr int ks = decltype(ks_)::value, i = decltype(i_)::value, pl = i / (MB + NB), q = i % (MB + NB);
    if (GEMMX6_KO & 4) return;
    if constexpr (q < MB) fa[ks][pl][q] = *reinterpret_cast<const bf16x8 *>(st + pl * PLANE + 2 * ks * PS + offA + q * 32 * 16);
    else fb[ks][pl][q - MB] = *reinterpret_cast<const bf16x8 *>(st + pl * PLANE + 2 * ks * PS + offB + (q - MB) * 32 * 16);
  };
  // MFMA m of K-step ks: term t = m / (MB NB) in the order a1 b3, a3 b1, a2 b2, a1 b2, a2 b1, a1 b1 (small first), then
  // the blocks: consecutive MFMAs go to different accumulators where there are several
  auto mfma_one = [&](auto ks_, auto m_) {
    constexpr int ks = decltype(ks_)::value, m = decltype(m_)::value;
    constexpr int t = m / (MB * NB), a = (m / NB) % MB, b = m % NB;
    constexpr int pa = t == 0 ? 0 : t == 1 ? 2 : t == 2 ? 1 : t == 3 ? 0 : t == 4 ? 1 : 0;
    constexpr int pb = t == 0 ? 2 : t == 1 ? 0 : t == 2 ? 1 : t == 3 ? 1 : t == 4 ? 0 : 0;
    if (GEMMX6_KO & 16) return;
    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][pa][a], fb[ks][pb][b], acc[a][b], 0, 0, 0);
  };
  typedef std::integral_constant<int, 0> I0;
  typedef std::integral_constant<int, 1> I1;
  constexpr int NST = A_PER + B_PER;     // store units per chunk

#ifdef GEMM16_TIMING
  int stamp_n = 0;
#define GEMMX6_STAMP() do { if (blockIdx.x == 0 && tid == 0 && stamp_n < 60) p.dbg[stamp_n] = __builtin_readcyclecounter(); ++stamp_n; } while (0)
#else
#define GEMMX6_STAMP() do { } while (0)
#endif
  GEMMX6_STAMP();
  // ---- prologue: PF chunks requested, chunks 0 and 1 staged, K-step 0 of chunk 0 in registers
  static_for<0, PF>([&](auto d_) {
    constexpr int d = decltype(d_)::value;
    static_for<0, NLOAD>([&](auto i_) { load_unit(d, i_, ring_a[d], ring_b[d], ring_m[d]); });
  });
  wait_chunk(std::integral_constant<int, (PF - 1) * NLOAD>(), ring_a[0], ring_b[0], ring_m[0]);
  static_for<0, NST>([&](auto i_) { store_unit(smem, 0, i_, ring_a[0], ring_b[0], ring_m[0]); });
  wait_chunk(std::integral_constant<int, (PF - 2) * NLOAD>(), ring_a[1], ring_b[1], ring_m[1]);
  static_for<0, NST>([&](auto i_) { store_unit(smem + STAGE, 1, i_, ring_a[1], ring_b[1], ring_m[1]); });
  __syncthreads();
  static_for<0, NF>([&](auto i_) { read_frag(smem, I0(), i_); });
  GEMMX6_STAMP();

  // Iteration c (as gemm16_kernel): stages c % 3 and (c + 1) % 3 hold chunks c and c + 1; ring slot c % PF takes the
  // requests of chunk c + PF; chunk c + 2 is split and stored.  K-step 0: its MFMAs, the fragment reads of K-step 1 and
  // the requests; K-step 1: its MFMAs, the split + stores of chunk c + 2 and the fragment reads of the next chunk.
  constexpr int NFILL0 = NF + NLOAD, NFILL1 = NST + NF;
  int s0 = 0;
  for (int c0 = 0; c0 < nchunks; c0 += PF) {
    static_for<0, PF>([&](auto d_) {
      constexpr int d = decltype(d_)::value, d2 = (d + 2) % PF;
      const int c = c0 + d;
      if (c >= nchunks) return;
      const int s1 = s0 == 2 ? 0 : s0 + 1, s2 = s1 == 2 ? 0 : s1 + 1;
      const char *st0 = smem + s0 * STAGE, *st1 = smem + s1 * STAGE;
      char *st2 = smem + s2 * STAGE;
      static_for<0, H>([&](auto m_) {
        constexpr int m = decltype(m_)::value;
        mfma_one(I0(), m_);
        static_for<m * NFILL0 / H, (m + 1) * NFILL0 / H>([&](auto k_) {
          constexpr int k = decltype(k_)::value;
          if constexpr (k < NF) read_frag(st0, I1(), k_);
          else load_unit(c + PF, std::integral_constant<int, k - NF>(), ring_a[d], ring_b[d], ring_m[d]);
        });
        __builtin_amdgcn_sched_barrier(0);
      });
      wait_chunk(std::integral_constant<int, (PF - 2) * NLOAD>(), ring_a[d2], ring_b[d2], ring_m[d2]);
      static_for<0, H>([&](auto m_) {
        constexpr int m = decltype(m_)::value;
        mfma_one(I1(), m_);
        static_for<m * NFILL1 / H, (m + 1) * NFILL1 / H>([&](auto k_) {
          constexpr int k = decltype(k_)::value;
          if constexpr (k < NST) store_unit(st2, c + 2, k_, ring_a[d2], ring_b[d2], ring_m[d2]);
          else read_frag(st1, I0(), std::integral_constant<int, k - NST>());
        });
        __builtin_amdgcn_sched_barrier(0);
      });
      if (!(GEMMX6_KO & 8)) __syncthreads();
      GEMMX6_STAMP();
      s0 = s1;
    });
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  GEMMX6_STAMP();

  // ---- epilogue.  C/D map of the 32x32 MFMA: column = lane & 31, row = 8 (reg >> 2) + 4 (lane >> 5) + (reg & 3).
  float *__restrict__ C = (z == 0 ? p.C : p.Cs + (size_t)(z - 1) * p.slab) + (size_t)batch * p.sC;
  const float *bias = (p.bias && z == 0) ? p.bias + (size_t)batch * p.sBias : nullptr;
  const float *addend = (p.addend && z == 0) ? p.addend + (size_t)batch * p.sC : nullptr;
  float *aux = p.aux ? p.aux + (size_t)batch * p.sC : nullptr;
  const int act = p.act;
#pragma unroll
  for (int a = 0; a < MB; ++a) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int col = n0 + (wn * NB + b) * 32 + lr;
      const bool col_ok = col < N;
      const float bv = (bias && col_ok) ? bias[col] : 0.f;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int row0 = m0 + (wm * MB + a) * 32 + 8 * g + 4 * lg;
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
          float v = acc[a][b][4 * g + r] + bv;
          if (act == 1) {
            if (aux) aux[(size_t)(row0 + r) * p.ldc + col] = v;
            v = gemm16::gelu(v);
          } else if (act == 2) {
            v *= gemm16::gelu_grad(xin[r]);
          }
          C[(size_t)(row0 + r) * p.ldc + col] = v + cin[r];
        }
      }
    }
  }
  GEMMX6_STAMP();
#undef GEMMX6_STAMP
}

template <int MB, int NB, int WGM, int WGN>
constexpr size_t lds_bytes() {
  return (size_t)3 * 3 * 4 * ((32 * MB * WGM + 32 * NB * WGN) * 16 + 16);
}

template <int MB, int NB, int WGM, int WGN, int PF>
hipError_t launch(Problem p, int bmode, hipStream_t stream) {
  constexpr int TM = 32 * MB * WGM, TN = 32 * NB * WGN;
  constexpr size_t lds = lds_bytes<MB, NB, WGM, WGN>();
  p.ntm = (p.M + TM - 1) / TM;
  p.ntn = (p.N + TN - 1) / TN;
  const unsigned grid = (unsigned)(p.ntm * p.ntn * p.splits * p.batch);
  if (grid == 0) return hipSuccess;
  const bool kedge = (p.K % BK) != 0;
#define GEMMX6_GO(BM, KE)                                                                                   \
  do {                                                                                                      \
    auto kern = gemmx6_kernel<MB, NB, WGM, WGN, PF, BM, KE>;                                                \
    static bool attr_done = false;                                                                          \
    if (!attr_done && lds > 64 * 1024) {                                                                    \
      hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,    \
                                         (int)lds);                                                         \
      if (e != hipSuccess) return e;                                                                        \
      attr_done = true;                                                                                     \
    }                                                                                                       \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WGM * WGN), lds, stream, p);                             \
  } while (0)
  if (bmode == B_KC) { if (kedge) GEMMX6_GO(B_KC, true); else GEMMX6_GO(B_KC, false); }
  else { if (kedge) GEMMX6_GO(B_MC, true); else GEMMX6_GO(B_MC, false); }
#undef GEMMX6_GO
  return hipGetLastError();
}

}  // namespace gemmx6
