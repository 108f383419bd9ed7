// gemmp_core.h -- f32 GEMM on the bf16 matrix cores over operands that ARRIVE split: three bf16 planes each (gfx950).
//
// The products of the Q-Former's dense layers (3DLLM_BLIP2-base/lavis/models/blip2_models/Qformer.py:116-118 query /
// key / value, :238 BertSelfOutput.dense, :305 BertIntermediate.dense + erf-GELU, :320 BertOutput.dense), of their
// input gradients (dX = dY W) and of their weight gradients (dW = dY^T X).
// Arithmetic as round 4's gemmx6_core.h (removed in round 6; git history): an f32 number is the exact sum of three bf16 numbers (x = x1 + x2 + x3); the six cross
// products with i + j <= 4, each exact in f32, are accumulated in f32 by v_mfma_f32_32x32x16_bf16, smallest first --
// f32-equivalent results (closer to the float64 product than an f32 GEMM's: DESIGN.md 4g).
// What round 4 measured (profiles/r04_gemmx6.md): with the split made INSIDE the product's loop the loop is bound by the
// VALU work of the split (1600 cycles per chunk against 768 of the matrix pipe), and at 416 rows the weight -- the big
// operand -- is re-split by every row tile of workgroups.  Here nobody splits inside a product: the PRODUCER of a
// matrix writes its planes once (AdamW for the weights, the LayerNorm tails / attention / GELU epilogues for the
// activations; planes.hip), and a product only moves 16-byte pieces of planes from memory to LDS.
//
// An operand is `planes` = [3][rows][cols] bf16 (plane stride given), in ONE of two orientations:
//   OP_K  rows are the operand's output index (m or n), cols the reduction index: what x and W of y = x W^T are;
//   OP_T  rows are the reduction index, cols the output index: W of dX = dY W, both operands of dW = dY^T X.
// Nothing is transposed in memory: an OP_K chunk sits in LDS as [k octet][row][8 k] and a fragment is ONE
// ds_read_b128 per plane, block and K-step (consecutive lanes at consecutive 16-byte slots: conflict-free); an OP_T
// chunk sits as [16-column subtile][reduction row][16 columns] (subtile stride 1152 bytes = 128 mod 256) and a fragment
// is two ds_read_b64_tr_b16 -- the LDS transposing read of gfx950 hands lane i of a 16-lane group column i of a
// [4 rows][16 columns] block (tools/micro/tr16_probe.hip prints it).  Stores: eight consecutive lanes fill 128
// consecutive bytes in both images (ds_write_b128 is served in groups of eight lanes over a 128-byte bank window).
//
// Tiles: at 416 rows a product is a K loop of one wave per 32 x 32 block and the chip has 1024 SIMDs; the tilings keep
// the row tile at 32 (416 = 13 x 32: no padding rows) and take as many 32-column blocks per workgroup as still give
// >= ~250 workgroups.  No split reductions, no slabs.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "gemm16_core.h"

#ifndef GEMMP_VARIANT
#define GEMMP_VARIANT 0   // measurement builds: 1 no scheduling barriers between the MFMA slots, 2 MFMA waves at raised priority
#endif
#ifndef GEMMP_KO
#define GEMMP_KO 0   // measurement builds: 1 no global loads, 2 no LDS stores, 4 no LDS reads, 8 no barrier, 16 no MFMAs (wrong results)
#endif

namespace gemmp {

using gemm16::static_for;
using gemm16::gelu;
using gemm16::gelu_grad;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 32;          // reduction indices per LDS chunk
constexpr int CCB = 32 * 64;     // bytes of one [32 reduction rows][32 columns] block of an OP_T chunk in LDS (one plane)
constexpr int ROWB = 192;       // bytes of one row of one 32-column chunk in memory: [3 planes][32 bf16]

enum { OP_K = 0, OP_T = 1 };

// Storage of a split matrix X (R rows, C columns, C % 32 == 0), "chunked planes": [C / 32][row capacity][3][32] bf16 --
// the three planes of 32 consecutive columns of a row are 192 consecutive bytes, rows follow each other, 32-column
// chunks are `cs` elements apart.  Whatever a product needs of X is then a run of whole 128-byte lines: the rows
// r0 .. r0 + T of chunk c (X's columns reduced over: OP_K) are T * 192 consecutive bytes, and 32 rows of the chunks
// c0 .. c0 + T / 32 (X's rows reduced over: OP_T) are T / 32 runs of 6144 bytes.  (Plain [3][R][C] planes gave 64-byte
// runs per row and request, half a line each: 14 bytes per clock and CU from L2, 25 with whole lines.)
struct Problem {
  const unsigned short *A;   // OP_K: X = (M, K); OP_T: X = (K, M)
  const unsigned short *B;   // OP_K: X = (N, K); OP_T: X = (K, N)
  long csA, csB;             // chunk strides in elements (row capacity * 96)
  long extA, extB;           // bytes that may be read from A / B of one batch element (requests beyond return zeros)
  float *C;                  // (M, N) f32 result, row stride ldc (or null)
  unsigned short *Cp;        // chunked planes of the result (or null), chunk stride csC
  long csC;
  const float *bias;         // [N] (or null)
  const float *addend;       // same layout as C (or null); may alias C
  float *aux;                // act 1: the pre-activation is kept here (or null); act 2: read from here
  float *ws;                 // splits > 1: partial tiles, splits * tiles * TM * TN floats
  unsigned *cnt;             // splits > 1: one arrival counter per tile, zero before the launch, zero after it
  int M, N, K;
  int ldc;
  long sA, sB, sC, sCp, sBias;   // batch strides in elements
  int batch, splits;
  int act;                   // 0 none, 1 erf-GELU, 2 times gelu'(aux)
  int ntm, ntn;
#ifdef GEMMP_TIMING
  unsigned long long *dbg;
#endif
};

// x -> three bf16 terms, x = t1 + t2 + t3 exactly (round to nearest even, subtract, repeat)
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned &t1, unsigned &t2, unsigned &t3) {
  auto pk = [](float a, float b) {
    const bf16x2 h = __builtin_convertvector(f32x2{a, b}, bf16x2);   // v_cvt_pk_bf16_f32
    return __builtin_bit_cast(unsigned, h);
  };
  t1 = pk(x0, x1);
  const float r0 = x0 - __builtin_bit_cast(float, t1 << 16), r1 = x1 - __builtin_bit_cast(float, t1 & 0xffff0000u);
  t2 = pk(r0, r1);
  const float s0 = r0 - __builtin_bit_cast(float, t2 << 16), s1 = r1 - __builtin_bit_cast(float, t2 & 0xffff0000u);
  t3 = pk(s0, s1);
}

// OP_K part: four k octets of [rows][16 bytes], 32 bytes of padding behind each
constexpr int octet_bytes(int t) { return t * 16 + 32; }
constexpr int part_bytes(int mode, int t) { return mode == OP_K ? 4 * octet_bytes(t) : (t / 32) * CCB; }
// a plane = A part + B part + a skew: the pieces eight adjacent lanes store (four octets of two planes of a row) then go
// to different 16-byte bank groups: + 16 bytes when only OP_K parts are there, + 64 with an OP_T part
constexpr int plane_bytes(int amode, int tm, int bmode, int tn) {
  return part_bytes(amode, tm) + part_bytes(bmode, tn) + ((amode == OP_T || bmode == OP_T) ? 64 : 16);
}

// MB x NB blocks of 32 x 32 per wave, WGM x WGN waves per workgroup, PF >= 3 chunks in flight (register ring),
// OCC workgroups per CU the register budget is sized for.
template <int MB, int NB, int WGM, int WGN, int PF, int OCC, int AMODE, int BMODE, bool KEDGE>
__global__ __launch_bounds__(64 * WGM * WGN, OCC) void gemmp_kernel(const Problem p) {
  static_assert(PF >= 3, "chunk c + 2 is stored while chunk c + PF is requested into chunk c's slot");
  constexpr int NW = WGM * WGN, NT = 64 * NW;
  constexpr int TM = 32 * MB * WGM, TN = 32 * NB * WGN;
  constexpr int PARTA = part_bytes(AMODE, TM);
  constexpr int PLANE = plane_bytes(AMODE, TM, BMODE, TN), STAGE = 3 * PLANE;
  constexpr int PSA = octet_bytes(TM), PSB = octet_bytes(TN);      // OP_K: bytes of one k octet of a part
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [3 stages][3 planes][A part | B part]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wv / WGN, wn = wv - wm * WGN;

  // ---- which tile: workgroups of one XCD (blockIdx % 8) take a contiguous run of work ids ordered (batch, n tile,
  // split, m tile) with the m tile fastest: neighbours share their weight tile in the XCD's L2
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

  const unsigned short *__restrict__ A = p.A + (size_t)batch * p.sA;
  const unsigned short *__restrict__ B = p.B + (size_t)batch * p.sB;
  const int M = p.M, N = p.N, K = p.K;
  const int nchunks_all = (K + BK - 1) / BK;
  const int c_lo = (int)((long)nchunks_all * z / p.splits), c_hi = (int)((long)nchunks_all * (z + 1) / p.splits);
  const int nchunks = c_hi - c_lo;

  // ---- global -> register ring -> LDS in 16-byte pieces; piece j of an operand's chunk is at byte 16 j of the run(s)
  // described above, so a wave's request is 1 KiB of consecutive memory.
  constexpr int A_UNITS = TM * 12, A_PER = (A_UNITS + NT - 1) / NT;
  constexpr int B_UNITS = TN * 12, B_PER = (B_UNITS + NT - 1) / NT;
  constexpr int NU = A_PER + B_PER;
  auto descriptor = [](const unsigned short *base_, size_t bytes) {
    const unsigned long long a = (unsigned long long)base_;
    i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));
    r[2] = __builtin_amdgcn_readfirstlane((int)(unsigned)bytes);
    r[3] = 0x00020000;
    return r;
  };
  const i32x4 rsA = descriptor(A, (size_t)p.extA);
  const i32x4 rsB = descriptor(B, (size_t)p.extB);
  int vo[NU];      // byte offset of the piece in chunk 0
  int so[NU];      // LDS byte offset inside a stage
  int ko[NU];      // OP_T: reduction row of the piece inside the chunk
  // The OP_T image of a plane is the memory image of that plane: [32-column block][reduction row][64 bytes]; the two
  // 16-lane groups of a transposing read take the two 32-byte halves of four rows = 256 consecutive bytes.
  auto place = [&](int mode, int j, int t, int t0, int extent, long cs, int lds_base, int &v, int &s, int &k) {
    const int rem = j % 12, pl = rem >> 2, q = rem & 3;
    if (mode == OP_K) {
      const int row = j / 12;
      v = min(t0 + row, extent - 1) * ROWB + rem * 16;
      s = pl * PLANE + lds_base + q * octet_bytes(t) + row * 16;
      k = 0;
    } else {
      const int ccl = j / 384, r = (j % 384) / 12;
      v = (int)(unsigned)((long)min(t0 / 32 + ccl, extent / 32 - 1) * cs * 2 + r * ROWB + rem * 16);   // < 4 GB
      s = pl * PLANE + lds_base + ccl * CCB + r * 64 + q * 16;
      k = r;
    }
  };
#pragma unroll
  for (int i = 0; i < A_PER; ++i)
    place(AMODE, min(tid + NT * i, A_UNITS - 1), TM, m0, M, p.csA, 0, vo[i], so[i], ko[i]);
#pragma unroll
  for (int i = 0; i < B_PER; ++i)
    place(BMODE, min(tid + NT * i, B_UNITS - 1), TN, n0, N, p.csB, PARTA, vo[A_PER + i], so[A_PER + i], ko[A_PER + i]);
  // byte offsets are UNSIGNED 32-bit numbers (an operand may be up to 4 GB: 320 000 point tokens x 1408 x 6 bytes)
  const unsigned a_step = AMODE == OP_K ? (unsigned)(p.csA * 2) : BK * ROWB, b_step = BMODE == OP_K ? (unsigned)(p.csB * 2) : BK * ROWB;
  const unsigned a_first = (unsigned)c_lo * a_step, b_first = (unsigned)c_lo * b_step;

  u32x4 ring[PF][NU];
  if (GEMMP_KO & 1) {
    for (int d = 0; d < PF; ++d) for (int u = 0; u < NU; ++u) ring[d][u] = u32x4{(unsigned)tid, 1u, 2u, 3u};
  }
  constexpr int NLOAD = NU;
  static_assert((PF - 1) * NLOAD <= 63, "vmcnt is a 6-bit counter");
  // the requests are hidden from hipcc: under register pressure it would park a ring register in an AGPR straight
  // after the asm statement that names it, i.e. copy a register whose load has not landed
  static_assert(PF * NLOAD * 4 <= 160, "ring too large: hipcc would move in-flight registers");

  // request i of chunk c_ (A pieces first); issued UNCONDITIONALLY (hand-counted vmcnt): beyond the last chunk the
  // scalar offset points past the operand and the request returns zeros at once (or, wrapping, reads something
  // inside it that nobody looks at)
  auto load_unit = [&](int c_, auto i_, u32x4 (&rg)[NU]) {
    constexpr int u = decltype(i_)::value;
    (void)vo; (void)rsA; (void)rsB;
    if (GEMMP_KO & 1) return;
    const bool live = c_ < nchunks;
    if constexpr (u < A_PER) {
      const unsigned s = live ? a_first + (unsigned)c_ * a_step : 0xffffff00u;
      asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(rg[u]) : "v"(vo[u]), "s"(rsA), "s"(s) : "memory");
    } else {
      const unsigned s = live ? b_first + (unsigned)c_ * b_step : 0xffffff00u;
      asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(rg[u]) : "v"(vo[u]), "s"(rsB), "s"(s) : "memory");
    }
  };
  auto wait_chunk = [&](auto newer, u32x4 (&rg)[NU]) {
    if (GEMMP_KO & 1) return;
    asm volatile("s_waitcnt vmcnt(%0)" : : "i"(decltype(newer)::value) : "memory");
#pragma unroll
    for (int u = 0; u < NU; ++u) asm volatile("" : "+v"(rg[u]));
  };
  // piece u of chunk c to LDS: one ds_write_b128, no arithmetic
  auto store_unit = [&](char *__restrict__ st, int c, auto u_, const u32x4 (&rg)[NU]) {
    constexpr int u = decltype(u_)::value;
    if (GEMMP_KO & 2) return;
    if constexpr (u < A_PER) {
      if (A_UNITS % NT != 0 && tid + NT * u >= A_UNITS) return;
    } else {
      if (B_UNITS % NT != 0 && tid + NT * (u - A_PER) >= B_UNITS) return;
    }
    constexpr bool op_t = (u < A_PER ? AMODE : BMODE) == OP_T;
    const bool zero = KEDGE && op_t && (c_lo + c) * BK + ko[u] >= K;    // reduction rows beyond K (OP_K: K % 32 == 0)
    *reinterpret_cast<u32x4 *>(st + so[u]) = zero ? u32x4{0u, 0u, 0u, 0u} : rg[u];
  };

  f32x16 acc[MB][NB];
#pragma unroll
  for (int a = 0; a < MB; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  // operand reads of this lane.  The MFMA wants row (lane & 31) of a block with the 8 reduction indices 8 (lane >> 5) ..
  // of a K-step.  OP_K: one 16-byte slot.  OP_T: lane i of a 16-lane group supplies the address of row (i >> 2), bytes
  // 8 (i & 3) .. of a [4 rows][16 columns] block and receives column i: group (lane >> 4) & 1 picks the 16-column
  // subtile, (lane >> 5) the reduction octet; the second read is four rows (128 bytes) further.
  const int lr = lane & 31, lg = lane >> 5, li = lane & 15, lh = (lane >> 4) & 1;
  const int offA = AMODE == OP_K ? lg * PSA + (wm * 32 * MB + lr) * 16
                                 : wm * MB * CCB + (8 * lg + (li >> 2)) * 64 + lh * 32 + (li & 3) * 8;
  const int offB = PARTA + (BMODE == OP_K ? lg * PSB + (wn * 32 * NB + lr) * 16
                                          : wn * NB * CCB + (8 * lg + (li >> 2)) * 64 + lh * 32 + (li & 3) * 8);
  bf16x8 fa[2][3][MB], fb[2][3][NB];   // [K-step][plane][block]
  if (GEMMP_KO & (4 | 1)) {
    for (int h = 0; h < 2; ++h)
      for (int pl = 0; pl < 3; ++pl) {
        for (int a = 0; a < MB; ++a) for (int e = 0; e < 8; ++e) fa[h][pl][a][e] = (__bf16)(float)(lane & 3);
        for (int b = 0; b < NB; ++b) for (int e = 0; e < 8; ++e) fb[h][pl][b][e] = (__bf16)1.f;
      }
  }
  constexpr int NF = 3 * (MB + NB), H = 6 * MB * NB;   // fragments / MFMAs per K-step
  auto read_one = [&](const char *st, int mode, int off, int ks, int blk, int ps) -> bf16x8 {
    if (mode == OP_K) return *reinterpret_cast<const bf16x8 *>(st + off + 2 * ks * ps + blk * 32 * 16);
    const char *q = st + off + blk * CCB + ks * 16 * 64;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(q));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(q + 256));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
  };
  auto read_frag = [&](const char *st, auto ks_, auto i_) {
    constexpr int ks = decltype(ks_)::value, i = decltype(i_)::value, pl = i / (MB + NB), q = i % (MB + NB);
    if (GEMMP_KO & 4) return;
    if constexpr (q < MB) fa[ks][pl][q] = read_one(st + pl * PLANE, AMODE, offA, ks, q, PSA);
    else fb[ks][pl][q - MB] = read_one(st + pl * PLANE, BMODE, offB, ks, q - MB, PSB);
  };
  // MFMA m of K-step ks: term t = m / (MB NB) in the order a1 b3, a3 b1, a2 b2, a1 b2, a2 b1, a1 b1 (small first), then
  // the blocks: consecutive MFMAs go to different accumulators where there are several
  auto mfma_one = [&](auto ks_, auto m_) {
    constexpr int ks = decltype(ks_)::value, m = decltype(m_)::value;
    constexpr int t = m / (MB * NB), a = (m / NB) % MB, b = m % NB;
    constexpr int pa = t == 0 ? 0 : t == 1 ? 2 : t == 2 ? 1 : t == 3 ? 0 : t == 4 ? 1 : 0;
    constexpr int pb = t == 0 ? 2 : t == 1 ? 0 : t == 2 ? 1 : t == 3 ? 1 : t == 4 ? 0 : 0;
    if (GEMMP_KO & 16) return;
    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][pa][a], fb[ks][pb][b], acc[a][b], 0, 0, 0);
  };
  typedef std::integral_constant<int, 0> I0;
  typedef std::integral_constant<int, 1> I1;

#ifdef GEMMP_TIMING
  int stamp_n = 0;
#define GEMMP_STAMP() do { if (blockIdx.x == 0 && tid == 0 && stamp_n < 60) p.dbg[stamp_n] = __builtin_readcyclecounter(); ++stamp_n; } while (0)
#else
#define GEMMP_STAMP() do { } while (0)
#endif
  GEMMP_STAMP();
  if (GEMMP_VARIANT & 2) { if (wv & 1) __builtin_amdgcn_s_setprio(1); }
  // ---- prologue: PF chunks requested, chunks 0 and 1 staged, K-step 0 of chunk 0 in registers
  static_for<0, PF>([&](auto d_) {
    constexpr int d = decltype(d_)::value;
    static_for<0, NLOAD>([&](auto i_) { load_unit(d, i_, ring[d]); });
  });
  wait_chunk(std::integral_constant<int, (PF - 1) * NLOAD>(), ring[0]);
  static_for<0, NU>([&](auto u_) { store_unit(smem, 0, u_, ring[0]); });
  wait_chunk(std::integral_constant<int, (PF - 2) * NLOAD>(), ring[1]);
  static_for<0, NU>([&](auto u_) { store_unit(smem + STAGE, 1, u_, ring[1]); });
  __syncthreads();
  static_for<0, NF>([&](auto i_) { read_frag(smem, I0(), i_); });
  GEMMP_STAMP();

  // Iteration c: stages c % 3 and (c + 1) % 3 hold chunks c and c + 1; ring slot c % PF takes the requests of chunk
  // c + PF; chunk c + 2 (requested PF - 2 iterations ago) goes to LDS.  The LDS store path takes 13 cycles per
  // ds_write_b128 and wave and serves the CU's waves one after the other -- a chunk's stores are as long as its MFMAs
  // -- and LDS operations complete in order, so a fragment read queues behind every store issued before it: the stores
  // are SPREAD over both K-steps (they were bunched in the second one, and the reads of the next chunk's first K-step
  // waited ~330 cycles behind them: knock-out runs of tools/micro/gemmp_bench.hip).
  // K-step 0: its MFMAs | the fragment reads of K-step 1, half the stores, the requests;
  // K-step 1: its MFMAs | the other stores, the fragment reads of the next chunk's K-step 0.
  constexpr int NS0 = NU / 2, NS1 = NU - NS0;
  constexpr int NFILL0 = NF + NS0 + NLOAD, NFILL1 = NS1 + NF;
  int s0 = 0;
  for (int c0 = 0; c0 < nchunks; c0 += PF) {
    static_for<0, PF>([&](auto d_) {
      constexpr int d = decltype(d_)::value, d2 = (d + 2) % PF;
      const int c = c0 + d;
      if (c >= nchunks) return;
      const int s1 = s0 == 2 ? 0 : s0 + 1, s2 = s1 == 2 ? 0 : s1 + 1;
      const char *st0 = smem + s0 * STAGE, *st1 = smem + s1 * STAGE;
      char *st2 = smem + s2 * STAGE;
      wait_chunk(std::integral_constant<int, (PF - 3) * NLOAD>(), ring[d2]);
      static_for<0, H>([&](auto m_) {
        constexpr int m = decltype(m_)::value;
        mfma_one(I0(), m_);
        static_for<m * NFILL0 / H, (m + 1) * NFILL0 / H>([&](auto k_) {
          constexpr int k = decltype(k_)::value;
          if constexpr (k < NF) read_frag(st0, I1(), k_);
          else if constexpr (k < NF + NS0) store_unit(st2, c + 2, std::integral_constant<int, k - NF>(), ring[d2]);
          else load_unit(c + PF, std::integral_constant<int, k - NF - NS0>(), ring[d]);
        });
        if (!(GEMMP_VARIANT & 1)) __builtin_amdgcn_sched_barrier(0);
      });
      static_for<0, H>([&](auto m_) {
        constexpr int m = decltype(m_)::value;
        mfma_one(I1(), m_);
        static_for<m * NFILL1 / H, (m + 1) * NFILL1 / H>([&](auto k_) {
          constexpr int k = decltype(k_)::value;
          if constexpr (k < NS1) store_unit(st2, c + 2, std::integral_constant<int, NS0 + k>(), ring[d2]);
          else read_frag(st1, I0(), std::integral_constant<int, k - NS1>());
        });
        if (!(GEMMP_VARIANT & 1)) __builtin_amdgcn_sched_barrier(0);
      });
      if (!(GEMMP_KO & 8)) __syncthreads();
      GEMMP_STAMP();
      s0 = s1;
    });
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  GEMMP_STAMP();

  // ---- split reduction: every split parks its tile in the work space and counts itself in; the LAST one to arrive adds
  // the others to its registers and runs the epilogue (two splits: a + b in either order is the same number; more: all
  // partial tiles are read back and added in split order, the result does not depend on who was last).  The tiles
  // travel as write-through (sc1) stores and sc1 loads: the XCDs' L2s are not coherent with each other for plain
  // accesses, and fences at agent scope write back / invalidate whole caches (measured: 4-5 x the time of the product).
  constexpr int EPW = 32 * 36 * 4;   // bytes of a wave's epilogue tile
  if (p.splits > 1) {
    const int tile = (batch * p.ntn + tn) * p.ntm + tm;
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(p.ws, 0, 0x7fffffff, 0x00020000);
    const int slab = TM * TN * 4;
    const int mine = (tile * p.splits + z) * slab;
#pragma unroll
    for (int a = 0; a < MB; ++a)
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const u32x4 v = __builtin_bit_cast(u32x4, f32x4{acc[a][b][4 * g], acc[a][b][4 * g + 1], acc[a][b][4 * g + 2], acc[a][b][4 * g + 3]});
          __builtin_amdgcn_raw_buffer_store_b128(v, rsW, (((a * NB + b) * 4 + g) * NT + tid) * 16, mine, 16);
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned *flag = reinterpret_cast<unsigned *>(smem + NW * EPW);
    if (tid == 0) *flag = __hip_atomic_fetch_add(p.cnt + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const unsigned arrived = __builtin_amdgcn_readfirstlane(*flag);
    if (arrived != (unsigned)(p.splits - 1)) return;
    if (tid == 0) __hip_atomic_store(p.cnt + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next launch
    if (p.splits > 2) {
#pragma unroll
      for (int a = 0; a < MB; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
    }
    for (int zz = 0; zz < p.splits; ++zz) {
      if (p.splits == 2 && zz == z) continue;
      const int theirs = (tile * p.splits + zz) * slab;
#pragma unroll
      for (int a = 0; a < MB; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsW, (((a * NB + b) * 4 + g) * NT + tid) * 16, theirs, 16));
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[a][b][4 * g + e] += v[e];
          }
    }
  }

  // ---- epilogue.  C/D map of the 32x32 MFMA: column = lane & 31, row = 8 (reg >> 2) + 4 (lane >> 5) + (reg & 3).  A
  // block goes through a wave-private LDS tile so that a lane ends up with 8 consecutive columns of a row: 16-byte
  // loads of bias / addend / aux, 16-byte stores of C, aux and of the three planes of the result.
  constexpr int EPS = 36;   // floats per row of the tile
  float *ep = reinterpret_cast<float *>(smem + wv * EPW);
  float *__restrict__ C = p.C ? p.C + (size_t)batch * p.sC : nullptr;
  unsigned short *__restrict__ Cp = p.Cp ? p.Cp + (size_t)batch * p.sCp : nullptr;
  const float *bias = p.bias ? p.bias + (size_t)batch * p.sBias : nullptr;
  const float *addend = p.addend ? p.addend + (size_t)batch * p.sC : nullptr;
  float *aux = p.aux ? p.aux + (size_t)batch * p.sC : nullptr;
  const int act = p.act;
  const int er = lane >> 2, ec = (lane & 3) * 8;
#pragma unroll
  for (int a = 0; a < MB; ++a) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
#pragma unroll
      for (int r = 0; r < 16; ++r) ep[(8 * (r >> 2) + 4 * lg + (r & 3)) * EPS + lr] = acc[a][b][r];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const int col = n0 + (wn * NB + b) * 32 + ec;
      const bool col_ok = col < N;      // N % 8 == 0: eight columns are inside or outside together
      f32x4 bv[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
      if (bias && col_ok) {
        bv[0] = *reinterpret_cast<const f32x4 *>(bias + col);
        bv[1] = *reinterpret_cast<const f32x4 *>(bias + col + 4);
      }
#pragma unroll
      for (int ps = 0; ps < 2; ++ps) {
        const int row = m0 + (wm * MB + a) * 32 + 16 * ps + er;
        f32x4 v[2];
        v[0] = *reinterpret_cast<const f32x4 *>(ep + (16 * ps + er) * EPS + ec);
        v[1] = *reinterpret_cast<const f32x4 *>(ep + (16 * ps + er) * EPS + ec + 4);
        if (!(col_ok && row < M)) continue;
        const size_t o = (size_t)row * p.ldc + col;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          f32x4 x = v[h] + bv[h];
          if (act == 1) {
            if (aux) *reinterpret_cast<f32x4 *>(aux + o + 4 * h) = x;
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = gelu(x[e]);
          } else if (act == 2) {
            const f32x4 u = *reinterpret_cast<const f32x4 *>(aux + o + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] *= gelu_grad(u[e]);
          }
          if (addend) x += *reinterpret_cast<const f32x4 *>(addend + o + 4 * h);
          v[h] = x;
        }
        if (C) {
          *reinterpret_cast<f32x4 *>(C + o) = v[0];
          *reinterpret_cast<f32x4 *>(C + o + 4) = v[1];
        }
        if (Cp) {   // chunked planes of the result: 16 bytes per plane, the three 64 bytes apart
          unsigned t1[4], t2[4], t3[4];
          split_pair(v[0][0], v[0][1], t1[0], t2[0], t3[0]);
          split_pair(v[0][2], v[0][3], t1[1], t2[1], t3[1]);
          split_pair(v[1][0], v[1][1], t1[2], t2[2], t3[2]);
          split_pair(v[1][2], v[1][3], t1[3], t2[3], t3[3]);
          unsigned short *q = Cp + (size_t)(col >> 5) * p.csC + (size_t)row * 96 + (col & 31);
          *reinterpret_cast<u32x4 *>(q) = u32x4{t1[0], t1[1], t1[2], t1[3]};
          *reinterpret_cast<u32x4 *>(q + 32) = u32x4{t2[0], t2[1], t2[2], t2[3]};
          *reinterpret_cast<u32x4 *>(q + 64) = u32x4{t3[0], t3[1], t3[2], t3[3]};
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
  GEMMP_STAMP();
#undef GEMMP_STAMP
}

template <int MB, int NB, int WGM, int WGN, int AMODE, int BMODE>
constexpr size_t lds_bytes() {
  return (size_t)3 * 3 * plane_bytes(AMODE, 32 * MB * WGM, BMODE, 32 * NB * WGN);
}

template <int MB, int NB, int WGM, int WGN, int PF, int OCC, int AMODE, int BMODE>
hipError_t launch_modes(Problem p, hipStream_t stream) {
  constexpr int TM = 32 * MB * WGM, TN = 32 * NB * WGN;
  constexpr size_t lds = lds_bytes<MB, NB, WGM, WGN, AMODE, BMODE>();
  static_assert(lds >= (size_t)WGM * WGN * 32 * 36 * 4 + 16, "the epilogue tiles live in the stage memory");
  p.ntm = (p.M + TM - 1) / TM;
  p.ntn = (p.N + TN - 1) / TN;
  if (p.splits < 1) p.splits = 1;
  const unsigned grid = (unsigned)(p.ntm * p.ntn * p.batch * p.splits);
  if (grid == 0) return hipSuccess;
  const bool kedge = (p.K % BK) != 0;
#define GEMMP_GO(KE)                                                                                        \
  do {                                                                                                      \
    auto kern = gemmp_kernel<MB, NB, WGM, WGN, PF, OCC, AMODE, BMODE, KE>;                                  \
    static gemm16::OncePerDevice attr_done;                                                                 \
    if (lds > 64 * 1024 && attr_done.pending()) {                                                           \
      hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,    \
                                         (int)lds);                                                         \
      if (e != hipSuccess) return e;                                                                        \
      attr_done.done();                                                                                     \
    }                                                                                                       \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WGM * WGN), lds, stream, p);                             \
  } while (0)
  if (kedge) GEMMP_GO(true); else GEMMP_GO(false);
#undef GEMMP_GO
  return hipGetLastError();
}

// modes: 0 = (OP_K, OP_K) forward, 1 = (OP_K, OP_T) input gradient, 2 = (OP_T, OP_T) weight gradient
template <int MB, int NB, int WGM, int WGN, int PF, int OCC>
hipError_t launch(const Problem &p, int modes, hipStream_t stream) {
  switch (modes) {
    case 0: return launch_modes<MB, NB, WGM, WGN, PF, OCC, OP_K, OP_K>(p, stream);
    case 1: return launch_modes<MB, NB, WGM, WGN, PF, OCC, OP_K, OP_T>(p, stream);
    default: return launch_modes<MB, NB, WGM, WGN, PF, OCC, OP_T, OP_T>(p, stream);
  }
}

}  // namespace gemmp
