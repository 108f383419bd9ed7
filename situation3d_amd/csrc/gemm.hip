// gemm.hip -- exact-f32 MFMA GEMM family for the Q-Former's dense layers on gfx950.
//
// Replaces the library (rocBLAS / hipBLASLt Tensile) GEMMs behind the nn.Linear layers of
//   3DLLM_BLIP2-base/lavis/models/blip2_models/Qformer.py:116-118 (query / key / value),
//   :238 (BertSelfOutput.dense), :305 (BertIntermediate.dense), :320 (BertOutput.dense)
// and their backward products, for row counts of a few hundred (B x (32 queries + 20 question
// tokens)) where a step was ~230 small library launches.  One kernel family with what the layers need:
//   forward  C = A W^T (+ bias) (-> erf-GELU, pre-activation kept for the backward pass)
//   dX       C = dY W  (+ C: the residual path's gradient, beta = 1) (* gelu'(pre))
//   dW       C = dY^T X, with the bias gradient (column sums of dY) taken from the A operand on the way
//   ragged batches: the query branch and the text branch of a layer (different row counts, different
//            weights) as ONE launch without padding rows
//   groups : up to four independent products (e.g. the dX and the dW product of one layer, which share
//            only their dY operand) as ONE launch -- at these sizes ~6 us of every launch is fixed cost
//            (launch, first HBM-cold weight tile, drain), measured with tools/micro/gemm_variants.hip
// so that bias_gelu, accumulate, column_sum and half of the GEMM launches disappear.
//
// Arithmetic: v_mfma_f32_32x32x2_f32 -- exact f32 products, f32 accumulation (bitwise an fmaf chain per
// k-slice; the north star's 1e-4 bar rules out bf16 operands).  Peak 157 TFLOP/s (MI355X_MICROARCH.md);
// a pure stream of these MFMAs sustains 135-144 TFLOP/s at the clock this part holds under that load.
//
// Structure: a workgroup of 4 waves owns a tile of C (64x64, 32x128, 64x128 or 128x64) and walks K in
// chunks of 32 through a double-buffered LDS stage (global -> registers one chunk ahead, registers -> LDS
// after the MFMAs of the current chunk; one barrier per chunk).  LDS tiles are [k][m] with m contiguous, so
// the MFMA operand reads (lane = m, half-wave = k parity) are conflict-free ds_read_b32, issued one group of
// four k-steps ahead of the MFMAs that use them (sched_group_barrier: the compiler otherwise sinks every
// read next to its use and the matrix pipe idles for an LDS round trip per step).  Operands whose global
// layout is k-contiguous (activations, nn.Linear weights in the forward product) are turned on the way in
// with an odd row stride; operands that are m-contiguous (weights in dX, both operands of dW) are copied
// with 16-byte stores.  Row counts are small, so K is split over several workgroups when the tile grid
// alone cannot fill 256 CUs x 4 SIMDs: partial tiles then meet through float atomics in L2 (the destination
// holds the addend: zeros, or the beta = 1 term).  Tiles are dealt so that the workgroups of one XCD
// (blockIdx % 8) share weight columns: every weight tile is fetched from HBM by ONE XCD's L2.
#include "sig3d_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int GK = 32;       // K chunk per LDS stage
constexpr int MAX_GROUP = 4;

enum { MODE_KC = 0, MODE_MC = 1 };  // operand element (m, k) at  p[m*ld + k]  /  p[k*ld + m]

struct GemmArgs {
  const float *A, *B;
  float *C;
  const float *bias;   // [N] per output column (or null)
  float *aux;          // act 1: pre-activation out (or null); act 2: pre-activation in
  float *rowsum;       // [M]: sum over k of A(m, k) (bias gradient of a dW product), or null
  int amode, bmode;
  int M, N, K, m_last, k_last, batch;
  int lda, ldb, ldc;
  long sA, sB, sC, sBias, sRowsum;  // batch strides in elements
  int ksplit, accumulate, act, vec;
  int ntm, ntn;
};

struct GemmGroup {
  GemmArgs p[MAX_GROUP];
  int nprob;
  int start[MAX_GROUP + 1];  // first workgroup of every problem (prefix sums of ntm * ntn * batch * ksplit)
};

__device__ __forceinline__ float gemm_gelu(float u) { return 0.5f * u * (1.f + erff(u * 0.70710678118654752440f)); }
__device__ __forceinline__ float gemm_gelu_grad(float u) {
  const float cdf = 0.5f * (1.f + erff(u * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * u * u);
  return cdf + u * pdf;
}

// One operand tile (TT values of m  x  GK values of k): global -> 4-float register groups -> LDS [k][m].
// KC: unit u -> row m = u / 8, k quad = u % 8;   MC: unit u -> k row = u / (TT/4), m quad = u % (TT/4).
// load() only ISSUES the (address-clamped, unconditional) loads: nothing may consume r[] before the MFMAs of
// the current chunk have been issued, or the compiler parks the wave on s_waitcnt right behind the load and
// the HBM latency of every chunk is exposed.  Out-of-range elements are zeroed in mask().
template <int TT>
struct OperandTile {
  static constexpr int UNITS = TT * GK / 4;          // float4 units per chunk: 256 (TT = 32), 512, 1024
  static constexpr int PER = UNITS / 256;            // per thread
  f32x4 r[PER];

  __device__ __forceinline__ static void unit(int mode, int u, int &m, int &k) {
    if (mode == MODE_KC) { m = u / (GK / 4); k = 4 * (u % (GK / 4)); }
    else { k = u / (TT / 4); m = 4 * (u % (TT / 4)); }
  }
  // LDS tile [k][LD] with LD = TT + 4 (16-byte aligned rows for the m-contiguous copy) and the column XOR-ed
  // with 4 * ((k >> 3) & 3): the MFMA operand reads stay 32 consecutive words per half-wave (a permutation
  // inside a 16-word group), the 16-byte copies stay whole, and the transposing 4-byte stores of a k-contiguous
  // operand (8 lanes per row, 4 rows per half-wave) hit 32 distinct banks instead of 8.
  static constexpr int LD = TT + 4;
  __device__ __forceinline__ static int swz(int k) { return ((k >> 3) & 3) << 2; }

  __device__ __forceinline__ void load(int mode, bool vec, const float *__restrict__ p, int ldg, int m0, int k0,
                                       int mdim, int kdim, int tid) {
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      int m, k;
      unit(mode, tid + 256 * q, m, k);
      m += m0; k += k0;
      if (vec) {
        // the contiguous dimension is a multiple of 4 and 16-byte aligned: a quad is all in or all out
        const bool in = (m < mdim) && (k < kdim);
        const int mc = in ? m : 0, kc = in ? k : 0;
        const float *src = (mode == MODE_KC) ? p + (size_t)mc * ldg + kc : p + (size_t)kc * ldg + mc;
        r[q] = *reinterpret_cast<const f32x4 *>(src);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int me = (mode == MODE_KC) ? m : m + e, ke = (mode == MODE_KC) ? k + e : k;
          const bool in = (me < mdim) && (ke < kdim);
          const int mc = in ? me : 0, kc = in ? ke : 0;
          r[q][e] = (mode == MODE_KC) ? p[(size_t)mc * ldg + kc] : p[(size_t)kc * ldg + mc];
        }
      }
    }
  }

  __device__ __forceinline__ void mask(int mode, int m0, int k0, int mdim, int kdim, int tid) {
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      int m, k;
      unit(mode, tid + 256 * q, m, k);
      m += m0; k += k0;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int me = (mode == MODE_KC) ? m : m + e, ke = (mode == MODE_KC) ? k + e : k;
        if (!((me < mdim) && (ke < kdim))) r[q][e] = 0.f;
      }
    }
  }

  __device__ __forceinline__ void store(int mode, float *__restrict__ s, int tid) const {
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      int m, k;
      unit(mode, tid + 256 * q, m, k);
      if (mode == MODE_KC) {
        const int mm = m ^ swz(k);      // k .. k+3 share (k >> 3)
#pragma unroll
        for (int e = 0; e < 4; ++e) s[(k + e) * LD + mm] = r[q][e];
      } else {
        *reinterpret_cast<f32x4 *>(s + k * LD + (m ^ swz(k))) = r[q];
      }
    }
  }
};

// WGM x WGN waves (product 4), each wave a (32 WM) x (32 WN) block of 32x32 MFMA tiles
template <int WGM, int WGN, int WM, int WN>
__global__ __launch_bounds__(256) void qf_gemm_kernel(GemmGroup grp) {
  constexpr int TM = 32 * WM * WGM, TN = 32 * WN * WGN;
  typedef OperandTile<TM> TileA;
  typedef OperandTile<TN> TileB;
  constexpr int LDA = TileA::LD, LDB = TileB::LD;
  extern __shared__ __attribute__((aligned(16))) float gemm_smem[];
  float *const s_a = gemm_smem;                          // [2][GK][LDA]
  float *const s_b = gemm_smem + 2 * GK * LDA;        // [2][GK][LDB]
  float *const s_rowsum = gemm_smem + 2 * GK * (LDA + LDB);

  const int tid = threadIdx.x;
  const int lane = tid & 63, l31 = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave - wm * WGN;

  // ---- which problem, which tile.  Within a problem XCD x = (workgroup index) % 8 owns a contiguous run of
  // tile ids, ids ordered n-major so that the m-tiles of one weight column block sit on one XCD.
  int pi = 0;
#pragma unroll
  for (int i = 1; i < MAX_GROUP; ++i)
    if (i < grp.nprob && (int)blockIdx.x >= grp.start[i]) pi = i;
  const GemmArgs &g = grp.p[pi];
  const int wg = blockIdx.x - grp.start[pi];
  const int T = g.ntm * g.ntn;
  const int z = wg / T, x = wg - z * T;
  const int xcd = x & 7, local = x >> 3;
  int base = 0;
  for (int y = 0; y < xcd; ++y) base += (T - y + 7) >> 3;
  const int id = base + local;
  const int tn = id / g.ntm, tm = id - tn * g.ntm;
  const int m0 = tm * TM, n0 = tn * TN;
  const int batch = z / g.ksplit, kz = z - batch * g.ksplit;
  // ragged batch: the last element may have fewer rows (forward / dX products) or a shorter contraction
  // (dW products, whose K runs over the rows)
  const int M = (batch == g.batch - 1) ? g.m_last : g.M;
  const int K = (batch == g.batch - 1) ? g.k_last : g.K;
  if (m0 >= M) return;
  const int amode = g.amode, bmode = g.bmode;
  // K range of this split: chunks of GK dealt as evenly as possible
  const int nchunks_all = (K + GK - 1) / GK;
  const int c_lo = (int)((long)nchunks_all * kz / g.ksplit), c_hi = (int)((long)nchunks_all * (kz + 1) / g.ksplit);
  const int nchunks = c_hi - c_lo;
  const bool vec = g.vec != 0;

  const float *A = g.A + (size_t)batch * g.sA;
  const float *B = g.B + (size_t)batch * g.sB;
  float *C = g.C + (size_t)batch * g.sC;

  f32x16 acc[WM][WN];
#pragma unroll
  for (int a = 0; a < WM; ++a)
#pragma unroll
    for (int b = 0; b < WN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const bool want_rowsum = (g.rowsum != nullptr) && (tn == 0);
  f32x4 rs = f32x4{0.f, 0.f, 0.f, 0.f};

  // interior tiles (the common case) need no masking at all
  const bool edge = (m0 + TM > M) || (n0 + TN > g.N) || (K % GK != 0);
  TileA ta;
  TileB tb;
  if (nchunks > 0) {
    ta.load(amode, vec, A, g.lda, m0, c_lo * GK, M, K, tid);
    tb.load(bmode, vec, B, g.ldb, n0, c_lo * GK, g.N, K, tid);
    if (edge) {
      ta.mask(amode, m0, c_lo * GK, M, K, tid);
      tb.mask(bmode, n0, c_lo * GK, g.N, K, tid);
    }
    if (want_rowsum) {
#pragma unroll
      for (int q = 0; q < TileA::PER; ++q) rs += ta.r[q];
    }
    ta.store(amode, s_a, tid);
    tb.store(bmode, s_b, tid);
  }
  __syncthreads();

  for (int c = 0; c < nchunks; ++c) {
    const int st = c & 1;
    const bool more = c + 1 < nchunks;
    if (more) {
      ta.load(amode, vec, A, g.lda, m0, (c_lo + c + 1) * GK, M, K, tid);
      tb.load(bmode, vec, B, g.ldb, n0, (c_lo + c + 1) * GK, g.N, K, tid);
    }
    const float *sa = s_a + st * (GK * LDA) + wm * (32 * WM) + half * LDA;
    const float *sb = s_b + st * (GK * LDB) + wn * (32 * WN) + half * LDB;
    // operands of 4 k-steps per register group (one swizzle value per group: k >> 3 = step >> 2), the next
    // group's LDS reads issued before this group's MFMAs
    constexpr int GS = 4, NG = GK / 2 / GS;
    float av[2][GS][WM], bv[2][GS][WN];
#pragma unroll
    for (int t = 0; t < GS; ++t) {
#pragma unroll
      for (int a = 0; a < WM; ++a) av[0][t][a] = sa[(2 * t) * LDA + 32 * a + l31];
#pragma unroll
      for (int b = 0; b < WN; ++b) bv[0][t][b] = sb[(2 * t) * LDB + 32 * b + l31];
    }
    __builtin_amdgcn_sched_group_barrier(0x100, GS * (WM + WN), 0);
#pragma unroll
    for (int gq = 0; gq < NG; ++gq) {
      if (gq + 1 < NG) {
        const int lx = l31 ^ (((gq + 1) & 3) << 2);
#pragma unroll
        for (int t = 0; t < GS; ++t) {
#pragma unroll
          for (int a = 0; a < WM; ++a) av[(gq + 1) & 1][t][a] = sa[(2 * ((gq + 1) * GS + t)) * LDA + 32 * a + lx];
#pragma unroll
          for (int b = 0; b < WN; ++b) bv[(gq + 1) & 1][t][b] = sb[(2 * ((gq + 1) * GS + t)) * LDB + 32 * b + lx];
        }
      }
#pragma unroll
      for (int t = 0; t < GS; ++t) {
#pragma unroll
        for (int a = 0; a < WM; ++a)
#pragma unroll
          for (int b = 0; b < WN; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[gq & 1][t][a], bv[gq & 1][t][b], acc[a][b], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x8, WM * WN, 0);
        if (gq + 1 < NG) __builtin_amdgcn_sched_group_barrier(0x100, WM + WN, 0);
      }
    }
    if (more) {
      if (edge) {
        ta.mask(amode, m0, (c_lo + c + 1) * GK, M, K, tid);
        tb.mask(bmode, n0, (c_lo + c + 1) * GK, g.N, K, tid);
      }
      if (want_rowsum) {
#pragma unroll
        for (int q = 0; q < TileA::PER; ++q) rs += ta.r[q];
      }
      ta.store(amode, s_a + (st ^ 1) * (GK * LDA), tid);
      tb.store(bmode, s_b + (st ^ 1) * (GK * LDB), tid);
    }
    __syncthreads();
  }

  // ---- bias gradient of a dW product: rowsum[m] = sum_k A(m, k); A is m-contiguous here, a thread's
  // float4 units all cover the same four m (256 % (TM/4) == 0), threads with equal m meet in LDS
  if (want_rowsum) {
    for (int i = tid; i < TM; i += 256) s_rowsum[i] = 0.f;
    __syncthreads();
    const int mq = 4 * (tid % (TM / 4));
#pragma unroll
    for (int e = 0; e < 4; ++e) atomicAdd(&s_rowsum[mq + e], rs[e]);
    __syncthreads();
    float *rsum = g.rowsum + (size_t)batch * g.sRowsum;
    for (int i = tid; i < TM; i += 256) {
      if (m0 + i < M) {
        if (g.ksplit > 1) unsafeAtomicAdd(rsum + m0 + i, s_rowsum[i]);
        else rsum[m0 + i] = s_rowsum[i];
      }
    }
  }

  // ---- epilogue: C/D map of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  // (a half-wave writes 128 contiguous bytes of one row).  Reads of C (beta = 1) and of the pre-activation are
  // issued for all 16 rows of a tile before the first use: one round trip instead of sixteen.
  const float *bias = g.bias ? g.bias + (size_t)batch * g.sBias : nullptr;
  float *aux = g.aux ? g.aux + (size_t)batch * g.sC : nullptr;
  const bool split = g.ksplit > 1;
#pragma unroll
  for (int a = 0; a < WM; ++a) {
#pragma unroll
    for (int b = 0; b < WN; ++b) {
      const int col = n0 + wn * (32 * WN) + 32 * b + l31;
      const int row0 = m0 + wm * (32 * WM) + 32 * a + 4 * half;
      const bool col_ok = col < g.N;
      const float bv = (bias && kz == 0 && col_ok) ? bias[col] : 0.f;
      float *cp = C + (size_t)row0 * g.ldc + col;
      if (split) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dr = (r & 3) + 8 * (r >> 2);
          if (col_ok && row0 + dr < M) unsafeAtomicAdd(cp + (size_t)dr * g.ldc, acc[a][b][r] + bv);
        }
        continue;
      }
      float cin[16], xin[16];
      if (g.accumulate) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dr = (r & 3) + 8 * (r >> 2);
          cin[r] = (col_ok && row0 + dr < M) ? cp[(size_t)dr * g.ldc] : 0.f;
        }
      }
      if (g.act == 2) {
        const float *xp = aux + (size_t)row0 * g.ldc + col;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dr = (r & 3) + 8 * (r >> 2);
          xin[r] = (col_ok && row0 + dr < M) ? xp[(size_t)dr * g.ldc] : 0.f;
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int dr = (r & 3) + 8 * (r >> 2);
        if (!(col_ok && row0 + dr < M)) continue;
        float v = acc[a][b][r] + bv;
        if (g.act == 1) {
          if (aux) aux[(size_t)(row0 + dr) * g.ldc + col] = v;
          v = gemm_gelu(v);
        } else if (g.act == 2) {
          v *= gemm_gelu_grad(xin[r]);
        }
        if (g.accumulate) v += cin[r];
        cp[(size_t)dr * g.ldc] = v;
      }
    }
  }
}

template <int WGM, int WGN, int WM, int WN>
int launch_gemm(const GemmGroup &grp, hipStream_t stream) {
  constexpr int TM = 32 * WM * WGM, TN = 32 * WN * WGN;
  constexpr size_t lds = sizeof(float) * (2 * GK * (TM + 4 + TN + 4) + TM);
  static bool attr_done = false;  // per template instance
  if (!attr_done) {
    SIG3D_HIP_TRY(hipFuncSetAttribute((const void *)qf_gemm_kernel<WGM, WGN, WM, WN>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    attr_done = true;
  }
  hipLaunchKernelGGL((qf_gemm_kernel<WGM, WGN, WM, WN>), dim3((unsigned)grp.start[grp.nprob]), dim3(256), lds, stream,
                     grp);
  SIG3D_LAUNCH_CHECK("qf_gemm_kernel");
  return 0;
}

void tile_dims(int tile, int *tm, int *tn) {
  switch (tile) {
    case 1: *tm = 64; *tn = 64; break;
    case 2: *tm = 32; *tn = 128; break;
    case 3: *tm = 64; *tn = 128; break;
    default: *tm = 128; *tn = 64; break;
  }
}

// Tile / split heuristic from the sweeps of tools/gemm_bench.py on MI355X: the 64x64 tile wins at every
// Q-Former shape (the wider tiles only at thousands of tiles, where they tie); about three workgroups per CU
// (~768) keep the matrix pipes fed, so K is split -- when the caller allows it -- until that many exist,
// never below 4 chunks (128 k) per split.
void choose_tile(const sig3d_gemm_problem &q, int *tile, int *ksplit) {
  const bool can_split = (q.act == 0) && q.accumulate;
  const long rows = (long)(q.batch - 1) * q.m + q.m_last;
  const long wgs = (long)sig3d_ceil_div(rows, 64) * sig3d_ceil_div(q.n, 64);
  int s = 1;
  if (can_split && wgs < 640) {
    const int chunks = (q.k + GK - 1) / GK;
    s = (int)((768 + wgs / 2) / wgs);
    if (s > chunks / 4) s = chunks / 4;
    if (s < 1) s = 1;
  }
  *tile = 1;
  *ksplit = s;
}

}  // namespace

extern "C" int sig3d_gemm_group(int nprob, const sig3d_gemm_problem *probs, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SIG3D_REQUIRE(nprob >= 1 && nprob <= MAX_GROUP, "1..4 problems per launch");
  SIG3D_REQUIRE(probs != nullptr, "null problem list");
  GemmGroup grp;
  grp.nprob = 0;
  grp.start[0] = 0;
  int tile = probs[0].tile;
  for (int i = 0; i < nprob; ++i) {
    const sig3d_gemm_problem &q = probs[i];
    SIG3D_REQUIRE(q.batch >= 0 && q.m >= 0 && q.n >= 0 && q.k >= 0 && q.m_last >= 0 && q.m_last <= q.m &&
                      q.k_last >= 0 && q.k_last <= q.k, "bad sizes");
    SIG3D_REQUIRE((q.amode == 0 || q.amode == 1) && (q.bmode == 0 || q.bmode == 1),
                  "operand mode must be 0 (k-contiguous) or 1");
    SIG3D_REQUIRE(!(q.amode == 1 && q.bmode == 0), "A m-contiguous with B k-contiguous is not supported");
    SIG3D_REQUIRE(q.act >= 0 && q.act <= 2, "act must be 0 (none), 1 (erf-GELU) or 2 (times gelu'(aux))");
    SIG3D_REQUIRE(q.act != 2 || q.aux != nullptr, "act 2 needs the pre-activation matrix");
    SIG3D_REQUIRE(q.rowsum == nullptr || q.amode == 1, "row sums are taken from an m-contiguous A operand");
    SIG3D_REQUIRE(q.tile >= 0 && q.tile <= 4 && q.ksplit >= 0, "bad tile / split");
    SIG3D_REQUIRE(q.tile == probs[0].tile, "one tile shape per launch");
    if (q.batch == 0 || q.m == 0 || q.n == 0 || q.m_last == 0) continue;
    int ht, hs;
    choose_tile(q, &ht, &hs);
    if (tile == 0) tile = ht;   // the first non-empty problem decides for the group
    int s = q.ksplit ? q.ksplit : hs;
    // a split K meets through atomics on C: everything nonlinear in the sum must see the whole sum, and C
    // must hold the addend (accumulate = 1: zeros or the beta = 1 term)
    if (!((q.act == 0) && q.accumulate)) s = 1;
    const int chunks = (q.k + GK - 1) / GK;
    if (s > chunks) s = chunks > 0 ? chunks : 1;
    GemmArgs &g = grp.p[grp.nprob];
    g.A = q.A; g.B = q.B; g.C = q.C; g.bias = q.bias; g.aux = q.aux; g.rowsum = q.rowsum;
    g.amode = q.amode; g.bmode = q.bmode;
    g.M = q.m; g.N = q.n; g.K = q.k; g.m_last = q.m_last; g.k_last = q.k_last; g.batch = q.batch;
    g.lda = q.lda; g.ldb = q.ldb; g.ldc = q.ldc;
    g.sA = q.stride_a; g.sB = q.stride_b; g.sC = q.stride_c; g.sBias = q.stride_bias; g.sRowsum = q.stride_rowsum;
    g.ksplit = s; g.accumulate = q.accumulate; g.act = q.act;
    // 16-byte loads need aligned bases, strides and a contiguous dimension that is a multiple of 4
    auto al = [](const void *p) { return ((size_t)p & 15) == 0; };
    const int ca = q.amode == MODE_KC ? q.k : q.m, cb = q.bmode == MODE_KC ? q.k : q.n;
    const bool ragged_ok = (q.amode == MODE_KC || q.m_last % 4 == 0) && (q.amode == MODE_MC || q.k_last % 4 == 0);
    g.vec = al(q.A) && al(q.B) && q.lda % 4 == 0 && q.ldb % 4 == 0 && q.stride_a % 4 == 0 && q.stride_b % 4 == 0 &&
            ca % 4 == 0 && cb % 4 == 0 && ragged_ok;
    int tm, tn;
    tile_dims(tile, &tm, &tn);
    g.ntm = sig3d_ceil_div(q.m, tm);
    g.ntn = sig3d_ceil_div(q.n, tn);
    grp.start[grp.nprob + 1] = grp.start[grp.nprob] + g.ntm * g.ntn * q.batch * s;
    ++grp.nprob;
  }
  if (grp.nprob == 0) return 0;
  for (int i = grp.nprob; i < MAX_GROUP; ++i) grp.start[i + 1] = grp.start[grp.nprob];
  switch (tile) {
    case 1: return launch_gemm<2, 2, 1, 1>(grp, stream);
    case 2: return launch_gemm<1, 4, 1, 1>(grp, stream);
    case 3: return launch_gemm<2, 2, 1, 2>(grp, stream);
    default: return launch_gemm<2, 2, 2, 1>(grp, stream);
  }
}

extern "C" int sig3d_gemm(int amode, int bmode, int batch, int m, int n, int k, const float *A, int lda,
                          long stride_a, const float *B, int ldb, long stride_b, float *C, int ldc, long stride_c,
                          const float *bias, long stride_bias, int act, float *aux, int accumulate, float *rowsum,
                          long stride_rowsum, int tile, int ksplit, void *stream_) {
  sig3d_gemm_problem q;
  q.amode = amode; q.bmode = bmode; q.batch = batch; q.m = m; q.n = n; q.k = k; q.m_last = m; q.k_last = k;
  q.A = A; q.lda = lda; q.stride_a = stride_a; q.B = B; q.ldb = ldb; q.stride_b = stride_b;
  q.C = C; q.ldc = ldc; q.stride_c = stride_c; q.bias = bias; q.stride_bias = stride_bias; q.act = act; q.aux = aux;
  q.accumulate = accumulate; q.rowsum = rowsum; q.stride_rowsum = stride_rowsum; q.tile = tile; q.ksplit = ksplit;
  return sig3d_gemm_group(1, &q, stream_);
}
