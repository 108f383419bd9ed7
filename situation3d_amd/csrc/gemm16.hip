// gemm16.hip -- C ABI of the exact-f32 MFMA GEMM core (gemm16_core.h) behind the Q-Former's dense layers:
//   3DLLM_BLIP2-base/lavis/models/blip2_models/Qformer.py:116-118 (query / key / value), :238 (BertSelfOutput.dense),
//   :305 (BertIntermediate.dense + erf-GELU), :320 (BertOutput.dense) and the input-gradient products of their
//   backward passes (dX = dY W: the weight is then read with its OUTPUT index as the reduction index).
// Which tiling a product gets is decided here from measurements of tools/micro/gemm16_bench.hip on MI355X
// (profiles/r04_gemm16_shapes.md): a wave cannot hide an LDS read, an LDS store or a memory request behind its OWN
// f32 MFMA (tools/micro/mfma16_fill.hip: one ds_read_b128 costs a lone wave ~41 cycles wherever it is placed), only
// behind the MFMAs of the other waves on its SIMD -- so the tilings keep two to four waves per SIMD resident:
//   A  64 x 64 per workgroup, 8 waves of 16 x 32, 2 workgroups / CU   few hundred rows, up to ~300 tiles
//   B  32 x 64,               4 waves of 16 x 32, 3 workgroups / CU   300 .. 1500 tiles (unsplittable epilogues)
//   C  64 x 128,              8 waves of 32 x 32                      thousands of tiles
// and the reduction is split (every split writes its own slab of C, gemm16_core.h) until ~256 (k <= 1024) or ~512
// workgroups exist.
#include <cstdlib>
#include "gemm16_core.h"
#include "gemmx6_core.h"
#include "sig3d_common.h"

namespace {

const int SIG3D_GEMM16_TARGET_WGS = getenv("SIG3D_GEMM16_TARGET_WGS") ? atoi(getenv("SIG3D_GEMM16_TARGET_WGS")) : 0;

int tiles_of(const sig3d_gemm16_problem &q, int tm, int tn) {
  return q.batch * sig3d_ceil_div(q.m, tm) * sig3d_ceil_div(q.n, tn);
}

int choose_config(const sig3d_gemm16_problem &q) {
  if (q.config) return q.config;
  const int t64 = tiles_of(q, 64, 64);
  if (t64 < 300) return 1;
  if (t64 < 1500) return 2;
  return 3;
}

int choose_splits(const sig3d_gemm16_problem &q, int config) {
  if (q.act == 0 && config >= 11) {    // 64 x 128 tiles: ~256 workgroups, at least 4 chunks each
    const int t = tiles_of(q, 64, 128), chunks = sig3d_ceil_div(q.k, gemm16::BK);
    int s = (256 + t / 2) / (t > 0 ? t : 1);
    if (s > chunks / 4) s = chunks / 4;
    if (s > 8) s = 8;
    return s < 1 ? 1 : s;
  }
  if (q.act != 0 || config != 1) return 1;
  const int t64 = tiles_of(q, 64, 64);
  const int chunks = sig3d_ceil_div(q.k, gemm16::BK);
  // ~one workgroup per CU for short reductions, two for long ones (measured alone AND inside the training step:
  // 256 / 384 everywhere cost the step +0.15 / +0.07 ms against this rule although every slab is re-read by the
  // LayerNorm tail that consumes the product); SIG3D_GEMM16_TARGET_WGS overrides both
  const int target = SIG3D_GEMM16_TARGET_WGS > 0 ? SIG3D_GEMM16_TARGET_WGS : (q.k <= 1024 ? 256 : 512);
  int s = (target + t64 / 2) / (t64 > 0 ? t64 : 1);
  if (s > chunks / 4) s = chunks / 4;
  if (s > 8) s = 8;
  return s < 1 ? 1 : s;
}

}  // namespace

extern "C" int sig3d_gemm16_splits(int bmode, int batch, int m, int n, int k, int act, int config) {
  sig3d_gemm16_problem q = {};
  q.bmode = bmode; q.batch = batch; q.m = m; q.n = n; q.k = k; q.act = act; q.config = config;
  return choose_splits(q, choose_config(q));
}

extern "C" int sig3d_gemm16(const sig3d_gemm16_problem *qp, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SIG3D_REQUIRE(qp != nullptr, "null problem");
  const sig3d_gemm16_problem &q = *qp;
  SIG3D_REQUIRE(q.batch >= 0 && q.m >= 0 && q.n >= 0 && q.k >= 1, "bad sizes");
  SIG3D_REQUIRE(q.bmode == 0 || q.bmode == 1, "bmode must be 0 (B rows are k-contiguous) or 1 (n-contiguous)");
  SIG3D_REQUIRE(q.act >= 0 && q.act <= 2, "act must be 0 (none), 1 (erf-GELU) or 2 (times gelu'(aux))");
  SIG3D_REQUIRE(q.act != 2 || q.aux != nullptr, "act 2 needs the pre-activation matrix");
  SIG3D_REQUIRE(q.splits >= 1 && q.splits <= 64, "splits must be 1 .. 64 (sig3d_gemm16_splits proposes a count)");
  SIG3D_REQUIRE(q.splits == 1 || (q.act == 0 && q.C_slabs != nullptr), "a split product has no activation and needs slabs");
  SIG3D_REQUIRE((q.config >= 0 && q.config <= 3) || q.config == 11 || q.config == 12,
                "config must be 0 (choose) .. 3, or 11 / 12 (the bf16 x 6 core, gemmx6_core.h)");
  if (q.batch == 0 || q.m == 0 || q.n == 0) return 0;
  // 16-byte requests: the k-contiguous operands need k % 4 == 0 and aligned rows, an n-contiguous B needs n % 4 == 0
  auto al = [](const void *p) { return ((size_t)p & 15) == 0; };
  SIG3D_REQUIRE(al(q.A) && q.lda % 4 == 0 && q.stride_a % 4 == 0 && q.k % 4 == 0, "A: 16-byte aligned rows, k % 4 == 0");
  SIG3D_REQUIRE(al(q.B) && q.ldb % 4 == 0 && q.stride_b % 4 == 0 && (q.bmode == 0 || q.n % 4 == 0),
                "B: 16-byte aligned rows (n % 4 == 0 when n-contiguous)");
  // 32-bit byte offsets inside an operand (buffer addressing)
  const size_t ext_a = ((size_t)(q.m - 1) * q.lda + q.k) * 4;
  const size_t ext_b = (q.bmode == 0 ? (size_t)(q.n - 1) * q.ldb + q.k : (size_t)(q.k - 1) * q.ldb + q.n) * 4;
  SIG3D_REQUIRE(ext_a < (1ull << 31) && ext_b < (1ull << 31), "operand larger than 2 GB per batch element");

  gemm16::Problem p = {};
  p.A = q.A; p.B = q.B; p.C = q.C; p.Cs = q.C_slabs;
  p.bias = q.bias; p.addend = q.addend; p.aux = q.aux;
  p.M = q.m; p.N = q.n; p.K = q.k;
  p.lda = q.lda; p.ldb = q.ldb; p.ldc = q.ldc;
  p.sA = q.stride_a; p.sB = q.stride_b; p.sC = q.stride_c; p.sBias = q.stride_bias;
  p.slab = q.slab_stride;
  p.batch = q.batch; p.splits = q.splits; p.act = q.act;
  const int chunks = sig3d_ceil_div(q.k, gemm16::BK);
  if (p.splits > chunks) p.splits = chunks;   // the surplus slabs are the caller's to ignore: documented
  SIG3D_REQUIRE(p.splits == q.splits, "more splits than 32-deep chunks of k");
  hipError_t e;
  switch (choose_config(q)) {
    case 1: e = gemm16::launch<1, 2, 4, 2, 4, 2>(p, q.bmode, stream); break;
    case 2: e = gemm16::launch<1, 2, 2, 2, 4, 3>(p, q.bmode, stream); break;
    case 11: e = gemmx6::launch<1, 1, 2, 4, 4>(p, q.bmode, stream); break;   // 64 x 128, 8 waves of 32 x 32
    case 12: e = gemmx6::launch<1, 2, 2, 2, 4>(p, q.bmode, stream); break;   // 64 x 128, 4 waves of 32 x 64
    default: e = gemm16::launch<2, 2, 2, 4, 4, 1>(p, q.bmode, stream); break;
  }
  if (e != hipSuccess) {
    sig3d_set_error("gemm16_kernel", e);
    return (int)e;
  }
  return 0;
}
