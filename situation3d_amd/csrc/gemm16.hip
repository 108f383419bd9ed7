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
#include "sig3d_common.h"

namespace {


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
  if (q.act != 0 || config != 1) return 1;
  const int t64 = tiles_of(q, 64, 64);
  const int chunks = sig3d_ceil_div(q.k, gemm16::BK);
  // ~one workgroup per CU for short reductions, two for long ones (measured alone AND inside the training step:
  // 256 / 384 everywhere cost the step +0.15 / +0.07 ms against this rule although every slab is re-read by the
  // LayerNorm tail that consumes the product)
  const int target = q.k <= 1024 ? 256 : 512;
  int s = (target + t64 / 2) / (t64 > 0 ? t64 : 1);
  if (s > chunks / 4) s = chunks / 4;
  if (s > 8) s = 8;
  return s < 1 ? 1 : s;
}

// dW = slab 0 (already in dW) + the other slabs.  64 float4 columns x 4 slab groups per workgroup: a thread folds every
// fourth slab of its column, eight loads in flight, the four partial sums meet in LDS (one thread per column and 127
// dependent trips took 14 us for 8 MB that sit in L2).
__global__ __launch_bounds__(256) void sum_slabs_kernel(int n4, int nslabs, size_t slab4, float4 *__restrict__ dst,
                                                        const float4 *__restrict__ slabs) {
  __shared__ float4 s_part[4][64];
  const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int i = min((int)blockIdx.x * 64 + c, n4 - 1);
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int s0 = g; s0 < nslabs; s0 += 32) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = slabs[(size_t)min(s0 + 4 * u, nslabs - 1) * slab4 + i];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (s0 + 4 * u < nslabs) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
  }
  s_part[g][c] = a;
  __syncthreads();
  if (g == 0 && (int)blockIdx.x * 64 + c < n4) {
    const float4 d = dst[i], p0 = s_part[0][c], p1 = s_part[1][c], p2 = s_part[2][c], p3 = s_part[3][c];
    dst[i] = make_float4(d.x + ((p0.x + p1.x) + (p2.x + p3.x)), d.y + ((p0.y + p1.y) + (p2.y + p3.y)),
                         d.z + ((p0.z + p1.z) + (p2.z + p3.z)), d.w + ((p0.w + p1.w) + (p2.w + p3.w)));
  }
}

constexpr int SIG3D_DW_STREAM_WGS = 512;   // workgroups a k-streaming weight gradient aims for (256 / 1024: slower in the step)

// 131 / 259 input channels (3 coordinates in front of 128 / 256 features): 64-wide column tiles would give a third / a
// fifth of the workgroups 3 live columns; 48-wide ones (4 waves of 16 x 48) waste 9-10 % instead
inline bool dw_narrow_tiles(int cin) { return cin % 64 != 0 && cin % 64 <= 16 && cin > 64; }

int dw_stream_splits(int b, int cin, int cout, long e) {
  const int tiles = sig3d_ceil_div(cout, 64) * sig3d_ceil_div(cin, dw_narrow_tiles(cin) ? 48 : 64);
  int s = SIG3D_DW_STREAM_WGS / (tiles * (b > 0 ? b : 1));          // default: two 64 x 64 workgroups per CU
  const long chunks = (e + 31) / 32;
  if (s > chunks / 4) s = (int)(chunks / 4);
  if (s > 16) s = 16;
  return s < 1 ? 1 : s;
}

}  // namespace

// The weight gradient of a SharedMLP layer (pytorch_utils.py:11-36: Conv2d 1x1 backward) as a k-STREAMING product:
// dW (cout, cin) = sum_b dY[b] a[b]^T, positions as the reduction index, both operands read along their rows in 16-byte
// requests (mlp_dw_kernel reads one 64-byte run per LANE and row: 1-2.5 TB/s and a tail of f32 atomics; this form is
// bound by the f32 matrix pipe or by HBM).  a = x, or relu(x * pscale + pshift) applied on the way to LDS.  Every
// (sample, split) pair writes its own 64 x 64 tiles into a slab; sum_slabs_kernel folds them: fixed order, no atomics.
// n_act (compact lists): sample i reduces over its first n_act[i] positions only; e stays the row stride.
// dW is OVERWRITTEN.  work: sig3d_mlp_layer_dw_stream_work_floats(b, cin, cout, e) floats of scratch.
extern "C" long sig3d_mlp_layer_dw_stream_work_floats(int b, int cin, int cout, long e) {
  const long pairs = (long)b * dw_stream_splits(b, cin, cout, e);
  return (pairs > 1 ? pairs - 1 : 0) * (((long)cout * cin + 3) / 4 * 4);
}

static int dw_stream_impl(int b, int cin, int cout, long e, const float *dY, const float *x, const float *pscale,
                          const float *pshift, const int *n_act, float *dW, float *work, bool fold, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SIG3D_REQUIRE(b >= 0 && cin >= 1 && cout >= 1 && e >= 0 && e % 4 == 0, "bad size (rows of 16-byte multiples)");
  SIG3D_REQUIRE((pscale == nullptr) == (pshift == nullptr), "pscale/pshift must come together");
  SIG3D_REQUIRE(dW, "null operand");
  if (b == 0 || e == 0) {   // an empty sum
    SIG3D_HIP_TRY(hipMemsetAsync(dW, 0, sizeof(float) * (size_t)cout * cin, stream));
    return 0;
  }
  SIG3D_REQUIRE(dY && x, "null operand");
  auto al = [](const void *q) { return ((size_t)q & 15) == 0; };
  SIG3D_REQUIRE(al(dY) && al(x), "operands must be 16-byte aligned");
  SIG3D_REQUIRE((size_t)cout * e * 4 < (1ull << 31) && (size_t)cin * e * 4 < (1ull << 31), "a sample's operand exceeds 2 GB");
  const int splits = dw_stream_splits(b, cin, cout, e);
  const long pairs = (long)b * splits;
  SIG3D_REQUIRE(pairs == 1 || work != nullptr, "work space missing (sig3d_mlp_layer_dw_stream_work_floats)");
  const long slab = ((long)cout * cin + 3) / 4 * 4;
  gemm16::Problem p = {};
  p.A = dY; p.B = x; p.C = dW; p.Cs = work;
  p.M = cout; p.N = cin; p.K = (int)e;
  p.lda = (int)e; p.ldb = (int)e; p.ldc = cin;
  p.sA = (long)cout * e; p.sB = (long)cin * e; p.sC = 0; p.slab = slab;
  p.batch = b; p.splits = splits; p.act = 0;
  p.k_dev = n_act; p.b_scale = pscale; p.b_shift = pshift;
  hipError_t err = dw_narrow_tiles(cin) ? gemm16::launch<1, 3, 4, 1, 4, 3, true>(p, gemm16::B_KC, stream)
                                        : gemm16::launch<1, 2, 4, 2, 4, 2, true>(p, gemm16::B_KC, stream);
  if (err != hipSuccess) { sig3d_set_error("gemm16_kernel (weight gradient)", err); return (int)err; }
  if (pairs > 1 && fold) {
    SIG3D_REQUIRE(((size_t)dW & 15) == 0 && ((long)cout * cin) % 4 == 0, "dW: 16-byte aligned, cout * cin a multiple of 4");
    const int n4 = (int)((long)cout * cin / 4);
    hipLaunchKernelGGL(sum_slabs_kernel, dim3(sig3d_ceil_div(n4, 64)), dim3(256), 0, stream, n4, (int)(pairs - 1),
                       (size_t)(slab / 4), reinterpret_cast<float4 *>(dW), reinterpret_cast<const float4 *>(work));
    SIG3D_LAUNCH_CHECK("sum_slabs_kernel");
  }
  return 0;
}

// For shared_mlp.hip's launch of a layer's weight gradient and input gradient as two workgroup ranges: the weight
// gradient's problem as dw_stream_impl sets it up (no launch).  *usable = 0 when the product would take another tiling
// than <1, 2, 4, 2> (48-wide column tiles) or is empty; grid = workgroups of the product.
extern "C" int sig3d_internal_dw_stream_problem(int b, int cin, int cout, long e, const float *dY, const float *x,
                                                const float *pscale, const float *pshift, const int *n_act, float *dW,
                                                float *work, void *problem_out, int *grid, int *usable) {
  *usable = 0; *grid = 0;
  if (b <= 0 || e <= 0 || e % 4 != 0 || dw_narrow_tiles(cin) || !dY || !x || !dW) return 0;
  auto al = [](const void *q) { return ((size_t)q & 15) == 0; };
  if (!al(dY) || !al(x) || (size_t)cout * e * 4 >= (1ull << 31) || (size_t)cin * e * 4 >= (1ull << 31)) return 0;
  const int splits = dw_stream_splits(b, cin, cout, e);
  if ((long)b * splits > 1 && work == nullptr) return 0;
  gemm16::Problem p = {};
  p.A = dY; p.B = x; p.C = dW; p.Cs = work;
  p.M = cout; p.N = cin; p.K = (int)e;
  p.lda = (int)e; p.ldb = (int)e; p.ldc = cin;
  p.sA = (long)cout * e; p.sB = (long)cin * e; p.sC = 0; p.slab = ((long)cout * cin + 3) / 4 * 4;
  p.batch = b; p.splits = splits; p.act = 0;
  p.k_dev = n_act; p.b_scale = pscale; p.b_shift = pshift;
  p.ntm = (p.M + 63) / 64; p.ntn = (p.N + 63) / 64;
  *reinterpret_cast<gemm16::Problem *>(problem_out) = p;
  *grid = p.ntm * p.ntn * p.splits * p.batch;
  *usable = 1;
  return 0;
}

extern "C" int sig3d_mlp_layer_dw_stream(int b, int cin, int cout, long e, const float *dY, const float *x,
                                         const float *pscale, const float *pshift, const int *n_act, float *dW,
                                         float *work, void *stream_) {
  return dw_stream_impl(b, cin, cout, e, dY, x, pscale, pshift, n_act, dW, work, true, stream_);
}

// The same product WITHOUT the fold: dW holds slab 0, `work` the others; the caller folds the layers of a level with one
// sig3d_sum_slabs_multi launch (a fold per layer was 11 launches of ~5 us per training step).
extern "C" int sig3d_mlp_layer_dw_stream_nofold(int b, int cin, int cout, long e, const float *dY, const float *x,
                                                const float *pscale, const float *pshift, const int *n_act,
                                                float *dW, float *work, void *stream_) {
  return dw_stream_impl(b, cin, cout, e, dY, x, pscale, pshift, n_act, dW, work, false, stream_);
}

namespace {
struct SumSlabsJobs {
  sig3d_sum_slabs_job job[SIG3D_SUM_SLABS_MAX_JOBS];
  int first_block[SIG3D_SUM_SLABS_MAX_JOBS + 1];
  int njobs;
  // riding along: cvt_dst[i] = (float)cvt_src[i], i < cvt_n, by the workgroups from first_block[njobs] on (the f64
  // BatchNorm-gradient sums of the same stack become its f32 d gamma / d beta: a conversion launch per stack before)
  const double *cvt_src;
  float *cvt_dst;
  int cvt_n;
};

// sum_slabs_kernel for several (dst, slabs) pairs: same columns per workgroup, same order of additions per job
__global__ __launch_bounds__(256) void sum_slabs_multi_kernel(SumSlabsJobs js) {
  __shared__ float4 s_part[4][64];
  if ((int)blockIdx.x >= js.first_block[js.njobs]) {      // (uniform) the conversion's workgroups
    const int i = ((int)blockIdx.x - js.first_block[js.njobs]) * 256 + (int)threadIdx.x;
    if (i < js.cvt_n) js.cvt_dst[i] = (float)js.cvt_src[i];
    return;
  }
  int j = 0;
  while (j + 1 < js.njobs && (int)blockIdx.x >= js.first_block[j + 1]) ++j;
  const sig3d_sum_slabs_job q = js.job[j];
  const int n4 = (int)(q.n / 4), nslabs = q.nslabs;
  const size_t slab4 = (size_t)(q.slab_stride / 4);
  float4 *dst = reinterpret_cast<float4 *>(q.dst);
  const float4 *slabs = reinterpret_cast<const float4 *>(q.slabs);
  const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int blk = (int)blockIdx.x - js.first_block[j];
  const int i = min(blk * 64 + c, n4 - 1);
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int s0 = g; s0 < nslabs; s0 += 32) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = slabs[(size_t)min(s0 + 4 * u, nslabs - 1) * slab4 + i];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (s0 + 4 * u < nslabs) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
  }
  s_part[g][c] = a;
  __syncthreads();
  if (g == 0 && blk * 64 + c < n4) {
    const float4 d = dst[i], p0 = s_part[0][c], p1 = s_part[1][c], p2 = s_part[2][c], p3 = s_part[3][c];
    dst[i] = make_float4(d.x + ((p0.x + p1.x) + (p2.x + p3.x)), d.y + ((p0.y + p1.y) + (p2.y + p3.y)),
                         d.z + ((p0.z + p1.z) + (p2.z + p3.z)), d.w + ((p0.w + p1.w) + (p2.w + p3.w)));
  }
}
}  // namespace

extern "C" int sig3d_sum_slabs_multi(int njobs, const sig3d_sum_slabs_job *jobs, const double *cvt_src, float *cvt_dst,
                                     int cvt_n, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SIG3D_REQUIRE(cvt_n >= 0 && (cvt_n == 0 || (cvt_src && cvt_dst)), "conversion: null operand");
  SIG3D_REQUIRE(njobs >= 0 && njobs <= SIG3D_SUM_SLABS_MAX_JOBS, "at most SIG3D_SUM_SLABS_MAX_JOBS jobs per launch");
  SIG3D_REQUIRE(njobs == 0 || jobs != nullptr, "null job list");
  SumSlabsJobs js = {};
  int blocks = 0;
  for (int j = 0; j < njobs; ++j) {
    const sig3d_sum_slabs_job &q = jobs[j];
    SIG3D_REQUIRE(q.n >= 0 && q.nslabs >= 0 && q.n % 4 == 0 && q.slab_stride % 4 == 0 && q.slab_stride >= q.n,
                  "n and the slab stride in multiples of 4 floats");
    if (q.n == 0 || q.nslabs == 0) continue;
    SIG3D_REQUIRE(q.dst && q.slabs && ((size_t)q.dst & 15) == 0 && ((size_t)q.slabs & 15) == 0, "16-byte aligned operands");
    js.job[js.njobs] = q;
    js.first_block[js.njobs] = blocks;
    blocks += sig3d_ceil_div(q.n / 4, 64);
    ++js.njobs;
  }
  js.first_block[js.njobs] = blocks;
  js.cvt_src = cvt_src; js.cvt_dst = cvt_dst; js.cvt_n = cvt_n;
  blocks += sig3d_ceil_div(cvt_n, 256);
  if (blocks == 0) return 0;
  hipLaunchKernelGGL(sum_slabs_multi_kernel, dim3(blocks), dim3(256), 0, stream, js);
  SIG3D_LAUNCH_CHECK("sum_slabs_multi_kernel");
  return 0;
}

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
  SIG3D_REQUIRE(q.config >= 0 && q.config <= 3, "config must be 0 (choose) .. 3");
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
    default: e = gemm16::launch<2, 2, 2, 4, 4, 1>(p, q.bmode, stream); break;
  }
  if (e != hipSuccess) {
    sig3d_set_error("gemm16_kernel", e);
    return (int)e;
  }
  return 0;
}
