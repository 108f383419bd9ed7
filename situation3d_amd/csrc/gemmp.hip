// gemmp.hip -- C ABI of the f32 GEMM on the bf16 matrix cores over pre-split operands (gemmp_core.h) and of the pass
// that splits a matrix into its chunked bf16 planes.
//   3DLLM_BLIP2-base/lavis/models/blip2_models/Qformer.py:116-118, :238, :305, :320: the weight gradients
//   dW = dY^T X of the Q-Former's dense layers (layer-batched: qformer._WeightGradArena), and -- as selectable forms --
//   the forward products y = x W^T and the input gradients dX = dY W.
// Which tiling a product gets: measurements of tools/micro/gemmp_bench.hip on MI355X (profiles/r05_gemmp.md).  A CU
// pulls 20-29 bytes per clock from L2 into LDS and stores 79 bytes per clock from registers to LDS whatever the
// instruction mix, so a product runs at the matrix pipe's rate only with 128 x 128 tiles (32 bytes per MFMA clock);
// the weight gradients have thousands of those, the 416-row products of the forward / input-gradient chain do not
// (DESIGN.md 4h).
#include <cstdlib>
#include "gemmp_core.h"
#include "sig3d_common.h"

namespace {

// X (rows, cols) f32, row stride ld  ->  chunked planes [cols / 32][row capacity][3][32] bf16 (x = p1 + p2 + p3 exactly).
// A thread splits 8 consecutive columns; threads are ordered (32-column chunk, row, octet): a wave reads 16 rows x 128
// bytes (whole lines) and writes 16 rows x 192 bytes = 3 KiB of consecutive memory.
__global__ __launch_bounds__(256) void planes_split_kernel(int rows, int cols, const float *__restrict__ src, int ld,
                                                           long src_stride, unsigned short *__restrict__ dst, long chunk,
                                                           long dst_stride) {
  const long per = (long)rows * (cols >> 3);
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= per) return;
  const int q = (int)(i & 3);
  const long t = i >> 2;
  const int r = (int)(t % rows), cc = (int)(t / rows);
  const float *s = src + (size_t)blockIdx.y * src_stride + (size_t)r * ld + cc * 32 + q * 8;
  const gemmp::f32x4 lo = *reinterpret_cast<const gemmp::f32x4 *>(s), hi = *reinterpret_cast<const gemmp::f32x4 *>(s + 4);
  unsigned t1[4], t2[4], t3[4];
  gemmp::split_pair(lo[0], lo[1], t1[0], t2[0], t3[0]);
  gemmp::split_pair(lo[2], lo[3], t1[1], t2[1], t3[1]);
  gemmp::split_pair(hi[0], hi[1], t1[2], t2[2], t3[2]);
  gemmp::split_pair(hi[2], hi[3], t1[3], t2[3], t3[3]);
  unsigned short *d = dst + (size_t)blockIdx.y * dst_stride + (size_t)cc * chunk + (size_t)r * 96 + q * 8;
  *reinterpret_cast<gemmp::u32x4 *>(d) = gemmp::u32x4{t1[0], t1[1], t1[2], t1[3]};
  *reinterpret_cast<gemmp::u32x4 *>(d + 32) = gemmp::u32x4{t2[0], t2[1], t2[2], t2[3]};
  *reinterpret_cast<gemmp::u32x4 *>(d + 64) = gemmp::u32x4{t3[0], t3[1], t3[2], t3[3]};
}

struct Tiling { int tm, tn; };

Tiling tiling_of(int config) {
  switch (config) {
    case 1: return {64, 64};
    case 2: return {64, 128};
    default: return {128, 128};
  }
}

int choose_config(const sig3d_gemmp_problem &q) {
  if (q.config) return q.config;
  auto tiles = [&](int tm, int tn) { return (long)q.batch * sig3d_ceil_div(q.m, tm) * sig3d_ceil_div(q.n, tn); };
  if (tiles(128, 128) >= 400) return 3;     // two rounds of the chip and more: the tile with the most reuse
  if (tiles(64, 128) >= 200) return 2;
  return 1;
}

}  // namespace

extern "C" int sig3d_planes_split(int batch, int rows, int cols, const float *src, int ld, long src_stride,
                                  void *planes, long chunk_stride, long planes_stride, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SIG3D_REQUIRE(batch >= 0 && rows >= 0 && cols >= 0 && cols % 32 == 0, "bad size (columns in chunks of 32)");
  SIG3D_REQUIRE(ld >= cols && ld % 4 == 0 && src_stride % 4 == 0, "source rows must be 16-byte aligned");
  SIG3D_REQUIRE(chunk_stride >= (long)rows * 96 && chunk_stride % 8 == 0 && planes_stride % 8 == 0, "bad plane strides");
  if (batch == 0 || rows == 0 || cols == 0) return 0;
  SIG3D_REQUIRE(src && planes && ((size_t)src & 15) == 0 && ((size_t)planes & 15) == 0, "null or misaligned operand");
  const long per = (long)rows * (cols / 8);
  hipLaunchKernelGGL(planes_split_kernel, dim3((unsigned)((per + 255) / 256), (unsigned)batch), dim3(256), 0, stream, rows, cols,
                     src, ld, src_stride, reinterpret_cast<unsigned short *>(planes), chunk_stride, planes_stride);
  SIG3D_LAUNCH_CHECK("planes_split_kernel");
  return 0;
}

extern "C" long sig3d_gemmp_work_floats(int batch, int m, int n, int splits, int config) {
  if (splits <= 1) return 0;
  sig3d_gemmp_problem q = {};
  q.batch = batch; q.m = m; q.n = n; q.config = config;
  const Tiling t = tiling_of(choose_config(q));
  return (long)splits * batch * sig3d_ceil_div(m, t.tm) * sig3d_ceil_div(n, t.tn) * t.tm * t.tn;
}

extern "C" int sig3d_gemmp(const sig3d_gemmp_problem *qp, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SIG3D_REQUIRE(qp != nullptr, "null problem");
  const sig3d_gemmp_problem &q = *qp;
  SIG3D_REQUIRE(q.batch >= 0 && q.m >= 0 && q.n >= 0 && q.k >= 1, "bad sizes");
  SIG3D_REQUIRE(q.modes >= 0 && q.modes <= 2, "modes: 0 forward (x W^T), 1 input gradient (dY W), 2 weight gradient (dY^T X)");
  SIG3D_REQUIRE(q.act >= 0 && q.act <= 2, "act must be 0 (none), 1 (erf-GELU) or 2 (times gelu'(aux))");
  SIG3D_REQUIRE(q.act != 2 || q.aux != nullptr, "act 2 needs the pre-activation matrix");
  SIG3D_REQUIRE(q.config >= 0 && q.config <= 3, "config must be 0 (choose) .. 3");
  SIG3D_REQUIRE(q.splits >= 0 && q.splits <= 16, "splits must be 0 / 1 (none) .. 16");
  if (q.batch == 0 || q.m == 0 || q.n == 0) return 0;
  // an operand whose columns are reduced over comes in whole 32-column chunks; output indices in pieces of 8
  SIG3D_REQUIRE(q.m % 8 == 0 && q.n % 8 == 0, "m and n must be multiples of 8");
  SIG3D_REQUIRE(q.modes == 2 || q.k % 32 == 0, "k must be a multiple of 32 when an operand's columns are the reduction index");
  SIG3D_REQUIRE(q.modes != 2 || q.m % 32 == 0, "weight gradient: m (the columns of A) in chunks of 32");
  SIG3D_REQUIRE(q.modes == 0 || q.n % 32 == 0, "n (the columns of B) in chunks of 32");
  auto al = [](const void *p) { return ((size_t)p & 15) == 0; };
  SIG3D_REQUIRE(q.A && q.B && al(q.A) && al(q.B) && q.chunk_a % 8 == 0 && q.chunk_b % 8 == 0 && q.stride_a % 8 == 0 &&
                    q.stride_b % 8 == 0, "A / B: 16-byte aligned chunked planes");
  SIG3D_REQUIRE(q.bytes_a > 0 && q.bytes_b > 0 && q.bytes_a < (1ll << 32) - 4096 && q.bytes_b < (1ll << 32) - 4096,
                "bytes_a / bytes_b: the readable extent of one batch element, below 4 GB");
  SIG3D_REQUIRE(q.C || q.C_planes, "no result requested");
  SIG3D_REQUIRE(!q.C || (al(q.C) && q.ldc % 4 == 0 && q.stride_c % 4 == 0), "C: 16-byte aligned rows");
  SIG3D_REQUIRE(!q.C_planes || (al(q.C_planes) && q.n % 32 == 0 && q.chunk_c >= (long)q.m * 96), "C planes: n in chunks of 32");
  const int splits = q.splits < 1 ? 1 : q.splits;
  SIG3D_REQUIRE(splits == 1 || (q.work && q.counters), "a split reduction needs work space and zeroed counters");
  SIG3D_REQUIRE(splits <= sig3d_ceil_div(q.k, gemmp::BK), "more splits than 32-deep chunks of k");

  gemmp::Problem p = {};
  p.A = reinterpret_cast<const unsigned short *>(q.A);
  p.B = reinterpret_cast<const unsigned short *>(q.B);
  p.csA = q.chunk_a; p.csB = q.chunk_b;
  p.extA = q.bytes_a; p.extB = q.bytes_b;
  p.C = q.C; p.ldc = q.ldc;
  p.Cp = reinterpret_cast<unsigned short *>(q.C_planes); p.csC = q.chunk_c;
  p.bias = q.bias; p.addend = q.addend; p.aux = q.aux;
  p.ws = q.work; p.cnt = q.counters;
  p.M = q.m; p.N = q.n; p.K = q.k;
  p.sA = q.stride_a; p.sB = q.stride_b; p.sC = q.stride_c; p.sCp = q.stride_cp; p.sBias = q.stride_bias;
  p.batch = q.batch; p.splits = splits; p.act = q.act;
  hipError_t e;
  switch (choose_config(q)) {
    case 1: e = gemmp::launch<1, 1, 2, 2, 4, 1>(p, q.modes, stream); break;     // 64 x 64, 4 waves of 32 x 32
    case 2: e = gemmp::launch<1, 1, 2, 4, 4, 1>(p, q.modes, stream); break;     // 64 x 128, 8 waves of 32 x 32
    default: e = gemmp::launch<2, 1, 2, 4, 4, 1>(p, q.modes, stream); break;    // 128 x 128, 8 waves of 64 x 32
  }
  if (e != hipSuccess) {
    sig3d_set_error("gemmp_kernel", e);
    return (int)e;
  }
  return 0;
}
