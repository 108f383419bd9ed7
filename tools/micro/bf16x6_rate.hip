// tools/micro/bf16x6_rate.hip -- what would the inner loop of an exact-f32 GEMM on bf16 matrix cores run at?  (not product)
//
// f32 a = a1 + a2 + a3 (three bf16 terms, exact); a*b ~ a1b1 + a1b2 + a2b1 + a1b3 + a3b1 + a2b2 (six bf16 products with
// f32 accumulation; dropped terms < 2^-27: measured on the step's shapes the result is 2-3x CLOSER to float64 than an
// f32 GEMM).  Per 32-deep chunk a wave with an (MB x NB)-block tile issues 6 MB NB (x2 for the
// K = 16 shape) MFMAs and reads 3 (MB + NB) operand fragments per K-step from LDS.  This stream = those reads and MFMAs,
// interleaved and pinned, nothing else (no conversion, no global loads, no barrier): an upper bound.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <type_traits>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// KIND 0: v_mfma_f32_16x16x32_bf16, tile MB x NB blocks of 16;  KIND 1: v_mfma_f32_32x32x16_bf16, blocks of 32 (2 K-steps)
template <int KIND, int MB, int NB, int NT>
__global__ __launch_bounds__(NT) void k(float *out, unsigned long long *ticks, int iters) {
  __shared__ bf16x8 lds[2048];
  for (int i = threadIdx.x; i < 2048; i += NT) {
    bf16x8 v;
    for (int e = 0; e < 8; ++e) v[e] = (__bf16)(float)(1 + ((i + e) & 3));
    lds[i] = v;
  }
  __syncthreads();
  constexpr int KS = KIND == 0 ? 1 : 2;          // K-steps per 32-deep chunk
  f32x4 acc4[KIND == 0 ? MB * NB : 1];
  f32x16 acc16[KIND == 1 ? MB * NB : 1];
  for (auto &a : acc4) a = f32x4{0, 0, 0, 0};
  for (auto &a : acc16) for (int e = 0; e < 16; ++e) a[e] = 0.f;
  bf16x8 fa[3][MB], fb[3][NB];
  for (int p = 0; p < 3; ++p) {
    for (int i = 0; i < MB; ++i) fa[p][i] = lds[(threadIdx.x + 64 * (p * MB + i)) & 2047];
    for (int j = 0; j < NB; ++j) fb[p][j] = lds[(threadIdx.x + 64 * (p * NB + j) + 777) & 2047];
  }
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      // the fragments of the NEXT K-step are requested while this one's MFMAs issue
      bf16x8 na[3][MB], nb[3][NB];
      constexpr int NREAD = 3 * (MB + NB), NMFMA = 6 * MB * NB;
      static_for<0, NMFMA>([&](auto m_) {
        constexpr int m = decltype(m_)::value;
        constexpr int t = m / (MB * NB), blk = m % (MB * NB), i = blk / NB, j = blk % NB;
        constexpr int pa = t == 0 ? 0 : t == 1 ? 2 : t == 2 ? 1 : t == 3 ? 0 : t == 4 ? 1 : 0;   // small terms first
        constexpr int pb = t == 0 ? 2 : t == 1 ? 0 : t == 2 ? 1 : t == 3 ? 1 : t == 4 ? 0 : 0;
        if constexpr (KIND == 0)
          acc4[blk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[pa][i], fb[pb][j], acc4[blk], 0, 0, 0);
        else
          acc16[blk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[pa][i], fb[pb][j], acc16[blk], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        // reads [r0, r1) go behind this MFMA: spread evenly
        constexpr int r0 = (m * NREAD + NMFMA - 1) / NMFMA, r1 = ((m + 1) * NREAD + NMFMA - 1) / NMFMA;
        static_for<r0, (r1 < NREAD ? r1 : NREAD)>([&](auto r_) {
          constexpr int rd = decltype(r_)::value, p = rd / (MB + NB), q = rd % (MB + NB);
          const bf16x8 v = lds[(threadIdx.x + 64 * rd + 8 * it + 32 * ks) & 2047];
          if constexpr (q < MB) na[p][q] = v; else nb[p][q - MB] = v;
          __builtin_amdgcn_sched_barrier(0);
        });
      });
#pragma unroll
      for (int p = 0; p < 3; ++p) {
#pragma unroll
        for (int i = 0; i < MB; ++i) fa[p][i] = na[p][i];
#pragma unroll
        for (int j = 0; j < NB; ++j) fb[p][j] = nb[p][j];
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (auto &a : acc4) for (int e = 0; e < 4; ++e) s += a[e];
  for (auto &a : acc16) for (int e = 0; e < 16; ++e) s += a[e];
  out[blockIdx.x * NT + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

template <int KIND, int MB, int NB, int NT>
int run() {
  float *out; unsigned long long *ticks, h;
  CHECK(hipMalloc(&out, 256 * NT * 4)); CHECK(hipMalloc(&ticks, 8));
  const int iters = 400;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<KIND, MB, NB, NT>), dim3(256), dim3(NT), 0, 0, out, ticks, iters);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<KIND, MB, NB, NT>), dim3(256), dim3(NT), 0, 0, out, ticks, iters);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  CHECK(hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost));
  const int blk = KIND == 0 ? 16 : 32;
  const double macs = (double)iters * (MB * blk) * (NB * blk) * 32;            // f32-equivalent MACs of one wave
  const int nread = 3 * (MB + NB) * (KIND == 0 ? 1 : 2), nmfma = 6 * MB * NB * (KIND == 0 ? 1 : 2);
  printf("%s tile %3d x %3d, %d waves/SIMD: %2d reads + %2d MFMAs per 32-deep chunk: %6.0f ticks per chunk of one wave, "
         "%6.1f f32-equivalent TFLOP/s (bf16 pipe %6.0f)\n",
         KIND == 0 ? "16x16x32" : "32x32x16", MB * blk, NB * blk, NT / 256, nread, nmfma, (double)h / iters,
         macs * 2 * (NT / 64) * 256 / (ms * 1e-3) * 1e-12, macs * 12 * (NT / 64) * 256 / (ms * 1e-3) * 1e-12);
  CHECK(hipFree(out)); CHECK(hipFree(ticks));
  return 0;
}
int main() {
  run<0, 2, 2, 256>(); run<0, 2, 2, 512>(); run<0, 2, 2, 1024>();
  run<0, 2, 4, 256>(); run<0, 2, 4, 512>(); run<0, 2, 4, 1024>();
  run<0, 4, 4, 256>(); run<0, 4, 4, 512>();
  run<1, 1, 1, 256>(); run<1, 1, 1, 512>(); run<1, 1, 1, 1024>();
  run<1, 1, 2, 256>(); run<1, 1, 2, 512>(); run<1, 1, 2, 1024>();
  run<1, 2, 2, 256>(); run<1, 2, 2, 512>();
  return 0;
}
