// tools/micro/mfma16_rate.hip -- what one wave per SIMD gets out of v_mfma_f32_16x16x4_f32 (not product code).
//   hipcc --offload-arch=gfx950 -O3 -o mfma16_rate mfma16_rate.hip && ./mfma16_rate
// Pure streams of NACC independent accumulators, operands in registers; s_memtime ticks per MFMA and wall time per
// MFMA for grids of 64 / 256 / 512 workgroups (the clock the part holds depends on how much of it is busy).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int NACC, int KIND>   // KIND 0: 16x16x4, 1: 32x32x2
__global__ __launch_bounds__(256) void rate(float *out, unsigned long long *ticks, int iters) {
  float a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = threadIdx.x * 0.001f + i; b[i] = threadIdx.x * 0.002f - i; }
  f32x4 acc4[NACC];
  f32x16 acc16[NACC];
  for (int n = 0; n < NACC; ++n) { acc4[n] = f32x4{0, 0, 0, 0}; for (int r = 0; r < 16; ++r) acc16[n][r] = 0; }
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int n = 0; n < NACC; ++n) {
        if (KIND == 0) acc4[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[(j + n) & 3], acc4[n], 0, 0, 0);
        else acc16[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[(j + n) & 3], acc16[n], 0, 0, 0);
      }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int n = 0; n < NACC; ++n) { for (int r = 0; r < 4; ++r) s += acc4[n][r]; for (int r = 0; r < 16; ++r) s += acc16[n][r]; }
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

template <int NACC, int KIND>
int run(const char *name, int wgs) {
  float *out; unsigned long long *ticks, h;
  CHECK(hipMalloc(&out, 1024 * 256 * 4)); CHECK(hipMalloc(&ticks, 8));
  const int iters = 2000;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((rate<NACC, KIND>), dim3(wgs), dim3(256), 0, 0, out, ticks, iters);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((rate<NACC, KIND>), dim3(wgs), dim3(256), 0, 0, out, ticks, iters);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  CHECK(hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost));
  const double n = (double)iters * 4 * NACC;
  const double flop = n * (KIND == 0 ? 2048.0 : 4096.0) * 4 * wgs;
  printf("%-28s wgs %4d: %6.1f ticks / MFMA, %6.2f ns / MFMA (tick = %.2f GHz), %6.1f TFLOP/s\n", name, wgs, h / n,
         ms * 1e6 / n, h / (ms * 1e6), flop / (ms * 1e-3) * 1e-12);
  CHECK(hipFree(out)); CHECK(hipFree(ticks));
  return 0;
}

int main() {
  for (int wgs : {64, 256, 512}) {
    run<4, 0>("16x16x4, 4 accumulators", wgs);
    run<6, 0>("16x16x4, 6 accumulators", wgs);
    run<2, 0>("16x16x4, 2 accumulators", wgs);
    run<1, 0>("16x16x4, 1 accumulator", wgs);
    run<2, 1>("32x32x2, 2 accumulators", wgs);
    run<1, 1>("32x32x2, 1 accumulator", wgs);
  }
  return 0;
}
