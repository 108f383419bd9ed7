// tools/micro/mfma16_fill.hip -- can ONE wave hide an LDS read behind its own f32 MFMA?  (not product code)
// Stream: [MFMA x G, ds_read_b128 x 1] repeated, 1 or 2 waves per SIMD (256- or 512-thread workgroups, one per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int G, int NT>
__global__ __launch_bounds__(NT) void k(float *out, unsigned long long *ticks, int iters) {
  __shared__ f32x4 lds[1024];
  lds[threadIdx.x] = f32x4{1.f, 2.f, 3.f, 4.f};
  lds[threadIdx.x + 512] = f32x4{1.f, 2.f, 3.f, 4.f};
  __syncthreads();
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  f32x4 f = lds[threadIdx.x], g = lds[threadIdx.x ^ 1], p1 = lds[threadIdx.x ^ 2], p2 = lds[threadIdx.x ^ 3];
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int m = 0; m < G; ++m) {
        acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(f[m & 3], g[(m + r) & 3], acc[m & 3], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      {   // the value read here is consumed two groups later: no wait on a fresh read
        const f32x4 n = lds[(threadIdx.x + 64 * r + it) & 1023];
        g = p1; p1 = p2; p2 = n;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int n = 0; n < 4; ++n) for (int r = 0; r < 4; ++r) s += acc[n][r];
  out[blockIdx.x * NT + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

template <int G, int NT>
int run() {
  float *out; unsigned long long *ticks, h;
  CHECK(hipMalloc(&out, 256 * NT * 4)); CHECK(hipMalloc(&ticks, 8));
  const int iters = 500;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<G, NT>), dim3(256), dim3(NT), 0, 0, out, ticks, iters);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<G, NT>), dim3(256), dim3(NT), 0, 0, out, ticks, iters);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  CHECK(hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost));
  const double n = (double)iters * 8 * G;
  printf("%d MFMA per ds_read_b128, %d waves / SIMD: %6.1f ticks per MFMA of one wave, %6.1f TFLOP/s\n", G, NT / 256, h / n,
         n * 2048.0 * (NT / 64) * 256 / (ms * 1e-3) * 1e-12);
  return 0;
}
int main() {
  run<1, 256>(); run<2, 256>(); run<4, 256>(); run<8, 256>();
  run<1, 512>(); run<2, 512>(); run<4, 512>(); run<8, 512>();
  run<2, 1024>(); run<4, 1024>();
  return 0;
}
