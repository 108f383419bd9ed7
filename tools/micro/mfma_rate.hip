// Ground truth for v_mfma_f32_32x32x2_f32 on gfx950: cycles per instruction per SIMD at 1..4 waves
// per SIMD, with and without an LDS operand read per MFMA.   hipcc --offload-arch=gfx950 -O3 mfma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <bool LDS>
__global__ __launch_bounds__(256) void k(float *out, int iters, unsigned long long *cyc) {
  __shared__ float s[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) s[i] = 1.0f / (i + 1);
  __syncthreads();
  f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
  float x = threadIdx.x * 1e-3f;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    float b0 = x, b1 = x, b2 = x, b3 = x;
    if (LDS) { const int o = (i * 4 + (threadIdx.x & 63) * 65) & 4095; b0 = s[o]; b1 = s[(o + 1) & 4095]; b2 = s[(o + 2) & 4095]; b3 = s[(o + 3) & 4095]; }
    a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, b0, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, b1, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, b2, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, b3, a3, 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float r = 0;
  for (int j = 0; j < 16; ++j) r += a0[j] + a1[j] + a2[j] + a3[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
  float *out; unsigned long long *cyc, h;
  hipMalloc(&out, 256 * 4096 * 4); hipMalloc(&cyc, 8);
  const int iters = 4096;
  for (int lds = 0; lds < 2; ++lds)
    for (int wg_per_cu = 1; wg_per_cu <= 4; ++wg_per_cu) {   // 256 threads = 1 wave per SIMD per workgroup
      hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a);
        if (lds) hipLaunchKernelGGL(k<true>, dim3(256 * wg_per_cu), dim3(256), 0, 0, out, iters, cyc);
        else hipLaunchKernelGGL(k<false>, dim3(256 * wg_per_cu), dim3(256), 0, 0, out, iters, cyc);
        hipEventRecord(b); hipEventSynchronize(b);
      }
      float ms; hipEventElapsedTime(&ms, a, b);
      hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
      const double mfma_per_simd = 4.0 * iters * wg_per_cu;
      printf("lds=%d waves/SIMD=%d: %.1f us, %.1f shader cycles per MFMA per SIMD (wave 0: %.1f cycles per own MFMA), %.1f TFLOP/s\n",
             lds, wg_per_cu, ms * 1e3, (double)h / (4.0 * iters) / wg_per_cu, (double)h / (4.0 * iters),
             mfma_per_simd * 1024 * 4096 / (ms * 1e-3) / 1e12);
    }
  return 0;
}
