// tools/micro/tr16_probe.hip -- what ds_read_b64_tr_b16 returns (gfx950), printed lane by lane (not product code).
//   hipcc --offload-arch=gfx950 -O3 -o tr16_probe tr16_probe.hip && ./tr16_probe
// LDS holds lds[i] = i (16-bit elements).  Pattern 0: lane l reads at element 4 l (8 bytes per lane, lane-linear).
// Pattern 1: the operand pattern of csrc/gemmp_core.h: a 16-lane group reads a [4 rows][16 columns] block of a
// row-major image with 16 columns (32 bytes) per row: lane i at row (i >> 2), column 4 (i & 3).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(int pattern, unsigned short *dst) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const int l = threadIdx.x, i = l & 15, g = l >> 4;
  int elem = pattern == 0 ? 4 * l : g * 1024 + (i >> 2) * 16 + 4 * (i & 3);
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(lds + elem));
  for (int e = 0; e < 4; ++e) dst[l * 4 + e] = (unsigned short)v[e];
}
int main() {
  unsigned short *d, h[256];
  hipMalloc(&d, sizeof(h));
  for (int pattern = 0; pattern < 2; ++pattern) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, pattern, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("pattern %d\n", pattern);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3]);
  }
  return 0;
}
