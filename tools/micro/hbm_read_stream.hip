// Read-only HBM streaming ceiling of one MI355X (`torch.sum` over 2 GiB is a library reduction,
// not a ceiling).  Grid-stride float4 loads, U independent loads in flight per lane, default and non-temporal,
// several grid sizes; a row-segment variant reads 256-byte / 3072-byte segments out of 9216-float rows -- the access
// pattern of the key/value stream of the config-5 cross-attention (one head's 64 floats vs all 12 heads' 768 floats
// of a token row).      hipcc --offload-arch=gfx950 -O3 hbm_read_stream.hip -o hbm_read_stream && ./hbm_read_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { if ((x) != hipSuccess) { printf("HIP error at %s\n", #x); return 1; } } while (0)

template <int U, bool NT>
__global__ __launch_bounds__(256) void stream_kernel(const f4 *__restrict__ x, size_t n4, float *out) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  float acc = 0.f;
  for (; i + (U - 1) * stride < n4; i += U * stride) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
      v[u] = NT ? __builtin_nontemporal_load(x + i + u * stride) : x[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
  }
  for (; i < n4; i += stride) { const f4 v = x[i]; acc += v.x + v.y + v.z + v.w; }
  if (acc == 123.456f) out[0] = acc;   // never true: keeps the loads
}

// rows of `ld` floats; every workgroup reads `seg` floats (seg * 4 bytes, contiguous) at column offset col0 of ROWS
// consecutive rows per step: seg = 64 is one head's slice of a token row, seg = 768 all twelve heads.
template <int U>
__global__ __launch_bounds__(256) void segment_kernel(const float *__restrict__ x, long rows, int ld, int seg, int nseg,
                                                      float *out) {
  const int seg4 = seg / 4;                       // float4 per segment row
  const int rows_per_step = 256 / seg4 > 0 ? 256 / seg4 : 1;
  const int which = blockIdx.x % nseg;            // which column segment (head) this workgroup streams
  const long wg = blockIdx.x / nseg, nwg = gridDim.x / nseg;
  const int r_in = threadIdx.x / seg4, c4 = threadIdx.x % seg4;
  float acc = 0.f;
  if (seg4 <= 256) {
    for (long r0 = wg * rows_per_step * U; r0 < rows; r0 += nwg * rows_per_step * U) {
      float4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long r = r0 + u * rows_per_step + r_in;
        v[u] = r < rows ? *reinterpret_cast<const float4 *>(x + r * ld + which * seg + 4 * c4) : make_float4(0, 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
  }
  if (acc == 123.456f) out[0] = acc;
}

template <typename F>
static double time_ms(F launch, int reps = 10) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  launch(); launch();
  (void)hipEventRecord(a);
  for (int i = 0; i < reps; ++i) launch();
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms = 0.f; (void)hipEventElapsedTime(&ms, a, b);
  return ms / reps;
}

int main() {
  const size_t bytes = (size_t)4 << 30;   // 4 GiB: far beyond the 256 MB of Infinity Cache
  float *x, *out;
  CK(hipMalloc(&x, bytes)); CK(hipMalloc(&out, 64));
  CK(hipMemset(x, 0, bytes));
  const size_t n4 = bytes / 16;
  printf("read-only stream of %.1f GiB (grid-stride float4, U loads in flight per lane)\n", bytes / 1073741824.0);
  for (int wgs : {1024, 2048, 4096, 8192, 16384}) {
    double t;
    t = time_ms([&] { hipLaunchKernelGGL((stream_kernel<4, false>), dim3(wgs), dim3(256), 0, 0, (const f4 *)x, n4, out); });
    printf("  wgs %5d  U=4      %.3f ms  %.2f TB/s", wgs, t, bytes / t / 1e9);
    t = time_ms([&] { hipLaunchKernelGGL((stream_kernel<8, false>), dim3(wgs), dim3(256), 0, 0, (const f4 *)x, n4, out); });
    printf("   U=8 %.2f TB/s", bytes / t / 1e9);
    t = time_ms([&] { hipLaunchKernelGGL((stream_kernel<8, true>), dim3(wgs), dim3(256), 0, 0, (const f4 *)x, n4, out); });
    printf("   U=8 nt %.2f TB/s", bytes / t / 1e9);
    t = time_ms([&] { hipLaunchKernelGGL((stream_kernel<16, true>), dim3(wgs), dim3(256), 0, 0, (const f4 *)x, n4, out); });
    printf("   U=16 nt %.2f TB/s\n", bytes / t / 1e9);
  }
  // the K/V stream of the config-5 cross-attention: B x 80 000 token rows of 9216 floats (six cross layers stacked) or
  // 1536 floats (one layer's [K | V]); one layer's K (or V) half = 768 floats = 12 heads x 64
  for (int ld : {1536, 9216}) {
    const long rows = (long)(bytes / 4 / ld);
    for (int seg : {64, 768}) {
      const int nseg = 768 / seg;     // read exactly one 768-float half of every row in total
      for (int wgs : {1536, 6144}) {
        const int g = wgs / nseg * nseg;
        const double t = time_ms([&] { hipLaunchKernelGGL((segment_kernel<8>), dim3(g), dim3(256), 0, 0, x, rows, ld, seg, nseg, out); });
        const double moved = (double)rows * 768 * 4;
        printf("  rows of %4d floats, %3d-float segments (%4d B), %5d wgs: %.3f ms  %.2f TB/s of useful bytes\n", ld, seg,
               seg * 4, g, t, moved / t / 1e9);
      }
    }
  }
  return 0;
}
