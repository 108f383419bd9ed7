// rowops.hip -- small row/column helpers of the Q-Former's dense layers for gfx950.
//
// column_sum: bias gradient of an nn.Linear, db[c] = sum_r dY[r][c]
// (the backward of `self.dense(hidden_states)` etc., Qformer.py:242,311,324; torch computes it
// with a generic reduce kernel that takes ~12 us for a 416 x 768 input on MI355X -- 122 launches
// per Q-Former forward+backward).  The op is pure latency (1.3 MB): a workgroup owns 64
// consecutive columns (every row read is one coalesced 256-byte segment per wave) and spreads
// the rows over SIXTEEN waves with eight independent loads in flight per lane, partials meet in
// LDS: no atomics, deterministic summation order.
#include "sig3d_common.h"

namespace {

constexpr int CS_WAVES = 16;

__global__ __launch_bounds__(CS_WAVES * 64) void column_sum_kernel(int rows, int cols,
                                                                   const float *__restrict__ x,
                                                                   float *__restrict__ out) {
  __shared__ float part[CS_WAVES][64];
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  if (c < cols) {
    int r = wave;
    for (; r + 7 * CS_WAVES < rows; r += 8 * CS_WAVES) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] += x[(size_t)(r + i * CS_WAVES) * cols + c];
    }
    for (; r < rows; r += CS_WAVES) acc[0] += x[(size_t)r * cols + c];
  }
  part[wave][lane] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  __syncthreads();
  if (wave == 0 && c < cols) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < CS_WAVES; ++w) s += part[w][lane];
    out[c] = s;
  }
}

}  // namespace

extern "C" int sig3d_column_sum(int rows, int cols, const float *x, float *out, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SIG3D_REQUIRE(rows >= 0 && cols >= 0, "negative size");
  if (cols == 0) return 0;
  hipLaunchKernelGGL(column_sum_kernel, dim3(sig3d_ceil_div(cols, 64)), dim3(CS_WAVES * 64), 0, stream,
                     rows, cols, x, out);
  SIG3D_LAUNCH_CHECK("column_sum_kernel");
  return 0;
}
