// pos_embed.hip -- 3-axis sinusoidal position embedding gather + add for gfx950.
//
// Replaces the Python double loop of Blip2T5.forward
// (3DLLM_BLIP2-base/lavis/models/blip2_models/blip2_t5.py:106-118):
//   all_pcs = zeros_like(pc_embeds)            # allocated on the CPU, then .cuda()
//   for j in batch: all_pcs[j][:, :1407] = cat([pos_embedding[pc[j][:, i].long()] for i in 0..2], -1)
//   pc_embeds = pc_embeds + 0.01 * all_pcs
// i.e. out[b,n,ch] = feat[b,n,ch] + scale * table[(long)pc[b,n,ch / tw]][ch % tw]  for ch < 3*tw,
//      out[b,n,ch] = feat[b,n,ch]                                              otherwise.
// One streaming pass over the (B, N, C) feature tensor (8 B/element of HBM traffic; the 469x256
// table stays in L2) instead of a host-side tensor build + H2D copy of B*N*C floats every step.
// The sum is evaluated as feat + (scale * t), both individually rounded, like the reference.
#include "sig3d_common.h"

namespace {

__global__ __launch_bounds__(256) void pos_embed_add_kernel(long rows, int c, int tw, int trows,
                                                            float scale, const float *__restrict__ feat,
                                                            const float *__restrict__ pc,
                                                            const float *__restrict__ table,
                                                            float *__restrict__ out) {
  // one workgroup sweeps whole rows: lanes run along the channel axis (coalesced)
  for (long r = blockIdx.x; r < rows; r += gridDim.x) {
    const float *p = pc + r * 3;
    int idx[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      long v = (long)p[a];  // .long(): truncation toward zero (blip2_t5.py:107)
      idx[a] = (int)(v < 0 ? 0 : (v >= trows ? trows - 1 : v));
    }
    const float *f = feat + r * c;
    float *o = out + r * c;
    for (int ch = threadIdx.x; ch < c; ch += 256) {
      float v = f[ch];
      const int axis = ch / tw;
      if (axis < 3) v = __fadd_rn(v, __fmul_rn(scale, table[(size_t)idx[axis] * tw + (ch - axis * tw)]));
      o[ch] = v;
    }
  }
}

}  // namespace

extern "C" int sig3d_pos_embed_add(int b, int n, int c, int tw, int trows, float scale,
                                   const float *feat, const float *pc, const float *table,
                                   float *out, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SIG3D_REQUIRE(b >= 0 && n >= 0 && c >= 1 && tw >= 1 && trows >= 1, "bad size");
  const long rows = (long)b * n;
  if (rows == 0) return 0;
  const unsigned grid = (unsigned)(rows < 8192 ? rows : 8192);
  hipLaunchKernelGGL(pos_embed_add_kernel, dim3(grid), dim3(256), 0, stream, rows, c, tw, trows, scale,
                     feat, pc, table, out);
  SIG3D_LAUNCH_CHECK("pos_embed_add_kernel");
  return 0;
}
