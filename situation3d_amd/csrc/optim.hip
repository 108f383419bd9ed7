// optim.hip -- value clip + AdamW over flat parameter storage for gfx950.
//
// Replaces, for the training step of the reference's Solver (lib/solver.py:618-627):
//   nn.utils.clip_grad_value_(model.parameters(), clip_value)   # one clamp pass over all grads
//   optimizer.step()                                            # torch.optim.AdamW (train.py:226-238)
//   optimizer.zero_grad()                                       # next iteration (solver.py:620)
// torch runs this as ~36 multi-tensor launches (pointer tables are limited by kernel-argument
// size) that reach ~1.8 TB/s; parameters, gradients and both moments live here in FLAT buffers
// (one per parameter group), so the whole update is ONE streaming kernel per group with 16-byte
// accesses: read p, g, m, v -- write p, m, v (28 B/element, HBM-bound; sig3d_adamw_flat can also write
// zeros back to g, the table kernel never does: its gradients are dropped by the host afterwards).
// The step counter lives on the device (hipGraph replay must not bake it into the graph).
//
// Update rule == torch.optim.AdamW (amsgrad=False, maximize=False):
//   g = clamp(g, -clip, clip);  p *= 1 - lr*wd;  m = b1*m + (1-b1)*g;  v = b2*v + (1-b2)*g*g
//   p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
#include "sig3d_common.h"

namespace {

__global__ void step_increment_kernel(float *step) { *step += 1.f; }

__device__ __forceinline__ void adamw_one(float &p, float &g, float &m, float &v, float lr, float b1,
                                          float b2, float eps, float wd, float clip, float step_size,
                                          float inv_sqrt_bc2) {
  float gg = g;
  if (clip > 0.f) gg = fminf(fmaxf(gg, -clip), clip);
  p = p * (1.f - lr * wd);
  m = b1 * m + (1.f - b1) * gg;
  v = b2 * v + (1.f - b2) * gg * gg;
  const float denom = sqrtf(v) * inv_sqrt_bc2 + eps;
  p = p - step_size * (m / denom);
}

__global__ __launch_bounds__(256) void adamw_flat_kernel(long n, float *__restrict__ p,
                                                         float *__restrict__ g, float *__restrict__ m,
                                                         float *__restrict__ v,
                                                         const float *__restrict__ step, float lr,
                                                         const float *__restrict__ lr_device,
                                                         float b1, float b2, float eps, float wd,
                                                         float clip, int zero_grad) {
  if (lr_device) lr = *lr_device;  // a scheduler's value, read at execution time (hipGraph replays)
  const float t = *step;
  const float bc1 = 1.f - powf(b1, t), bc2 = 1.f - powf(b2, t);
  const float step_size = lr / bc1, inv_sqrt_bc2 = 1.f / sqrtf(bc2);
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    float4 pp = reinterpret_cast<float4 *>(p)[i], gg = reinterpret_cast<float4 *>(g)[i];
    float4 mm = reinterpret_cast<float4 *>(m)[i], vv = reinterpret_cast<float4 *>(v)[i];
    adamw_one(pp.x, gg.x, mm.x, vv.x, lr, b1, b2, eps, wd, clip, step_size, inv_sqrt_bc2);
    adamw_one(pp.y, gg.y, mm.y, vv.y, lr, b1, b2, eps, wd, clip, step_size, inv_sqrt_bc2);
    adamw_one(pp.z, gg.z, mm.z, vv.z, lr, b1, b2, eps, wd, clip, step_size, inv_sqrt_bc2);
    adamw_one(pp.w, gg.w, mm.w, vv.w, lr, b1, b2, eps, wd, clip, step_size, inv_sqrt_bc2);
    reinterpret_cast<float4 *>(p)[i] = pp;
    reinterpret_cast<float4 *>(m)[i] = mm;
    reinterpret_cast<float4 *>(v)[i] = vv;
    if (zero_grad) reinterpret_cast<float4 *>(g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {  // tail (flat buffers are padded, so normally empty)
    const long i = (n4 << 2) + threadIdx.x;
    float pp = p[i], gg = g[i], mm = m[i], vv = v[i];
    adamw_one(pp, gg, mm, vv, lr, b1, b2, eps, wd, clip, step_size, inv_sqrt_bc2);
    p[i] = pp; m[i] = mm; v[i] = vv;
    if (zero_grad) g[i] = 0.f;
  }
}

// ---- pointer-table variants ------------------------------------------------------------------
// Autograd hands every parameter a freshly allocated gradient tensor (when .grad is None it just
// adopts the tensor: no kernel).  Keeping .grad as a view of a flat buffer instead costs one tiny
// accumulate-add launch PER PARAMETER (+315 launches per step for this model), so the gradients
// stay where autograd put them and the kernels below walk a device-resident table of chunks
// {p, g, m, v, n, weight_decay}: one workgroup per chunk of <= 65536 elements, ONE launch for the
// whole model.  Under hipGraph replay all addresses are static, so the table is built once.
struct OptChunk {
  float *p;
  float *g;
  float *m;
  float *v;
  long long n;
  float wd;
  float pad;
};

__device__ __forceinline__ void adamw_chunk(const OptChunk c, float lr, float b1, float b2, float eps, float clip,
                                            float step_size, float inv_sqrt_bc2) {
  if (c.n <= 0) return;            // a parameter without a gradient this step: untouched (torch skips it)
  const bool vec = ((((size_t)c.p | (size_t)c.g | (size_t)c.m | (size_t)c.v) & 15) == 0);
  const long n4 = vec ? (c.n >> 2) : 0;
  // streaming update: nothing here is re-read before it falls out of the 256 MB Infinity Cache
  // (428 MB of parameters alone), so loads and stores are non-temporal; two float4 quadruples in
  // flight per lane (8 x 16 B loads) to cover HBM latency with ~6 workgroups per CU
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  f32x4 *p4 = reinterpret_cast<f32x4 *>(c.p), *m4 = reinterpret_cast<f32x4 *>(c.m);
  f32x4 *v4 = reinterpret_cast<f32x4 *>(c.v);
  const f32x4 *g4 = reinterpret_cast<const f32x4 *>(c.g);
  long i = threadIdx.x;
  for (; i + 256 < n4; i += 512) {
    f32x4 pp[2], gg[2], mm[2], vv[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      pp[u] = __builtin_nontemporal_load(p4 + i + 256 * u);
      gg[u] = __builtin_nontemporal_load(g4 + i + 256 * u);
      mm[u] = __builtin_nontemporal_load(m4 + i + 256 * u);
      vv[u] = __builtin_nontemporal_load(v4 + i + 256 * u);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float pe = pp[u][e], ge = gg[u][e], me = mm[u][e], ve = vv[u][e];
        adamw_one(pe, ge, me, ve, lr, b1, b2, eps, c.wd, clip, step_size, inv_sqrt_bc2);
        pp[u][e] = pe; mm[u][e] = me; vv[u][e] = ve;
      }
      __builtin_nontemporal_store(pp[u], p4 + i + 256 * u);
      __builtin_nontemporal_store(mm[u], m4 + i + 256 * u);
      __builtin_nontemporal_store(vv[u], v4 + i + 256 * u);
    }
  }
  for (; i < n4; i += 256) {
    f32x4 pp = p4[i], gg = g4[i], mm = m4[i], vv = v4[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float pe = pp[e], ge = gg[e], me = mm[e], ve = vv[e];
      adamw_one(pe, ge, me, ve, lr, b1, b2, eps, c.wd, clip, step_size, inv_sqrt_bc2);
      pp[e] = pe; mm[e] = me; vv[e] = ve;
    }
    p4[i] = pp; m4[i] = mm; v4[i] = vv;
  }
  for (i = (n4 << 2) + threadIdx.x; i < c.n; i += 256) {
    float pp = c.p[i], gg = c.g[i], mm = c.m[i], vv = c.v[i];
    adamw_one(pp, gg, mm, vv, lr, b1, b2, eps, c.wd, clip, step_size, inv_sqrt_bc2);
    c.p[i] = pp; c.m[i] = mm; c.v[i] = vv;
  }
}

// gridDim.x == nchunks: one workgroup per chunk (the whole chip streams).  gridDim.x < nchunks: a bounded set of
// workgroups walks the table (an update that shares the chip with another stream's kernels leaves them CU slots).
__global__ __launch_bounds__(256) void adamw_table_kernel(const OptChunk *__restrict__ table, int nchunks,
                                                          const float *__restrict__ step, float lr,
                                                          const float *__restrict__ lr_device,
                                                          float b1, float b2, float eps, float clip) {
  if (lr_device) lr = *lr_device;  // a scheduler's value, read at execution time (hipGraph replays)
  const float t = *step;
  const float bc1 = 1.f - powf(b1, t), bc2 = 1.f - powf(b2, t);
  const float step_size = lr / bc1, inv_sqrt_bc2 = 1.f / sqrtf(bc2);
  for (int k = blockIdx.x; k < nchunks; k += gridDim.x)
    adamw_chunk(table[k], lr, b1, b2, eps, clip, step_size, inv_sqrt_bc2);
}

// dst (chunk.m field reused as destination) <- src (chunk.g): gather scattered gradients into
// flat storage for the data-parallel all-reduce
__global__ __launch_bounds__(256) void gather_table_kernel(const OptChunk *__restrict__ table) {
  const OptChunk c = table[blockIdx.x];
  const bool vec = ((((size_t)c.g | (size_t)c.m) & 15) == 0);
  const long n4 = vec ? (c.n >> 2) : 0;
  for (long i = threadIdx.x; i < n4; i += 256)
    reinterpret_cast<float4 *>(c.m)[i] = reinterpret_cast<const float4 *>(c.g)[i];
  for (long i = (n4 << 2) + threadIdx.x; i < c.n; i += 256) c.m[i] = c.g[i];
}

}  // namespace

extern "C" int sig3d_adamw_table(int nchunks, const void *table, const float *step, float lr,
                                 const float *lr_device, float beta1, float beta2, float eps,
                                 float clip_value, void *stream_) {
  SIG3D_REQUIRE(nchunks >= 0, "negative size");
  if (nchunks == 0) return 0;
  hipLaunchKernelGGL(adamw_table_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream_,
                     (const OptChunk *)table, nchunks, step, lr, lr_device, beta1, beta2, eps, clip_value);
  SIG3D_LAUNCH_CHECK("adamw_table_kernel");
  return 0;
}

extern "C" int sig3d_adamw_table_bounded(int nchunks, const void *table, const float *step, float lr,
                                         const float *lr_device, float beta1, float beta2, float eps,
                                         float clip_value, int max_workgroups, void *stream_) {
  SIG3D_REQUIRE(nchunks >= 0 && max_workgroups >= 1, "bad size");
  if (nchunks == 0) return 0;
  const int grid = nchunks < max_workgroups ? nchunks : max_workgroups;
  hipLaunchKernelGGL(adamw_table_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream_,
                     (const OptChunk *)table, nchunks, step, lr, lr_device, beta1, beta2, eps, clip_value);
  SIG3D_LAUNCH_CHECK("adamw_table_kernel");
  return 0;
}

extern "C" int sig3d_gather_table(int nchunks, const void *table, void *stream_) {
  SIG3D_REQUIRE(nchunks >= 0, "negative size");
  if (nchunks == 0) return 0;
  hipLaunchKernelGGL(gather_table_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream_,
                     (const OptChunk *)table);
  SIG3D_LAUNCH_CHECK("gather_table_kernel");
  return 0;
}

extern "C" int sig3d_step_increment(float *step, void *stream_) {
  hipLaunchKernelGGL(step_increment_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream_, step);
  SIG3D_LAUNCH_CHECK("step_increment_kernel");
  return 0;
}

extern "C" int sig3d_adamw_flat(long n, float *p, float *g, float *m, float *v, const float *step,
                                float lr, const float *lr_device, float beta1, float beta2, float eps,
                                float weight_decay, float clip_value, int zero_grad, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SIG3D_REQUIRE(n >= 0, "negative size");
  if (n == 0) return 0;
  long blocks = (n / 4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(adamw_flat_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, n, p, g, m, v, step,
                     lr, lr_device, beta1, beta2, eps, weight_decay, clip_value, zero_grad);
  SIG3D_LAUNCH_CHECK("adamw_flat_kernel");
  return 0;
}
