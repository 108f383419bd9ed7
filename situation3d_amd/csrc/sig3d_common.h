// sig3d_common.h -- shared device/host helpers for the gfx950 kernels of libsig3d_hip.so.
// CDNA4 only: 64-wide wavefronts, DPP row operations, exact (non-contracted) f32 helpers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "../../include/sig3d_hip.h"
#include "../../include/sig3d_debug.h"

// ---- host-side error plumbing (capi.hip owns the thread-local buffer) -------------------
void sig3d_set_error(const char *where, hipError_t err);
void sig3d_set_error_msg(const char *where, const char *msg);

#define SIG3D_HIP_TRY(expr)                         \
  do {                                              \
    hipError_t _e = (expr);                         \
    if (_e != hipSuccess) {                         \
      sig3d_set_error(#expr, _e);                   \
      return (int)_e;                               \
    }                                               \
  } while (0)

#define SIG3D_LAUNCH_CHECK(name)                    \
  do {                                              \
    hipError_t _e = hipGetLastError();              \
    if (_e != hipSuccess) {                         \
      sig3d_set_error(name, _e);                    \
      return (int)_e;                               \
    }                                               \
  } while (0)

#define SIG3D_REQUIRE(cond, msg)                    \
  do {                                              \
    if (!(cond)) {                                  \
      sig3d_set_error_msg(__func__, msg);           \
      return (int)hipErrorInvalidValue;             \
    }                                               \
  } while (0)

static inline int sig3d_ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of a kernel ON ONE DEVICE: a process that drives a second GPU
// must set it there too.  One of these per call site: `pending()` is true until `done()` was called for the CURRENT device
// (a bit per device id; two threads racing set the attribute twice, which is harmless).
struct sig3d_once_per_device {
  std::atomic<uint64_t> mask{0};
  static uint64_t current() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) d = 0;
    return 1ull << (d & 63);
  }
  bool pending() const { return !(mask.load(std::memory_order_acquire) & current()); }
  void done() { mask.fetch_or(current(), std::memory_order_release); }
};

// ---- exact f32 arithmetic ----------------------------------------------------------------
// The parity contract (oracle/pointnet2_oracle.c header) is "every * and +/- individually
// rounded, in source order".  hipcc defaults to -ffp-contract=fast, so distances are spelled
// with the _rn intrinsics, which the compiler may not fuse.
__device__ __forceinline__ float sq_dist3(float ax, float ay, float az, float bx, float by,
                                          float bz) {
  const float dx = __fsub_rn(ax, bx), dy = __fsub_rn(ay, by), dz = __fsub_rn(az, bz);
  return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

// ---- DPP row (16-lane) all-reduce ---------------------------------------------------------
// quad_perm[1,0,3,2]=0xB1, quad_perm[2,3,0,1]=0x4E, row_half_mirror=0x141, row_mirror=0x140.
// After the four steps every lane of a 16-lane row holds the row's reduction.
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) {
  return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false);
}

__device__ __forceinline__ int row_allreduce_max_i32(int v) {
  v = max(v, dpp_i32<0xB1>(v));
  v = max(v, dpp_i32<0x4E>(v));
  v = max(v, dpp_i32<0x141>(v));
  v = max(v, dpp_i32<0x140>(v));
  return v;
}
__device__ __forceinline__ unsigned row_allreduce_min_u32(unsigned v) {
  v = min(v, (unsigned)dpp_i32<0xB1>((int)v));
  v = min(v, (unsigned)dpp_i32<0x4E>((int)v));
  v = min(v, (unsigned)dpp_i32<0x141>((int)v));
  v = min(v, (unsigned)dpp_i32<0x140>((int)v));
  return v;
}
__device__ __forceinline__ float row_allreduce_sum_f32(float v) {
  v += __builtin_bit_cast(float, dpp_i32<0xB1>(__builtin_bit_cast(int, v)));
  v += __builtin_bit_cast(float, dpp_i32<0x4E>(__builtin_bit_cast(int, v)));
  v += __builtin_bit_cast(float, dpp_i32<0x141>(__builtin_bit_cast(int, v)));
  v += __builtin_bit_cast(float, dpp_i32<0x140>(__builtin_bit_cast(int, v)));
  return v;
}
// Wave (64-lane) all-reduce: row reduce, then combine the four rows through SGPRs.
__device__ __forceinline__ int wave_allreduce_max_i32(int v) {
  v = row_allreduce_max_i32(v);
  const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
  const int c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
  return max(max(a, b), max(c, d));
}
__device__ __forceinline__ unsigned wave_allreduce_min_u32(unsigned v) {
  v = row_allreduce_min_u32(v);
  const unsigned a = __builtin_amdgcn_readlane((int)v, 0), b = __builtin_amdgcn_readlane((int)v, 16);
  const unsigned c = __builtin_amdgcn_readlane((int)v, 32), d = __builtin_amdgcn_readlane((int)v, 48);
  return min(min(a, b), min(c, d));
}
__device__ __forceinline__ float wave_allreduce_sum_f32(float v) {
  v = row_allreduce_sum_f32(v);
  const int iv = __builtin_bit_cast(int, v);
  const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0));
  const float b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16));
  const float c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32));
  const float d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
  return (a + b) + (c + d);
}

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }
