// capi.hip -- library info and the error channel of libsig3d_hip.so.
//
// The reference reports kernel failures with fprintf(stderr) + exit(-1)
// (lib/pointnet2/_ext_src/include/cuda_utils.h:30-39); a library must not kill its host
// process, so every entry point returns the hipError_t value and leaves a thread-local
// message here for the host language to turn into its own exception type.
#include <stdio.h>
#include <string.h>

#include "sig3d_common.h"

namespace {
thread_local char g_err[512] = "";
}

void sig3d_set_error(const char *where, hipError_t err) {
  snprintf(g_err, sizeof(g_err), "%s: %s (%d)", where, hipGetErrorString(err), (int)err);
}

void sig3d_set_error_msg(const char *where, const char *msg) {
  snprintf(g_err, sizeof(g_err), "%s: %s", where, msg);
}

extern "C" const char *sig3d_version(void) { return "sig3d-hip 0.1.0 gfx950"; }

extern "C" const char *sig3d_last_error(void) { return g_err; }

// ---- in-graph timestamps ----------------------------------------------------------------------
// A hipGraph replay has no per-node events, and a profiler serialises the branches it is asked to
// observe.  This one-lane kernel stores the constant-rate wall clock (100 MHz on gfx950) into a
// slot; captured between the launches of either branch it gives the true concurrent timeline of a
// replayed step (situation3d_amd/timeline.py, tools/branch_timeline.py).  Diagnostic only.
namespace {
__global__ void timestamp_kernel(unsigned long long *slot) { *slot = wall_clock64(); }
}  // namespace

extern "C" int sig3d_timestamp(unsigned long long *slot, void *stream_) {
  SIG3D_REQUIRE(slot != nullptr, "slot must not be null");
  hipLaunchKernelGGL(timestamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream_, slot);
  SIG3D_LAUNCH_CHECK("timestamp_kernel");
  return 0;
}

// ticks per second of the clock sig3d_timestamp stores
extern "C" int sig3d_timestamp_rate(int device, long long *hz) {
  SIG3D_REQUIRE(hz != nullptr, "hz must not be null");
  int khz = 0;
  SIG3D_HIP_TRY(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device));
  *hz = (long long)khz * 1000;
  return 0;
}

// ---- device-side stream handshake ---------------------------------------------------------------
// "Stream B starts after stream A reaches this point" is normally hipStreamWaitEvent: a barrier packet at the head
// of B's hardware queue.  While that barrier is BLOCKED the command processor polls it between the packets of every
// other queue: measured on this runtime, each kernel of the busy stream then costs ~1.7 us more
// (tools/probes/fork_penalty.py: a 450-node graph 5.85 -> 6.65 ms with nothing but a blocked barrier on a second
// stream).  A host that runs a step ahead keeps such a barrier blocked for milliseconds.  These two one-lane kernels
// move the wait onto a CU: A bumps a ticket counter, B's first kernel spins (agent-scope loads, s_sleep) until the
// ticket passes the count it has consumed.  Both are ordinary graph nodes.  The waiter gives up after `timeout_ticks`
// of the 100 MHz wall clock and raises *error (a crashed producer must not hang the device).
namespace {
__global__ void ticket_signal_kernel(unsigned int *ticket) {
  __threadfence();
  __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void ticket_wait_kernel(const unsigned int *ticket, unsigned int *consumed, unsigned long long timeout_ticks,
                                   int *error) {
  const unsigned int want = *consumed + 1u;
  const unsigned long long t0 = wall_clock64();
  // signed distance: tickets wrap after 2^32 steps
  while ((int)(__hip_atomic_load(ticket, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {
    __builtin_amdgcn_s_sleep(32);
    if (wall_clock64() - t0 > timeout_ticks) {
      *error = 1;
      break;
    }
  }
  *consumed = want;
  __threadfence();
}
}  // namespace

extern "C" int sig3d_ticket_signal(unsigned int *ticket, void *stream_) {
  SIG3D_REQUIRE(ticket != nullptr, "ticket must not be null");
  hipLaunchKernelGGL(ticket_signal_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream_, ticket);
  SIG3D_LAUNCH_CHECK("ticket_signal_kernel");
  return 0;
}

extern "C" int sig3d_ticket_wait(const unsigned int *ticket, unsigned int *consumed, long long timeout_us, int *error,
                                 void *stream_) {
  SIG3D_REQUIRE(ticket != nullptr && consumed != nullptr && error != nullptr, "null pointer");
  SIG3D_REQUIRE(timeout_us > 0, "timeout must be positive");
  hipLaunchKernelGGL(ticket_wait_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream_, ticket, consumed,
                     (unsigned long long)timeout_us * 100ull, error);
  SIG3D_LAUNCH_CHECK("ticket_wait_kernel");
  return 0;
}

// ---- CU-masked streams ---------------------------------------------------------------------------
// The geometry chain (cooperative FPS: 64 workgroups that sleep and poll for ~7 ms) shares the chip with the
// training step.  A step kernel's duration is that of its SLOWEST workgroup, so 64 CUs that also host an FPS
// workgroup stretch every chip-wide launch of the step.  A stream created with a CU mask confines the chain to a
// few CUs (and, optionally, the step to the others).  Mask bit k addresses XCD k % 8, CU slot k / 8 of that XCD
// (verified with sig3d_whereami, tools/probes/cu_mask_probe.py).
extern "C" int sig3d_stream_create_with_cu_mask(int words, const unsigned int *mask, void **stream) {
  SIG3D_REQUIRE(words > 0 && mask != nullptr && stream != nullptr, "bad arguments");
  hipStream_t s = nullptr;
  SIG3D_HIP_TRY(hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask));
  *stream = (void *)s;
  return 0;
}

extern "C" int sig3d_stream_destroy(void *stream) {
  if (stream) SIG3D_HIP_TRY(hipStreamDestroy((hipStream_t)stream));
  return 0;
}

namespace {
__global__ void queue_hold_kernel(unsigned long long hold_ticks) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < hold_ticks) __builtin_amdgcn_s_sleep(8);
}
}  // namespace

// One wave that sleeps for hold_us on `stream`.  Two of them on two streams take hold_us in total when the streams are
// served by different hardware queues and 2 x hold_us when HIP put them on one (streams.run_concurrently: the geometry
// pipeline draws streams until its chains run BESIDE the step).
extern "C" int sig3d_queue_hold(int hold_us, void *stream_) {
  SIG3D_REQUIRE(hold_us >= 0 && hold_us <= 100000, "hold_us out of range");
  hipLaunchKernelGGL(queue_hold_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream_, (unsigned long long)hold_us * 100ull);
  SIG3D_LAUNCH_CHECK("queue_hold_kernel");
  return 0;
}

namespace {
__global__ void whereami_kernel(unsigned int *slots, unsigned long long hold_ticks) {
  unsigned int hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  if (threadIdx.x == 0) {
    slots[2 * blockIdx.x + 0] = hw;
    slots[2 * blockIdx.x + 1] = xcc;
  }
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < hold_ticks) __builtin_amdgcn_s_sleep(8);   // keep the CU so the grid spreads out
}
}  // namespace

// Diagnostic: workgroup i of `blocks` x `threads` stores {HW_ID, XCC_ID} into slots[2i..2i+1] and holds its CU for
// hold_us microseconds.  HW_ID: cu_id bits 11:8, sh_id bit 12, se_id bits 15:13; XCC_ID: bits 3:0.
extern "C" int sig3d_whereami(unsigned int *slots, int blocks, int threads, int hold_us, void *stream_) {
  SIG3D_REQUIRE(slots != nullptr && blocks > 0 && threads > 0 && threads <= 1024, "bad arguments");
  hipLaunchKernelGGL(whereami_kernel, dim3(blocks), dim3(threads), 0, (hipStream_t)stream_, slots,
                     (unsigned long long)hold_us * 100ull);
  SIG3D_LAUNCH_CHECK("whereami_kernel");
  return 0;
}

// Diagnostic: `blocks` x `threads` workgroups that hold NV live VGPRs per lane and `lds` bytes of LDS while they
// sleep for hold_us -- what does a RESIDENT kernel of a given footprint cost the kernels of another stream?
// (tools/ab_step.py with SIG3D_PROBE_SPIN_US / SIG3D_PROBE_SPIN_SHAPE)
namespace {
template <int NV>
__global__ __launch_bounds__(NV > 128 ? 256 : 1024) void hold_kernel(float *sink, unsigned long long hold_ticks) {
  extern __shared__ float hold_lds[];
  float v[NV > 0 ? NV : 1];
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = (float)(threadIdx.x * (i + 1));
  if (threadIdx.x == 0) hold_lds[0] = 1.f;
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < hold_ticks) {
    __builtin_amdgcn_s_sleep(16);
#pragma unroll
    for (int i = 0; i < NV; ++i) asm volatile("" : "+v"(v[i]));   // keep every register live across the loop
  }
  float acc = hold_lds[0];
#pragma unroll
  for (int i = 0; i < NV; ++i) acc += v[i];
  if (acc == -1.f) sink[0] = acc;
}
}  // namespace

extern "C" int sig3d_hold(float *sink, int blocks, int threads, int hold_us, int vgprs, int lds_bytes, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SIG3D_REQUIRE(sink != nullptr && blocks > 0 && threads > 0 && threads <= 1024 && lds_bytes >= 4, "bad arguments");
  // (without the bound a 1024-thread launch caps a lane at 128 registers and the 220-register variant spills to scratch)
  SIG3D_REQUIRE(vgprs < 200 || threads <= 256, "more than 128 live registers per lane need workgroups of at most 256 threads");
  const unsigned long long ticks = (unsigned long long)hold_us * 100ull;
  if (lds_bytes > 48 * 1024) {
    SIG3D_HIP_TRY(hipFuncSetAttribute((const void *)hold_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    SIG3D_HIP_TRY(hipFuncSetAttribute((const void *)hold_kernel<100>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    SIG3D_HIP_TRY(hipFuncSetAttribute((const void *)hold_kernel<220>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
  }
  if (vgprs >= 200) hipLaunchKernelGGL(hold_kernel<220>, dim3(blocks), dim3(threads), lds_bytes, stream, sink, ticks);
  else if (vgprs >= 90) hipLaunchKernelGGL(hold_kernel<100>, dim3(blocks), dim3(threads), lds_bytes, stream, sink, ticks);
  else hipLaunchKernelGGL(hold_kernel<0>, dim3(blocks), dim3(threads), lds_bytes, stream, sink, ticks);
  SIG3D_LAUNCH_CHECK("hold_kernel");
  return 0;
}
