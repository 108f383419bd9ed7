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
