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
