// situational_pose.h -- the pose arithmetic of the situational re-encode, shared by csrc/situational.hip and the
// positional MLP that consumes the re-encoded token positions (csrc/small_mlp.hip).
//   situation3d/utils/temp.py:42-97: pose (7) = [t, q_xyzw] -> p' = R(q) p + t, R spelled exactly as temp.py:63-73;
//   inverse: the agent-frame map R^T (p - t).  Every product and sum individually rounded in the order written there.
#pragma once
#include "sig3d_common.h"

namespace {

struct Pose {
  float r[3][3];
  float t[3];
};

// temp.py:49-78, every product and sum individually rounded in the order written there
__device__ __forceinline__ Pose load_pose(const float *__restrict__ q7) {
  Pose P;
  const float x = q7[3], y = q7[4], z = q7[5], w = q7[6];
  const float x2 = __fmul_rn(x, x), y2 = __fmul_rn(y, y), z2 = __fmul_rn(z, z), w2 = __fmul_rn(w, w);
  const float xy = __fmul_rn(x, y), zw = __fmul_rn(z, w), xz = __fmul_rn(x, z);
  const float yw = __fmul_rn(y, w), yz = __fmul_rn(y, z), xw = __fmul_rn(x, w);
  P.r[0][0] = __fadd_rn(__fsub_rn(__fsub_rn(x2, y2), z2), w2);
  P.r[1][0] = __fmul_rn(2.f, __fadd_rn(xy, zw));
  P.r[2][0] = __fmul_rn(2.f, __fsub_rn(xz, yw));
  P.r[0][1] = __fmul_rn(2.f, __fsub_rn(xy, zw));
  P.r[1][1] = __fadd_rn(__fsub_rn(__fadd_rn(-x2, y2), z2), w2);
  P.r[2][1] = __fmul_rn(2.f, __fadd_rn(yz, xw));
  P.r[0][2] = __fmul_rn(2.f, __fadd_rn(xz, yw));
  P.r[1][2] = __fmul_rn(2.f, __fsub_rn(yz, xw));
  P.r[2][2] = __fadd_rn(__fadd_rn(__fsub_rn(-x2, y2), z2), w2);
  P.t[0] = q7[0]; P.t[1] = q7[1]; P.t[2] = q7[2];
  return P;
}

template <bool INVERSE>
__device__ __forceinline__ void apply_pose(const Pose &P, float p0, float p1, float p2, float *o) {
  if (!INVERSE) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
      o[r] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(p0, P.r[r][0]), __fmul_rn(p1, P.r[r][1])),
                                 __fmul_rn(p2, P.r[r][2])),
                       P.t[r]);
  } else {
    const float q0 = __fsub_rn(p0, P.t[0]), q1 = __fsub_rn(p1, P.t[1]), q2 = __fsub_rn(p2, P.t[2]);
#pragma unroll
    for (int r = 0; r < 3; ++r)
      o[r] = __fadd_rn(__fadd_rn(__fmul_rn(q0, P.r[0][r]), __fmul_rn(q1, P.r[1][r])),
                       __fmul_rn(q2, P.r[2][r]));
  }
}

}  // namespace
