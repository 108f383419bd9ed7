// mlp16.h -- arguments of the SharedMLP layer kernel on the 16 x 16 x 4 f32 matrix instruction (csrc/mlp16.hip), shared
// with csrc/shared_mlp.hip, whose entry points route to it.  Internal to libsig3d_hip.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>

struct Mlp16Args {
  int b, cin, cout;              // batch; reduction length (3 + C for a gathering first layer); output channels
  long E;                        // positions per (batch, channel) row: the row stride of x and y
  const float *x;                // (b, cin, E) activations, or dY for an input-gradient product; unused when gathering
  const float *w;                // (cout, cin) weights; w_t: the stored forward weight (cin_of_product rows... see below)
  const float *pscale, *pshift;  // BatchNorm + ReLU of the previous layer on operand load, or null
  float *y;                      // (b, cout, E)
  double *stat_sum, *stat_sq;    // per-channel sum / sum of squares of y (weighted by mult), or null
  const int *n_act;              // compact lists: positions per batch element, or null (E)
  const float *mult;             // compact lists: multiplicity of a position (statistics), or null
  int w_t;                       // 1: w is (cin, cout) as stored by the forward layer -- the product is y = w^T x
  // gathering first layer (MlpGather of shared_mlp.hip), gather != 0
  int gather;
  const float *g_xyz, *g_centre, *g_feat;
  const int *g_idx, *g_centre_of;
  int gN, gP, gS, gC, g_normalize;
  float g_radius;
};

// whether the kernel serves this problem (shapes, alignment); false: the caller keeps its own kernel
bool sig3d_mlp16_applies(const Mlp16Args &a);
// launches the product; 0 or a HIP error (sig3d_set_error has been called)
int sig3d_mlp16_launch(const Mlp16Args &a, hipStream_t stream);
