// compact.hip -- distinct neighbours of the ball-query lists.
//
// ball_query fills a list with the in-radius points in index order and pads a short list by repeating
// its FIRST hit (ball_query_gpu.cu:30-40): list = [i0 < i1 < ... < i(k-1), i0, i0, ...].  Every padded
// entry is an identical column of the grouped tensor, of every 1x1-conv / BatchNorm / ReLU layer above it
// and of the max-pool -- on sparse scenes most of the SharedMLP work of a set-abstraction level is such
// copies (tools/neighbour_stats.py).  This kernel lists the DISTINCT (centre, neighbour) pairs of a batch
// element back to back, with the multiplicity each one stands for; the SharedMLP kernels then run on the
// first n_act[b] positions only and weight statistics / gradient corrections by the multiplicity
// (shared_mlp.hip), which reproduces the dense result up to floating-point summation order.
#include "sig3d_common.h"

namespace {

constexpr int CP_THREADS = 1024;

// one workgroup per batch element; centres j are dealt to threads in contiguous runs
__global__ __launch_bounds__(CP_THREADS) void compact_lists_kernel(
    int m, int ns, const int *__restrict__ idx_all, int *__restrict__ cidx_all, int *__restrict__ ccent_all,
    float *__restrict__ mult_all, int *__restrict__ seg_all, int *__restrict__ n_act) {
  __shared__ int s_wave[CP_THREADS / 64];
  const int bi = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long total = (long)m * ns;
  const int *idx = idx_all + (size_t)bi * total;
  int *cidx = cidx_all + (size_t)bi * total;
  int *ccent = ccent_all + (size_t)bi * total;
  float *mult = mult_all + (size_t)bi * total;
  int *seg = seg_all + (size_t)bi * (m + 1);
  const int per = (m + CP_THREADS - 1) / CP_THREADS;
  const int j0 = min(tid * per, m), j1 = min(j0 + per, m);
  // pass 1: distinct count of my centres (the distinct entries are the leading ones)
  int mine = 0;
  for (int j = j0; j < j1; ++j) {
    const int *row = idx + (size_t)j * ns;
    const int first = row[0];
    int k = 1;
    for (int s = 1; s < ns; ++s) k += row[s] != first;
    mine += k;
  }
  int incl = mine;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int o = __shfl_up(incl, off);
    if (lane >= off) incl += o;
  }
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  int run = incl - mine;
  for (int w = 0; w < wave; ++w) run += s_wave[w];
  // pass 2: emit
  for (int j = j0; j < j1; ++j) {
    const int *row = idx + (size_t)j * ns;
    const int first = row[0];
    seg[j] = run;
    int k = 0;
    for (int s = 0; s < ns; ++s) {
      const int a = row[s];
      if (s == 0 || a != first) {
        cidx[run + k] = a;
        ccent[run + k] = j;
        mult[run + k] = 1.f;
        ++k;
      }
    }
    mult[run] = (float)(ns - k + 1);  // the first hit also stands for the padding
    run += k;
  }
  if (tid == CP_THREADS - 1) {
    seg[m] = run;
    n_act[bi] = run;
  }
}

}  // namespace

extern "C" int sig3d_compact_neighbour_lists(int b, int m, int nsample, const int *idx, int *cidx, int *ccent,
                                             float *mult, int *seg_off, int *n_act, void *stream_) {
  SIG3D_REQUIRE(b >= 0 && m >= 0 && nsample >= 1, "bad size");
  SIG3D_REQUIRE((long)m * nsample < (1L << 31), "m * nsample too large");
  if (b == 0) return 0;
  hipLaunchKernelGGL(compact_lists_kernel, dim3(b), dim3(CP_THREADS), 0, (hipStream_t)stream_, m, nsample, idx, cidx,
                     ccent, mult, seg_off, n_act);
  SIG3D_LAUNCH_CHECK("compact_lists_kernel");
  return 0;
}
