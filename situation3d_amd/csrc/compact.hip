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

// The same lists with coalesced reads: a row of NS entries is read by NS consecutive lanes (64 / NS rows per wave
// and load), the distinct entries of a row are its leading ones plus nothing else (a padded entry repeats the FIRST
// hit), so "keep" = (s == 0 || entry != first) and a row's output positions are the popcount of the keep bits below
// a lane.  Pass 1 counts per row into LDS, one workgroup-wide scan gives every row its offset, pass 2 emits.
// (The thread-per-row kernel above walks 256-byte rows with a 256-byte stride between lanes: 200 us at SA1, at the
// END of the geometry chain -- where every microsecond is one the next step may have to wait for.)
constexpr int CW_MAX_M = 8192;

template <int NS>
__global__ __launch_bounds__(CP_THREADS) void compact_lists_wave_kernel(
    int m, const int *__restrict__ idx_all, int *__restrict__ cidx_all, int *__restrict__ ccent_all,
    float *__restrict__ mult_all, int *__restrict__ seg_all, int *__restrict__ n_act) {
  constexpr int RPW = 64 / NS;                 // rows per wave and trip
  constexpr int NWAVES = CP_THREADS / 64;
  __shared__ int s_cnt[CW_MAX_M];              // per row: distinct count, then exclusive offset
  __shared__ int s_wave[NWAVES];
  const int bi = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long total = (long)m * NS;
  const int *idx = idx_all + (size_t)bi * total;
  int *cidx = cidx_all + (size_t)bi * total;
  int *ccent = ccent_all + (size_t)bi * total;
  float *mult = mult_all + (size_t)bi * total;
  int *seg = seg_all + (size_t)bi * (m + 1);
  const int sub = lane / NS, s_in = lane % NS;  // which row of the trip, which entry of the row
  const unsigned long long rowmask = (NS == 64 ? ~0ull : ((1ull << NS) - 1ull)) << (sub * NS);
  // pass 1: distinct count per row.  UNR trips per iteration with all their loads issued first (clamped row, masked
  // afterwards): a trip is one dependent global round trip, 128 of them in a row were the kernel's whole time
  constexpr int UNR = 4;
  for (int j0 = wave * RPW; j0 < m; j0 += UNR * NWAVES * RPW) {
    int a[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int j = min(j0 + u * NWAVES * RPW + sub, m - 1);
      a[u] = idx[(size_t)j * NS + s_in];
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int j = j0 + u * NWAVES * RPW + sub;
      const bool live = j < m;
      const int first = __shfl(a[u], sub * NS, 64);
      const bool keep = live && (s_in == 0 || a[u] != first);
      const unsigned long long bits = __builtin_amdgcn_ballot_w64(keep) & rowmask;
      if (live && s_in == 0) s_cnt[j] = __builtin_popcountll(bits);
    }
  }
  __syncthreads();
  // exclusive scan of s_cnt[0..m): thread t owns a contiguous run of rows
  const int per = (m + CP_THREADS - 1) / CP_THREADS;
  const int r0 = min(tid * per, m), r1 = min(r0 + per, m);
  int mine = 0;
  for (int j = r0; j < r1; ++j) mine += s_cnt[j];
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
  for (int j = r0; j < r1; ++j) {
    const int k = s_cnt[j];
    s_cnt[j] = run;
    seg[j] = run;
    run += k;
  }
  if (tid == CP_THREADS - 1) {
    seg[m] = run;
    n_act[bi] = run;
  }
  __syncthreads();
  // pass 2: emit
  for (int j0 = wave * RPW; j0 < m; j0 += UNR * NWAVES * RPW) {
    int a[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int j = min(j0 + u * NWAVES * RPW + sub, m - 1);
      a[u] = idx[(size_t)j * NS + s_in];
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int j = j0 + u * NWAVES * RPW + sub;
      const bool live = j < m;
      const int first = __shfl(a[u], sub * NS, 64);
      const bool keep = live && (s_in == 0 || a[u] != first);
      const unsigned long long bits = __builtin_amdgcn_ballot_w64(keep) & rowmask;
      if (keep) {
        const int pos = s_cnt[j] + __builtin_popcountll(bits & ((1ull << lane) - 1ull));
        cidx[pos] = a[u];
        ccent[pos] = j;
        // the first hit also stands for the padding
        mult[pos] = s_in == 0 ? (float)(NS - __builtin_popcountll(bits) + 1) : 1.f;
      }
    }
  }
}

}  // namespace

extern "C" int sig3d_compact_neighbour_lists(int b, int m, int nsample, const int *idx, int *cidx, int *ccent,
                                             float *mult, int *seg_off, int *n_act, void *stream_) {
  SIG3D_REQUIRE(b >= 0 && m >= 0 && nsample >= 1, "bad size");
  SIG3D_REQUIRE((long)m * nsample < (1L << 31), "m * nsample too large");
  if (b == 0) return 0;
  hipStream_t stream = (hipStream_t)stream_;
  if (m <= CW_MAX_M && (nsample == 64 || nsample == 32 || nsample == 16 || nsample == 8)) {
    if (nsample == 64)
      hipLaunchKernelGGL(compact_lists_wave_kernel<64>, dim3(b), dim3(CP_THREADS), 0, stream, m, idx, cidx, ccent, mult, seg_off, n_act);
    else if (nsample == 32)
      hipLaunchKernelGGL(compact_lists_wave_kernel<32>, dim3(b), dim3(CP_THREADS), 0, stream, m, idx, cidx, ccent, mult, seg_off, n_act);
    else if (nsample == 16)
      hipLaunchKernelGGL(compact_lists_wave_kernel<16>, dim3(b), dim3(CP_THREADS), 0, stream, m, idx, cidx, ccent, mult, seg_off, n_act);
    else
      hipLaunchKernelGGL(compact_lists_wave_kernel<8>, dim3(b), dim3(CP_THREADS), 0, stream, m, idx, cidx, ccent, mult, seg_off, n_act);
    SIG3D_LAUNCH_CHECK("compact_lists_wave_kernel");
    return 0;
  }
  hipLaunchKernelGGL(compact_lists_kernel, dim3(b), dim3(CP_THREADS), 0, stream, m, nsample, idx, cidx,
                     ccent, mult, seg_off, n_act);
  SIG3D_LAUNCH_CHECK("compact_lists_kernel");
  return 0;
}
