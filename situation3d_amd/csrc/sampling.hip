// sampling.hip -- furthest point sampling, gather_points and its gradient for gfx950.
//
// Replaces lib/pointnet2/_ext_src/src/sampling_gpu.cu of the reference (behaviour only; the
// decomposition below is MI355X-specific).
//
// FPS design.  The reference re-reads the whole scene (xyz + temp, 20 B/point) from global
// memory in every one of the m-1 dependent rounds and reduces through nine __syncthreads
// steps.  Here a scene lives in the REGISTERS of one workgroup for the whole run (PPT points
// per thread: x, y, z and the running min-distance), a round is one register sweep, two DPP
// wave reductions and ONE barrier, and HBM is touched once at setup.  Scenes larger than the
// register file keep their overflow tail in global memory (L2-resident).
//
// Bit-exactness.  The reference's winner is the point of maximal min-distance where ties are
// resolved by its thread/tree decomposition (sampling_gpu.cu:95-168): a thread scans
// k = tid, tid+bs, ... with a strict '>' (lowest k of its stride class wins), and every tree
// step keeps the lower slot on ties, the LAST step (slot 0 vs 1) being the most significant.
// That is a total order: maximise d2, then minimise key(k) = (bitrev_L(k mod bs), k div bs),
// bs = opt_n_threads(n) (cuda_utils.h:13-19), L = log2(bs).  Because it is a total order, any
// reduction shape that maximises (d2, -key) returns the reference's index; the key is packed
// as (bitrev << 22) | (k >> L).  Points with x^2+y^2+z^2 <= 1e-3 (double compare, :100-101)
// never take part: their running distance is pinned to -1, which also makes the all-skipped
// case return index 0 as the reference does (best = -1, besti = 0).
#include "sig3d_common.h"

#include <type_traits>

#ifndef SIG3D_FPS_PROBE
#define SIG3D_FPS_PROBE 0   // measurement builds only: 1 = the cooperative kernel sweeps half of a thread's points per round (wrong results); 2 = it claims 256 VGPRs a lane (same results); 3 = a round's sample coordinates are made up instead of loaded (wrong results; 3.49 -> 3.34 ms alone: the dependent load is 0.08 of a 1.72 us round)
#endif

namespace {

__device__ __forceinline__ unsigned fps_key(unsigned k, int L, unsigned bsmask) {
  const unsigned hi = L ? (__brev(k & bsmask) >> (32 - L)) : 0u;
  return (hi << 22) | (k >> L);
}
__device__ __forceinline__ unsigned fps_unkey(unsigned key, int L) {
  const unsigned hi = key >> 22, lo = key & 0x3FFFFFu;
  const unsigned low = L ? (__brev(hi) >> (32 - L)) : 0u;
  return (lo << L) | low;
}

// NT threads per workgroup (multiple of the reference block size bs, so that ascending
// register slot == ascending key inside a thread), PPT register-resident points per thread.
template <int NT, int PPT>
__global__ __launch_bounds__(NT) void fps_kernel(int n, int m, int L,
                                                 const float *__restrict__ dataset_all,
                                                 float *__restrict__ temp_all,
                                                 int *__restrict__ idxs_all,
                                                 const int *__restrict__ prefix_ok) {
  constexpr int NW = NT / 64;
  __shared__ int s_val[2][NW];
  __shared__ unsigned s_key[2][NW];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (prefix_ok != nullptr && prefix_ok[blockIdx.x] != 0) {
    // fps_prefix_check_kernel proved that every round of this scene picks the next point in
    // storage order: the result is 0..m-1 and the m dependent rounds need not run
    for (int j = tid; j < m; j += NT) idxs_all[(size_t)blockIdx.x * m + j] = j;
    return;
  }
  const float *dataset = dataset_all + (size_t)blockIdx.x * n * 3;
  float *temp = temp_all + (size_t)blockIdx.x * n;
  int *idxs = idxs_all + (size_t)blockIdx.x * m;
  const unsigned bsmask = (1u << L) - 1u;

  float px[PPT], py[PPT], pz[PPT], pt[PPT];
#pragma unroll
  for (int s = 0; s < PPT; ++s) {
    const int k = tid + NT * s;
    float x = 0.f, y = 0.f, z = 0.f, t = -1.f;
    if (k < n) {
      x = dataset[3 * k + 0];
      y = dataset[3 * k + 1];
      z = dataset[3 * k + 2];
      const float mag = __fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z));
      t = ((double)mag <= 1e-3) ? -1.f : 1e10f;  // sampling_gpu.cu:100-101, sampling.cpp:74-76
    }
    px[s] = x; py[s] = y; pz[s] = z; pt[s] = t;
  }
  // overflow tail: running distance lives in `temp` (global), same -1 pinning
  for (int k = NT * PPT + tid; k < n; k += NT) {
    const float x = dataset[3 * k + 0], y = dataset[3 * k + 1], z = dataset[3 * k + 2];
    const float mag = __fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z));
    temp[k] = ((double)mag <= 1e-3) ? -1.f : 1e10f;
  }

  int old = 0;
  if (tid == 0) idxs[0] = 0;

  for (int j = 1; j < m; ++j) {
    const float x1 = dataset[3 * old + 0], y1 = dataset[3 * old + 1], z1 = dataset[3 * old + 2];
    float best = -1.f;
    int bslot = 0;
#pragma unroll
    for (int s = 0; s < PPT; ++s) {
      const float d = sq_dist3(px[s], py[s], pz[s], x1, y1, z1);
      const float t = fminf(d, pt[s]);
      pt[s] = t;
      const bool gt = t > best;  // strict: lowest slot (== lowest key) of this thread wins
      best = gt ? t : best;
      bslot = gt ? s : bslot;
    }
    int bk = tid + NT * bslot;
    for (int k = NT * PPT + tid; k < n; k += NT) {
      const float d = sq_dist3(dataset[3 * k + 0], dataset[3 * k + 1], dataset[3 * k + 2], x1, y1, z1);
      const float t = fminf(d, temp[k]);
      temp[k] = t;
      const bool gt = t > best;
      best = gt ? t : best;
      bk = gt ? k : bk;
    }
    // (best >= 0) or -1: both order correctly as signed integers
    const int myv = __builtin_bit_cast(int, best);
    const int wv = wave_allreduce_max_i32(myv);
    const unsigned mykey = (myv == wv) ? fps_key((unsigned)bk, L, bsmask) : 0xFFFFFFFFu;
    const unsigned wk = wave_allreduce_min_u32(mykey);
    const int par = j & 1;
    if (lane == 0) {
      s_val[par][wave] = wv;
      s_key[par][wave] = wk;
    }
    __syncthreads();
    const int ov = s_val[par][lane & (NW - 1)];
    const unsigned ok = s_key[par][lane & (NW - 1)];
    const int gv = row_allreduce_max_i32(ov);
    const unsigned gk = row_allreduce_min_u32(ov == gv ? ok : 0xFFFFFFFFu);
    // every lane of every wave now holds the same (gv, gk); make that provable
    const int gvu = __builtin_amdgcn_readfirstlane(gv);
    const unsigned gku = (unsigned)__builtin_amdgcn_readfirstlane((int)gk);
    old = (gvu < 0) ? 0 : (int)fps_unkey(gku, L);
    if (tid == 0) idxs[j] = old;
  }
}

template <int NT, int PPT>
int launch_fps(int b, int n, int m, int L, const float *dataset, float *temp, int *idxs,
               hipStream_t stream, const int *prefix_ok = nullptr) {
  hipLaunchKernelGGL((fps_kernel<NT, PPT>), dim3(b), dim3(NT), 0, stream, n, m, L, dataset, temp,
                     idxs, prefix_ok);
  SIG3D_LAUNCH_CHECK("fps_kernel");
  return 0;
}

// ---- FPS over an FPS-ordered cloud: prove the answer instead of computing it ------------------
// The input of SA level l+1 is the output of the FPS of level l, stored in pick order.  FPS is
// greedy, so FPS over the first n' picks of an FPS run reproduces that run: unless two candidates
// tie exactly, round j picks point j and the result is 0,1,...,m-1.  Running the m strictly
// dependent rounds (0.6 us each, one workgroup per scene) to find that out costs 1.1 ms per step
// at B = 8; CHECKING it is embarrassingly parallel:
//   r[j]   = running distance of point j when round j is decided
//          = min(1e10, min_{i<j} d(x_j, x_i))            (-1 when x_j is a skipped point)
//   round j picks j  <=>  for every k != j:  t_j(k) < r[j],  or  t_j(k) == r[j] and key(k) > key(j)
// with t_j(k) the same running minimum for point k and key() the reference's tie order
// (fps_key).  Same sq_dist3, same fminf chain (min is exact, so its order is free), same skip
// rule as fps_kernel: the check decides exactly what fps_kernel would have computed, and any
// scene that fails it (ties, duplicates, zero padding, an input that is not FPS-ordered at all)
// runs the ordinary kernel.  Bit-exact either way.
__global__ __launch_bounds__(256) void fps_prefix_radius_kernel(int n, int m, int nflags,
                                                                const float *__restrict__ dataset_all,
                                                                float *__restrict__ r_all,
                                                                int *__restrict__ ok_all) {
  const float *dataset = dataset_all + (size_t)blockIdx.y * n * 3;
  const int lane = threadIdx.x & 63;
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);  // one wave per round
  if (blockIdx.x == 0 && threadIdx.x < (unsigned)nflags) ok_all[threadIdx.x * gridDim.y + blockIdx.y] = 1;
  if (j >= m) return;
  const float x = dataset[3 * j + 0], y = dataset[3 * j + 1], z = dataset[3 * j + 2];
  const float mag = __fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z));
  float t = ((double)mag <= 1e-3) ? -1.f : 1e10f;
  for (int i = lane; i < j; i += 64)
    t = fminf(sq_dist3(x, y, z, dataset[3 * i + 0], dataset[3 * i + 1], dataset[3 * i + 2]), t);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) t = fminf(t, __shfl_xor(t, off, 64));
  if (lane == 0) r_all[(size_t)blockIdx.y * n + j] = t;
}

// blockIdx.z = level of a CHAIN of nested samplings over one FPS-ordered array (sig3d_fps_nested_chain): level l
// samples m[l] of the first n[l] points (n[l+1] = m[l]); all levels read the same array (stride n[0]) and the same
// r[], and a violation at level l also fails every deeper level (whose input would no longer be this prefix).
struct FpsChain {
  int levels, stride;
  int n[4], m[4], L[4];
};

__global__ __launch_bounds__(256) void fps_prefix_check_kernel(FpsChain ch,
                                                               const float *__restrict__ dataset_all,
                                                               const float *__restrict__ r_all,
                                                               int *__restrict__ ok_all) {
  extern __shared__ float s_prefix[];  // x,y,z,r of the first m points
  const int level = blockIdx.z, n = ch.n[level], m = ch.m[level], L = ch.L[level], nb = gridDim.y;
  if ((int)(blockIdx.x * 256) >= n) return;
  const float *dataset = dataset_all + (size_t)blockIdx.y * ch.stride * 3;
  const float *r = r_all + (size_t)blockIdx.y * ch.stride;
  for (int i = threadIdx.x; i < m; i += 256) {
    s_prefix[4 * i + 0] = dataset[3 * i + 0];
    s_prefix[4 * i + 1] = dataset[3 * i + 1];
    s_prefix[4 * i + 2] = dataset[3 * i + 2];
    s_prefix[4 * i + 3] = r[i];
  }
  __syncthreads();
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= n) return;
  const unsigned bsmask = (1u << L) - 1u;
  const float x = dataset[3 * k + 0], y = dataset[3 * k + 1], z = dataset[3 * k + 2];
  const float mag = __fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z));
  float t = ((double)mag <= 1e-3) ? -1.f : 1e10f;
  const unsigned mykey = fps_key((unsigned)k, L, bsmask);
  bool bad = false;
  // one wave per SIMD (n * b threads in all): latency is hidden by unrolling, not by occupancy -- eight
  // broadcast LDS reads in flight per trip; the tie order is only evaluated on an exact tie (rare branch)
  constexpr int U = 8;
  int i = 0;
  for (; i + U < m; i += U) {
    float4 p[U];
    float rj[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      p[u] = *reinterpret_cast<const float4 *>(&s_prefix[4 * (i + u)]);
      rj[u] = s_prefix[4 * (i + u + 1) + 3];  // round j = i + u + 1 must pick point j
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      t = fminf(sq_dist3(x, y, z, p[u].x, p[u].y, p[u].z), t);
      if (t >= rj[u] && k != i + u + 1)
        bad |= (t > rj[u]) || mykey < fps_key((unsigned)(i + u + 1), L, bsmask);
    }
    // a wave that has seen a violation is done: on an input that is NOT FPS-ordered almost every wave leaves
    // after the first trip (half of all points beat point 1 in round 1), so calling the nested entry point
    // on arbitrary clouds costs little more than the plain one
    if (__builtin_amdgcn_ballot_w64(bad) != 0) {
      if (bad)
        for (int l = level; l < ch.levels; ++l) ok_all[l * nb + blockIdx.y] = 0;
      return;
    }
  }
  for (; i + 1 < m; ++i) {
    const float4 p = *reinterpret_cast<const float4 *>(&s_prefix[4 * i]);
    t = fminf(sq_dist3(x, y, z, p.x, p.y, p.z), t);
    const float rj = s_prefix[4 * (i + 1) + 3];
    if (t >= rj && k != i + 1) bad |= (t > rj) || mykey < fps_key((unsigned)(i + 1), L, bsmask);
  }
  if (bad)
    for (int l = level; l < ch.levels; ++l) ok_all[l * nb + blockIdx.y] = 0;
}

// The same check with the m - 1 rounds of a point split over S threads.  The running minimum t_j(k) is a prefix
// minimum over i < j, and min is exact, so a segment [a, b) of the rounds can be checked on its own once the minimum
// P over all rounds before `a` is known: t >= r  <=>  P >= r and (local running minimum) >= r.  Pass 1: every
// (point, segment) thread takes the minimum over its segment (no checks); pass 2: P = min of the earlier segments'
// results (LDS), then the original loop over the segment starting from P.  Twice the distances, 1 / S of the
// dependent chain: 1024 rounds x 6 instructions were 87 us at the end of the geometry chain.
template <int S>
__global__ __launch_bounds__(256) void fps_prefix_check_seg_kernel(FpsChain ch,
                                                                   const float *__restrict__ dataset_all,
                                                                   const float *__restrict__ r_all,
                                                                   int *__restrict__ ok_all) {
  constexpr int PTS = 256 / S;
  extern __shared__ float s_prefix[];  // x,y,z,r of the first m points, then S * PTS segment minima
  const int level = blockIdx.z, n = ch.n[level], m = ch.m[level], L = ch.L[level], nb = gridDim.y;
  if ((int)(blockIdx.x * PTS) >= n) return;
  float *s_min = s_prefix + 4 * (size_t)ch.m[0];
  const float *dataset = dataset_all + (size_t)blockIdx.y * ch.stride * 3;
  const float *r = r_all + (size_t)blockIdx.y * ch.stride;
  for (int i = threadIdx.x; i < m; i += 256) {
    s_prefix[4 * i + 0] = dataset[3 * i + 0];
    s_prefix[4 * i + 1] = dataset[3 * i + 1];
    s_prefix[4 * i + 2] = dataset[3 * i + 2];
    s_prefix[4 * i + 3] = r[i];
  }
  const int seg = threadIdx.x / PTS, kl = threadIdx.x % PTS;
  const int k_raw = blockIdx.x * PTS + kl;
  const bool valid = k_raw < n;
  const int k = valid ? k_raw : n - 1;
  const int len = m - 1;                                   // rounds i = 0 .. m - 2 (round j = i + 1)
  const int seglen = (len + S - 1) / S;
  const int a = min(seg * seglen, len), b = min(a + seglen, len);
  const unsigned bsmask = (1u << L) - 1u;
  const float x = dataset[3 * k + 0], y = dataset[3 * k + 1], z = dataset[3 * k + 2];
  const float mag = __fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z));
  const float t0 = ((double)mag <= 1e-3) ? -1.f : 1e10f;
  __syncthreads();
  constexpr int U = 8;
  {  // pass 1: the minimum over my segment
    float mn = 3.0e38f;
    int i = a;
    for (; i + U <= b; i += U) {
      float4 p[U];
#pragma unroll
      for (int u = 0; u < U; ++u) p[u] = *reinterpret_cast<const float4 *>(&s_prefix[4 * (i + u)]);
#pragma unroll
      for (int u = 0; u < U; ++u) mn = fminf(sq_dist3(x, y, z, p[u].x, p[u].y, p[u].z), mn);
    }
    for (; i < b; ++i) {
      const float4 p = *reinterpret_cast<const float4 *>(&s_prefix[4 * i]);
      mn = fminf(sq_dist3(x, y, z, p.x, p.y, p.z), mn);
    }
    s_min[seg * PTS + kl] = mn;
  }
  __syncthreads();
  float t = t0;
  for (int e = 0; e < seg; ++e) t = fminf(s_min[e * PTS + kl], t);
  const unsigned mykey = fps_key((unsigned)k, L, bsmask);
  bool bad = false;
  int i = a;
  for (; i + U <= b; i += U) {
    float4 p[U];
    float rj[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      p[u] = *reinterpret_cast<const float4 *>(&s_prefix[4 * (i + u)]);
      rj[u] = s_prefix[4 * (i + u + 1) + 3];  // round j = i + u + 1 must pick point j
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      t = fminf(sq_dist3(x, y, z, p[u].x, p[u].y, p[u].z), t);
      if (t >= rj[u] && k != i + u + 1)
        bad |= (t > rj[u]) || mykey < fps_key((unsigned)(i + u + 1), L, bsmask);
    }
    if (__builtin_amdgcn_ballot_w64(bad && valid) != 0) break;   // (no barrier follows: leaving early is safe)
  }
  for (; i < b && !bad; ++i) {
    const float4 p = *reinterpret_cast<const float4 *>(&s_prefix[4 * i]);
    t = fminf(sq_dist3(x, y, z, p.x, p.y, p.z), t);
    const float rj = s_prefix[4 * (i + 1) + 3];
    if (t >= rj && k != i + 1) bad |= (t > rj) || mykey < fps_key((unsigned)(i + 1), L, bsmask);
  }
  if (bad && valid)
    for (int l = level; l < ch.levels; ++l) ok_all[l * nb + blockIdx.y] = 0;
}

// ---- cooperative FPS: one scene spread over W workgroups ------------------------------------
// A 40 000-point scene does not fit the register file of ONE workgroup (16 waves x 128 VGPRs
// hold ~24 points per thread), and a global-memory overflow tail costs ~5 us per round.  Here W
// workgroups each keep n/W points in registers and exchange their per-round candidate
// (distance, key) through 8-byte granules in global memory -- ONE {round:16, key:16, value:32} granule per
// workgroup and round up to 65 536 points, a {round, value} / {round, key} pair above:
//   - a granule is ONE aligned 8-byte agent-scope relaxed store (sc1, write-through), so the
//     data is its own flag -- no fence, no separate flag word, no torn reads;
//   - consumers poll with agent-scope relaxed loads (L1 bypass) from one wave only;
//   - slots are double-buffered by round parity: a workgroup can be at most one round ahead of
//     its peers (to publish round j+1 it must have consumed every peer's round j), so a slot is
//     never overwritten before all peers have read it;
//   - all polled words are zeroed by a memset node on the same stream before EVERY launch;
//   - the result never depends on which XCD / CU a workgroup lands on.  Block ids are arranged so
//     that (with b a multiple of 8) the W workgroups of a scene share an XCD -- speed only.
// All W*b workgroups must be co-resident (W*b <= 256 CUs is checked by the launcher); every spin
// is bounded and a timeout poisons the output with -1 instead of hanging the GPU.
typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;

constexpr int FPS_SLOT_U64 = 128;  // per scene: 2 parities x W <= 32 workgroups x {val,key}

// scenes whose cooperative FPS gave up waiting for a peer workgroup (see the poison path below)
__device__ unsigned g_fps_timeouts = 0;

// BLOCKED: a wave owns a spatially compact block of the scene (`order`: the scene's point numbers along a Morton curve,
// fps_morton_order_kernel) instead of a stride of storage order, and sits a round out when the new sample is farther
// from its block's bounding box than its largest running distance: no point of the block can change then, and its
// candidate of the round before stands.  Same results (the tie key travels with each point, and a thread keeps its
// points in key order); the sweep is the VALU work the kernel takes from whatever runs beside it.
template <int NT, int PPT, int W, bool PACK, bool BLOCKED>
__global__ __launch_bounds__(NT) void fps_coop_kernel(int b, int n, int m, int L,
                                                      const float *__restrict__ dataset_all,
                                                      u64 *__restrict__ slots_all,
                                                      int *__restrict__ idxs_all,
                                                      const unsigned short *__restrict__ order_all) {
  constexpr int NW = NT / 64;
  __shared__ int s_val[2][NW];
  __shared__ unsigned s_key[2][NW];
  __shared__ int s_win[2][2];
  // BLOCKED: the tie keys of a thread's points live here, not in 20 registers -- what the kernel holds of a SIMD's
  // register file is what library kernels beside it cannot use (claiming 256 VGPRs instead of 150 cost the step 0.12 ms)
  __shared__ unsigned s_pkey[BLOCKED ? PPT : 1][BLOCKED ? NT : 1];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int scene = blockIdx.x % b, w = blockIdx.x / b;
#if SIG3D_FPS_PROBE == 2
  asm volatile("v_mov_b32 v255, 0" ::: "v255");
#endif
  const float *dataset = dataset_all + (size_t)scene * n * 3;
  int *idxs = idxs_all + (size_t)scene * m;
  gu64 *slots = (gu64 *)(slots_all + (size_t)scene * FPS_SLOT_U64);
  const unsigned bsmask = (1u << L) - 1u;

  // thread's points: k = r + 512*(s*G + grp) with g = w*NT + tid the thread's number in its scene, r = g mod 512,
  // grp = g div 512, G = NT*W/512 (NT = 512: k = tid + NT*(s*W + w)).  All points of a thread share the reference's
  // thread slot r = k mod 512, so ascending s == ascending tie key inside a thread (see fps_kernel) for ANY NT
  static_assert((NT * W) % 512 == 0, "a scene's threads must tile the reference block size");
  constexpr int G = NT * W / 512;
  const int g_id = w * NT + tid, r_slot = g_id & 511, grp = g_id >> 9;
  float px[PPT], py[PPT], pz[PPT], pt[PPT];
  if (BLOCKED) {
    unsigned pkey[PPT];
    // wave number wb of the scene's W * NW takes positions [wb * per, (wb + 1) * per) of the Morton order.  A thread's
    // strict `>` sweep picks its lowest slot among equal distances, so it keeps its points in key order: the keys are
    // sorted (odd-even transposition, a min and a max per exchange) before the coordinates are fetched
    const int per = (n + W * NW - 1) / (W * NW);
    const unsigned short *order = order_all + (size_t)scene * n;
#pragma unroll
    for (int s = 0; s < PPT; ++s) {
      const int in_block = s * 64 + lane, pos = (w * NW + wave) * per + in_block;
      pkey[s] = (in_block < per && pos < n) ? fps_key((unsigned)order[pos], L, bsmask) : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int pass = 0; pass < PPT; ++pass) {
#pragma unroll
      for (int s = pass & 1; s + 1 < PPT; s += 2) {
        const unsigned ka = pkey[s], kb = pkey[s + 1];
        pkey[s] = min(ka, kb); pkey[s + 1] = max(ka, kb);
      }
    }
#pragma unroll
    for (int s = 0; s < PPT; ++s) s_pkey[s][tid] = pkey[s];      // (read back by this thread only: no barrier)
  }
#pragma unroll
  for (int s = 0; s < PPT; ++s) {
    int k = r_slot + 512 * (s * G + grp);
    if (BLOCKED) {
      const unsigned key = s_pkey[s][tid];
      k = key == 0xFFFFFFFFu ? n : (int)fps_unkey(key, L);
    }
    float x = 0.f, y = 0.f, z = 0.f, t = -1.f;
    if (k < n) {
      x = dataset[3 * k + 0];
      y = dataset[3 * k + 1];
      z = dataset[3 * k + 2];
      const float mag = __fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z));
      t = ((double)mag <= 1e-3) ? -1.f : 1e10f;
    }
    px[s] = x; py[s] = y; pz[s] = z; pt[s] = t;
  }
  float lo_x = 0.f, lo_y = 0.f, lo_z = 0.f, hi_x = 0.f, hi_y = 0.f, hi_z = 0.f;
  if (BLOCKED) {
    // the block's bounding box over the points that take part (the others never change)
    float ax = 3.0e38f, ay = 3.0e38f, az = 3.0e38f, bx = -3.0e38f, by = -3.0e38f, bz = -3.0e38f;
#pragma unroll
    for (int s = 0; s < PPT; ++s)
      if (pt[s] >= 0.f) {
        ax = fminf(ax, px[s]); ay = fminf(ay, py[s]); az = fminf(az, pz[s]);
        bx = fmaxf(bx, px[s]); by = fmaxf(by, py[s]); bz = fmaxf(bz, pz[s]);
      }
    for (int o = 32; o >= 1; o >>= 1) {
      ax = fminf(ax, __shfl_xor(ax, o)); ay = fminf(ay, __shfl_xor(ay, o)); az = fminf(az, __shfl_xor(az, o));
      bx = fmaxf(bx, __shfl_xor(bx, o)); by = fmaxf(by, __shfl_xor(by, o)); bz = fmaxf(bz, __shfl_xor(bz, o));
    }
    auto uniform = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); };
    lo_x = uniform(ax); lo_y = uniform(ay); lo_z = uniform(az); hi_x = uniform(bx); hi_y = uniform(by); hi_z = uniform(bz);
  }
  int held_v = 0x7F800000;   // the wave's candidate of its last sweep (+inf: the first round always sweeps)
  unsigned held_k = 0xFFFFFFFFu;

  int old = 0;
  if (tid == 0 && w == 0) idxs[0] = 0;
  bool dead = false;

  for (int j = 1; j < m; ++j) {
#if SIG3D_FPS_PROBE == 3
    const float x1 = 0.5f + 1e-3f * (float)(old & 1023), y1 = 0.25f, z1 = 0.125f;   // no dependent load (wrong results)
#else
    const float x1 = dataset[3 * old + 0], y1 = dataset[3 * old + 1], z1 = dataset[3 * old + 2];
#endif
    int wv;
    unsigned wk;
    bool sit_out = false;
    if (BLOCKED) {
      // squared distance from the sample to the block's box, shrunk by more than any rounding of either side: every
      // point's individually rounded distance is then >= its running distance (NaN compares false: sweep)
      const float ex = fmaxf(fmaxf(lo_x - x1, x1 - hi_x), 0.f), ey = fmaxf(fmaxf(lo_y - y1, y1 - hi_y), 0.f);
      const float ez = fmaxf(fmaxf(lo_z - z1, z1 - hi_z), 0.f);
      const float d_box = (ex * ex + ey * ey + ez * ez) * 0.99998f;
      sit_out = held_v < 0 || d_box > __builtin_bit_cast(float, held_v);
      sit_out = __builtin_amdgcn_readfirstlane((int)sit_out) != 0;
    }
    if (!sit_out) {
      float best = -1.f;
      int bslot = 0;
#pragma unroll
      for (int s = 0; s < (SIG3D_FPS_PROBE == 1 ? PPT / 2 : PPT); ++s) {
        const float d = sq_dist3(px[s], py[s], pz[s], x1, y1, z1);
        const float t = fminf(d, pt[s]);
        pt[s] = t;
        const bool gt = t > best;
        best = gt ? t : best;
        bslot = gt ? s : bslot;
      }
      const int myv = __builtin_bit_cast(int, best);
      wv = wave_allreduce_max_i32(myv);
      const unsigned bkey = BLOCKED ? s_pkey[bslot][tid] : fps_key((unsigned)(r_slot + 512 * (bslot * G + grp)), L, bsmask);
      wk = wave_allreduce_min_u32((myv == wv) ? bkey : 0xFFFFFFFFu);
      held_v = wv; held_k = wk;
    } else {
      wv = held_v; wk = held_k;
    }
    const int par = j & 1;
    if (lane == 0) {
      s_val[par][wave] = wv;
      s_key[par][wave] = wk;
    }
    __syncthreads();
    if (wave == 0) {
      const int ov = s_val[par][lane & (NW - 1)];
      const unsigned ok = s_key[par][lane & (NW - 1)];
      const int lv = row_allreduce_max_i32(ov);
      const unsigned lk = row_allreduce_min_u32(ov == lv ? ok : 0xFFFFFFFFu);
      gu64 *slot = slots + (size_t)par * (2 * W);
      // PACK (n <= 65536, L >= 6): ONE granule {round:16 | key:16 | value:32} per workgroup and round -- the hop pays
      // per granule (each is its own write-through and its own polled load).  The 16-bit key is the tie key with its
      // two fields moved together: (bitrev_L(k mod 2^L) << (16 - L)) | (k >> L), same order.
      if (lane == 0) {
        if (PACK) {
          const unsigned ck = ((lk >> 22) << (16 - L)) | (lk & 0x3FFFFFu);
          __hip_atomic_store(slot + w, ((u64)(unsigned)(j & 0xFFFF) << 48) | ((u64)(ck & 0xFFFFu) << 32) | (unsigned)lv,
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          __hip_atomic_store(slot + 2 * w + 0, ((u64)(unsigned)j << 32) | (unsigned)lv,
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(slot + 2 * w + 1, ((u64)(unsigned)j << 32) | lk, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      // lanes 0..W-1 each wait for one peer's pair (lane w reads its own, already published)
      int pv = (int)0x80000000;
      unsigned pk = 0xFFFFFFFFu;
      unsigned spins = 0;
      bool ok_all = false;
      while (!ok_all) {
        bool mine = true;
        if (lane < W) {
          if (PACK) {
            const u64 a = __hip_atomic_load(slot + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            mine = (unsigned)(a >> 48) == (unsigned)(j & 0xFFFF);
            pv = (int)(unsigned)a;
            const unsigned ck = (unsigned)(a >> 32) & 0xFFFFu;
            pk = ((ck >> (16 - L)) << 22) | (ck & ((1u << (16 - L)) - 1u));
          } else {
          const u64 a = __hip_atomic_load(slot + 2 * lane + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const u64 c = __hip_atomic_load(slot + 2 * lane + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          mine = ((unsigned)(a >> 32) == (unsigned)j) && ((unsigned)(c >> 32) == (unsigned)j);
          pv = (int)(unsigned)a;
          pk = (unsigned)c;
          }
        }
        ok_all = __all(mine);
        if (!ok_all) {
          if (++spins > (1u << 22)) { dead = true; break; }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      if (lane >= W) { pv = (int)0x80000000; pk = 0xFFFFFFFFu; }
      const int gv = W <= 16 ? row_allreduce_max_i32(pv) : wave_allreduce_max_i32(pv);
      const unsigned gk = W <= 16 ? row_allreduce_min_u32(pv == gv ? pk : 0xFFFFFFFFu)
                                  : wave_allreduce_min_u32(pv == gv ? pk : 0xFFFFFFFFu);
      if (lane == 0) {
        s_win[par][0] = dead ? (int)0x80000001 : gv;
        s_win[par][1] = (int)gk;
      }
    }
    __syncthreads();
    const int gvu = __builtin_amdgcn_readfirstlane(s_win[par][0]);
    const unsigned gku = (unsigned)__builtin_amdgcn_readfirstlane(s_win[par][1]);
    if (gvu == (int)0x80000001) {  // a peer never showed up: poison instead of hanging
      if (w == 0)
        for (int jj = j + tid; jj < m; jj += NT) idxs[jj] = -1;
      // ... and REPORT it: the entry point returned 0 long ago (asynchronous launch, possibly a graph
      // replay), so the host reads this counter through sig3d_fps_timeout_count (once per step / epoch)
      if (w == 0 && tid == 0) atomicAdd(&g_fps_timeouts, 1u);
      return;
    }
    old = (gvu < 0) ? 0 : (int)fps_unkey(gku, L);
    if (tid == 0 && w == 0) idxs[j] = old;
  }
}

// The scene's point numbers in the order of a Morton curve over a 32 x 32 x 8 grid of its bounding box (8 x 8 x 8 cells
// interleaved bit by bit, the 4 x 4 such cubes in x and y above them): a counting
// sort in LDS, one workgroup per scene.  Which of a cell's points comes first is left to the atomics -- the sampling
// result does not depend on the order, only how compact a wave's block is.  n <= 65535 (16-bit point numbers).
__device__ __forceinline__ unsigned fps_spread3(unsigned v) {
  v = (v | (v << 8)) & 0x0300F00Fu;
  v = (v | (v << 4)) & 0x030C30C3u;
  v = (v | (v << 2)) & 0x09249249u;
  return v;
}
constexpr int FPS_CELLS = 32 * 32 * 8;

__global__ __launch_bounds__(1024) void fps_morton_order_kernel(int n, const float *__restrict__ dataset_all,
                                                                unsigned short *__restrict__ order_all) {
  __shared__ int s_bin[FPS_CELLS];
  __shared__ float s_box[6][16];
  __shared__ int s_scan[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float *dataset = dataset_all + (size_t)blockIdx.x * n * 3;
  unsigned short *order = order_all + (size_t)blockIdx.x * n;
  float ax = 3.0e38f, ay = 3.0e38f, az = 3.0e38f, bx = -3.0e38f, by = -3.0e38f, bz = -3.0e38f;
  for (int k = tid; k < n; k += 1024) {
    const float x = dataset[3 * k + 0], y = dataset[3 * k + 1], z = dataset[3 * k + 2];
    ax = fminf(ax, x); ay = fminf(ay, y); az = fminf(az, z);
    bx = fmaxf(bx, x); by = fmaxf(by, y); bz = fmaxf(bz, z);
  }
  for (int o = 32; o >= 1; o >>= 1) {
    ax = fminf(ax, __shfl_xor(ax, o)); ay = fminf(ay, __shfl_xor(ay, o)); az = fminf(az, __shfl_xor(az, o));
    bx = fmaxf(bx, __shfl_xor(bx, o)); by = fmaxf(by, __shfl_xor(by, o)); bz = fmaxf(bz, __shfl_xor(bz, o));
  }
  if (lane == 0) {
    s_box[0][wave] = ax; s_box[1][wave] = ay; s_box[2][wave] = az;
    s_box[3][wave] = bx; s_box[4][wave] = by; s_box[5][wave] = bz;
  }
  for (int i = tid; i < FPS_CELLS; i += 1024) s_bin[i] = 0;
  __syncthreads();
  for (int i = 0; i < 16; ++i) {
    ax = fminf(ax, s_box[0][i]); ay = fminf(ay, s_box[1][i]); az = fminf(az, s_box[2][i]);
    bx = fmaxf(bx, s_box[3][i]); by = fmaxf(by, s_box[4][i]); bz = fmaxf(bz, s_box[5][i]);
  }
  const float sx = 32.f / (bx - ax), sy = 32.f / (by - ay), sz = 8.f / (bz - az);
  // NaN, infinities and a flat box all land in a valid cell: fmaxf(NaN, 0) = 0
  auto cell_of = [&](int k) {
    const float x = dataset[3 * k + 0], y = dataset[3 * k + 1], z = dataset[3 * k + 2];
    const unsigned ix = (unsigned)fminf(fmaxf((x - ax) * sx, 0.f), 31.f);
    const unsigned iy = (unsigned)fminf(fmaxf((y - ay) * sy, 0.f), 31.f);
    const unsigned iz = (unsigned)fminf(fmaxf((z - az) * sz, 0.f), 7.f);
    const unsigned hx = ix >> 3, hy = iy >> 3;
    const unsigned hi = (hx & 1u) | ((hy & 1u) << 1) | ((hx & 2u) << 1) | ((hy & 2u) << 2);
    return (int)(fps_spread3(ix & 7u) | (fps_spread3(iy & 7u) << 1) | (fps_spread3(iz) << 2) | (hi << 9));
  };
  for (int k = tid; k < n; k += 1024) atomicAdd(&s_bin[cell_of(k)], 1);
  __syncthreads();
  // exclusive scan: 8 consecutive cells per thread
  constexpr int CPT = FPS_CELLS / 1024;
  int c[CPT], sum = 0;
#pragma unroll
  for (int i = 0; i < CPT; ++i) { c[i] = s_bin[tid * CPT + i]; sum += c[i]; }
  int inc = sum;
  for (int o = 1; o < 64; o <<= 1) {
    const int up = __shfl_up(inc, o);
    if (lane >= o) inc += up;
  }
  if (lane == 63) s_scan[wave] = inc;
  __syncthreads();
  int base = inc - sum;
  for (int i = 0; i < wave; ++i) base += s_scan[i];
#pragma unroll
  for (int i = 0; i < CPT; ++i) { s_bin[tid * CPT + i] = base; base += c[i]; }
  __syncthreads();
  for (int k = tid; k < n; k += 1024) {
    const int pos = atomicAdd(&s_bin[cell_of(k)], 1);
    if (pos >= 0 && pos < n) order[pos] = (unsigned short)k;
  }
}

template <int NT, int PPT, int W, bool BLOCKED = false>
int launch_fps_coop(int b, int n, int m, int L, const float *dataset, float *temp, int *idxs,
                    hipStream_t stream) {
  // one packed granule per workgroup and round when the key fits 16 bits, else the two-granule form
  const bool pack = n <= 65536 && m <= 65536 && L >= 6 && L <= 16;
  // the granule slots live at the front of the caller's temp scratch (b*n floats >= b*128), the Morton order of the
  // blocked form behind them (b * n 16-bit numbers: b * 1024 + 2 b n <= 4 b n bytes from n = 512)
  SIG3D_HIP_TRY(hipMemsetAsync(temp, 0, sizeof(u64) * (size_t)b * FPS_SLOT_U64, stream));
  unsigned short *order = nullptr;
  if (BLOCKED) {
    SIG3D_REQUIRE(n >= 512 && n <= 65535, "the blocked cooperative FPS numbers a scene's points in 16 bits (512 <= n <= 65535)");
    SIG3D_REQUIRE((n + W * (NT / 64) - 1) / (W * (NT / 64)) <= PPT * 64,
                  "the blocked cooperative FPS: a wave's block must fit its PPT x 64 register slots");
    order = reinterpret_cast<unsigned short *>(reinterpret_cast<char *>(temp) + sizeof(u64) * (size_t)b * FPS_SLOT_U64);
    hipLaunchKernelGGL(fps_morton_order_kernel, dim3(b), dim3(1024), 0, stream, n, dataset, order);
    SIG3D_LAUNCH_CHECK("fps_morton_order_kernel");
  }
  if (pack)
    hipLaunchKernelGGL((fps_coop_kernel<NT, PPT, W, true, BLOCKED>), dim3(b * W), dim3(NT), 0, stream, b, n, m, L,
                       dataset, (u64 *)temp, idxs, order);
  else
    hipLaunchKernelGGL((fps_coop_kernel<NT, PPT, W, false, BLOCKED>), dim3(b * W), dim3(NT), 0, stream, b, n, m, L,
                       dataset, (u64 *)temp, idxs, order);
  SIG3D_LAUNCH_CHECK("fps_coop_kernel");
  return 0;
}

// ---- block-list FPS: ONE workgroup per scene, the scene in L2, only the blocks a sample can reach are swept --------
// The cooperative kernel above keeps a scene in the registers of 8 workgroups: 64 CUs hold ~120 VGPRs a lane for
// 3.4 ms of every step and every round pays a global-memory hop between them.  Its blocked form already showed that a
// sample changes a small part of the scene after the first few dozen rounds.  Here that is the whole design:
//   * a pre-pass (fps_blocks_sort_kernel) stores the scene along a Morton curve as {x, y, z, tie key} rows; a block is
//     64 * PPL consecutive rows (625 blocks of 64 at n = 40 000); the running distances live beside them in global
//     memory -- 800 KB per scene, in the XCD's L2 / the memory-side cache (workgroup i lands on XCD i mod 8);
//   * lane l of wave w MANAGES blocks (64 s + l) * NW + w, s < MB: box, largest running distance, the tie key and the
//     coordinates of the point that holds it -- all in that lane's registers (11 VGPRs per block), nothing of it in LDS;
//   * a round: every lane tests the new sample against its blocks' boxes (the blocked kernel's test, same margin); a
//     wave sweeps the blocks of its own lanes that can change (ballot -> scalar loop; one point per lane and row, the
//     loads of up to U blocks in flight), reduces each (max distance, then min key: the reference's total order)
//     and hands the result to the managing lane; the wave's candidate {value, key, x, y, z} goes to LDS, ONE barrier,
//     every wave reduces the NW candidates itself.  The next sample's coordinates come with the candidate: no
//     dependent global load, no exchange between CUs.
// Waves only touch the running distances of their own blocks, so no memory ordering between waves is needed, and the
// barrier waits for the LDS write only (the distance stores drain behind it).  Same indices as every other kernel of
// this file: the same individually rounded distance, min, skip rule and (value, key) order; a block that sits a round
// out holds exactly what a sweep would have left.
// What it is for: a chain beside a training step costs the step the wave slots it holds -- 16
// waves per scene +0.27 ms, 4 waves -0.07 ms against the cooperative kernel -- but every round here waits for L2 /
// memory, 1.5-2 x longer beside the step's traffic, and the chain is then 10-14 ms long where the cooperative kernel's
// is 7.7.  Opt-in (SIG3D_FPS_BLOCKS=1); sig3d_furthest_point_sampling keeps the cooperative kernel.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __builtin_bit_cast(float, dpp_i32<CTRL>(__builtin_bit_cast(int, v)));
}
__device__ __forceinline__ float readlane_f32(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
__device__ __forceinline__ float wave_allreduce_fmin(float v) {
  v = fminf(v, dpp_f32<0xB1>(v)); v = fminf(v, dpp_f32<0x4E>(v)); v = fminf(v, dpp_f32<0x141>(v)); v = fminf(v, dpp_f32<0x140>(v));
  return fminf(fminf(readlane_f32(v, 0), readlane_f32(v, 16)), fminf(readlane_f32(v, 32), readlane_f32(v, 48)));
}
__device__ __forceinline__ float wave_allreduce_fmax(float v) {
  v = fmaxf(v, dpp_f32<0xB1>(v)); v = fmaxf(v, dpp_f32<0x4E>(v)); v = fmaxf(v, dpp_f32<0x141>(v)); v = fmaxf(v, dpp_f32<0x140>(v));
  return fmaxf(fmaxf(readlane_f32(v, 0), readlane_f32(v, 16)), fmaxf(readlane_f32(v, 32), readlane_f32(v, 48)));
}

constexpr unsigned FPSB_NOKEY = 0xFFFFFFFFu;

// Morton order of the scene (the grid of fps_morton_order_kernel) written out as rows {x, y, z, key}; rows [n, n_pad)
// are padding (key = FPSB_NOKEY: never take part).  Which of a cell's points comes first is left to the atomics.
__global__ __launch_bounds__(1024) void fps_blocks_sort_kernel(int n, int n_pad, int L,
                                                               const float *__restrict__ dataset_all,
                                                               float4 *__restrict__ rows_all) {
  __shared__ int s_bin[FPS_CELLS];
  __shared__ float s_box[6][16];
  __shared__ int s_scan[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float *dataset = dataset_all + (size_t)blockIdx.x * n * 3;
  float4 *rows = rows_all + (size_t)blockIdx.x * n_pad;
  const unsigned bsmask = (1u << L) - 1u;
  float ax = 3.0e38f, ay = 3.0e38f, az = 3.0e38f, bx = -3.0e38f, by = -3.0e38f, bz = -3.0e38f;
  for (int k = tid; k < n; k += 1024) {
    const float x = dataset[3 * k + 0], y = dataset[3 * k + 1], z = dataset[3 * k + 2];
    ax = fminf(ax, x); ay = fminf(ay, y); az = fminf(az, z);
    bx = fmaxf(bx, x); by = fmaxf(by, y); bz = fmaxf(bz, z);
  }
  ax = wave_allreduce_fmin(ax); ay = wave_allreduce_fmin(ay); az = wave_allreduce_fmin(az);
  bx = wave_allreduce_fmax(bx); by = wave_allreduce_fmax(by); bz = wave_allreduce_fmax(bz);
  if (lane == 0) {
    s_box[0][wave] = ax; s_box[1][wave] = ay; s_box[2][wave] = az;
    s_box[3][wave] = bx; s_box[4][wave] = by; s_box[5][wave] = bz;
  }
  for (int i = tid; i < FPS_CELLS; i += 1024) s_bin[i] = 0;
  __syncthreads();
  for (int i = 0; i < 16; ++i) {
    ax = fminf(ax, s_box[0][i]); ay = fminf(ay, s_box[1][i]); az = fminf(az, s_box[2][i]);
    bx = fmaxf(bx, s_box[3][i]); by = fmaxf(by, s_box[4][i]); bz = fmaxf(bz, s_box[5][i]);
  }
  const float sx = 32.f / (bx - ax), sy = 32.f / (by - ay), sz = 8.f / (bz - az);
  auto cell_of = [&](float x, float y, float z) {   // NaN, infinities and a flat box land in a valid cell
    const unsigned ix = (unsigned)fminf(fmaxf((x - ax) * sx, 0.f), 31.f);
    const unsigned iy = (unsigned)fminf(fmaxf((y - ay) * sy, 0.f), 31.f);
    const unsigned iz = (unsigned)fminf(fmaxf((z - az) * sz, 0.f), 7.f);
    const unsigned hx = ix >> 3, hy = iy >> 3;
    const unsigned hi = (hx & 1u) | ((hy & 1u) << 1) | ((hx & 2u) << 1) | ((hy & 2u) << 2);
    return (int)(fps_spread3(ix & 7u) | (fps_spread3(iy & 7u) << 1) | (fps_spread3(iz) << 2) | (hi << 9));
  };
  for (int k = tid; k < n; k += 1024)
    atomicAdd(&s_bin[cell_of(dataset[3 * k + 0], dataset[3 * k + 1], dataset[3 * k + 2])], 1);
  __syncthreads();
  constexpr int CPT = FPS_CELLS / 1024;
  int c[CPT], sum = 0;
#pragma unroll
  for (int i = 0; i < CPT; ++i) { c[i] = s_bin[tid * CPT + i]; sum += c[i]; }
  int inc = sum;
  for (int o = 1; o < 64; o <<= 1) {
    const int up = __shfl_up(inc, o);
    if (lane >= o) inc += up;
  }
  if (lane == 63) s_scan[wave] = inc;
  __syncthreads();
  int base = inc - sum;
  for (int i = 0; i < wave; ++i) base += s_scan[i];
#pragma unroll
  for (int i = 0; i < CPT; ++i) { s_bin[tid * CPT + i] = base; base += c[i]; }
  __syncthreads();
  for (int k = tid; k < n; k += 1024) {
    const float x = dataset[3 * k + 0], y = dataset[3 * k + 1], z = dataset[3 * k + 2];
    const int pos = atomicAdd(&s_bin[cell_of(x, y, z)], 1);
    if (pos >= 0 && pos < n)
      rows[pos] = make_float4(x, y, z, __builtin_bit_cast(float, fps_key((unsigned)k, L, bsmask)));
  }
  for (int k = n + tid; k < n_pad; k += 1024) rows[k] = make_float4(0.f, 0.f, 0.f, __builtin_bit_cast(float, FPSB_NOKEY));
}

template <int NW, int MB, int PPL, int U>
__global__ __launch_bounds__(NW * 64) void fps_blocks_kernel(int n, int n_pad, int m, int L,
                                                             const float *__restrict__ dataset_all,
                                                             const float4 *__restrict__ rows_all,
                                                             float *__restrict__ dist_all,
                                                             int *__restrict__ idxs_all) {
  constexpr int BP = 64 * PPL;
  static_assert(NW == 1 || NW == 2 || NW == 4 || NW == 8 || NW == 16, "the candidates of all waves are reduced inside a 16-lane row");
  // (Raised wave priority -- s_setprio 3 -- was tried for the case beside a training step, where the kernel takes 1.5 x
  // as long as alone: no effect, 14.1 against 14.2 ms per chain.  The extra time is memory latency under the step's
  // traffic, not issue arbitration.)
  __shared__ int s_part[2][NW][8];   // a wave's candidate: value bits, key, x, y, z

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float *dataset = dataset_all + (size_t)blockIdx.x * n * 3;
  const float4 *rows = rows_all + (size_t)blockIdx.x * n_pad;
  float *dist = dist_all + (size_t)blockIdx.x * n_pad;
  int *idxs = idxs_all + (size_t)blockIdx.x * m;
  const int nb = n_pad / BP;
  // wave w owns blocks w, w + NW, ...: its i-th block is MANAGED by slot i / 64 of lane i % 64 (i < mine <= 64 MB)
  const int mine = (nb - wave + NW - 1) / NW;

  // the reduction of one block's rows: largest running distance, lowest key among its holders, that point's coordinates
  struct Cand { int v; unsigned k; float x, y, z; };
  auto reduce_block = [&](const float4 (&p)[PPL], const float (&t)[PPL]) {
    float best = t[0];
    unsigned bk = __builtin_bit_cast(unsigned, p[0].w);
    float x = p[0].x, y = p[0].y, z = p[0].z;
#pragma unroll
    for (int q = 1; q < PPL; ++q) {
      const unsigned kq = __builtin_bit_cast(unsigned, p[q].w);
      const bool gt = t[q] > best || (t[q] == best && kq < bk);
      best = gt ? t[q] : best; bk = gt ? kq : bk;
      x = gt ? p[q].x : x; y = gt ? p[q].y : y; z = gt ? p[q].z : z;
    }
    const int myv = __builtin_bit_cast(int, best);   // >= 0 or -1: ordered as signed integers
    Cand c;
    c.v = wave_allreduce_max_i32(myv);
    c.k = wave_allreduce_min_u32(myv == c.v ? bk : FPSB_NOKEY);
    const unsigned long long holders = __builtin_amdgcn_ballot_w64(myv == c.v && bk == c.k);
    const int wl = (int)__builtin_ctzll(holders | (1ull << 63));
    c.x = readlane_f32(x, wl); c.y = readlane_f32(y, wl); c.z = readlane_f32(z, wl);
    return c;
  };

  // ---- prologue: running distances (1e10, or -1 for the points that never take part), boxes, first candidates
  int bval[MB];
  unsigned bkey[MB];
  float cx[MB], cy[MB], cz[MB], lo_x[MB], lo_y[MB], lo_z[MB], hi_x[MB], hi_y[MB], hi_z[MB];
#pragma unroll
  for (int s = 0; s < MB; ++s) {
    bval[s] = (int)0x80000000; bkey[s] = FPSB_NOKEY;
    cx[s] = cy[s] = cz[s] = lo_x[s] = lo_y[s] = lo_z[s] = hi_x[s] = hi_y[s] = hi_z[s] = 0.f;
  }
#pragma unroll
  for (int s = 0; s < MB; ++s) {
    for (int i = s * 64; i < min(mine, (s + 1) * 64); ++i) {
      const size_t base = (size_t)(i * NW + wave) * BP;
      float4 p[PPL];
      float t[PPL];
      float ax = 3.0e38f, ay = 3.0e38f, az = 3.0e38f, bx = -3.0e38f, by = -3.0e38f, bz = -3.0e38f;
#pragma unroll
      for (int q = 0; q < PPL; ++q) {
        p[q] = rows[base + q * 64 + lane];
        const float mag = __fadd_rn(__fadd_rn(__fmul_rn(p[q].x, p[q].x), __fmul_rn(p[q].y, p[q].y)), __fmul_rn(p[q].z, p[q].z));
        const bool pad = __builtin_bit_cast(unsigned, p[q].w) == FPSB_NOKEY;
        t[q] = (pad || (double)mag <= 1e-3) ? -1.f : 1e10f;  // sampling_gpu.cu:100-101, sampling.cpp:74-76
        dist[base + q * 64 + lane] = t[q];
        if (t[q] >= 0.f) {
          ax = fminf(ax, p[q].x); ay = fminf(ay, p[q].y); az = fminf(az, p[q].z);
          bx = fmaxf(bx, p[q].x); by = fmaxf(by, p[q].y); bz = fmaxf(bz, p[q].z);
        }
      }
      ax = wave_allreduce_fmin(ax); ay = wave_allreduce_fmin(ay); az = wave_allreduce_fmin(az);
      bx = wave_allreduce_fmax(bx); by = wave_allreduce_fmax(by); bz = wave_allreduce_fmax(bz);
      const Cand c = reduce_block(p, t);
      if (lane == i - s * 64) {
        bval[s] = c.v; bkey[s] = c.k; cx[s] = c.x; cy[s] = c.y; cz[s] = c.z;
        lo_x[s] = ax; lo_y[s] = ay; lo_z[s] = az; hi_x[s] = bx; hi_y[s] = by; hi_z[s] = bz;
      }
    }
  }

  // the wave's candidate over its managed blocks, with its coordinates, in every lane (wave-uniform)
  int pub_v = (int)0x80000000;
  unsigned pub_k = FPSB_NOKEY;
  float pub_x = 0.f, pub_y = 0.f, pub_z = 0.f;
  auto refresh_candidate = [&]() {
    int v = bval[0];
    unsigned k = bkey[0];
    float x = cx[0], y = cy[0], z = cz[0];
#pragma unroll
    for (int s = 1; s < MB; ++s) {
      const bool gt = bval[s] > v || (bval[s] == v && bkey[s] < k);
      v = gt ? bval[s] : v; k = gt ? bkey[s] : k;
      x = gt ? cx[s] : x; y = gt ? cy[s] : y; z = gt ? cz[s] : z;
    }
    pub_v = wave_allreduce_max_i32(v);
    pub_k = wave_allreduce_min_u32(v == pub_v ? k : FPSB_NOKEY);
    const int wl = (int)__builtin_ctzll(__builtin_amdgcn_ballot_w64(v == pub_v && k == pub_k) | (1ull << 63));
    pub_x = readlane_f32(x, wl); pub_y = readlane_f32(y, wl); pub_z = readlane_f32(z, wl);
  };
  refresh_candidate();

  float x1 = dataset[0], y1 = dataset[1], z1 = dataset[2];
  if (tid == 0) idxs[0] = 0;

  for (int j = 1; j < m; ++j) {
    bool swept = false;
#pragma unroll
    for (int s = 0; s < MB; ++s) {
      // which of my blocks can this sample change?  (the blocked cooperative kernel's test: squared distance to the
      // box, shrunk by more than any rounding of either side; NaN compares false and sweeps)
      const float ex = fmaxf(fmaxf(lo_x[s] - x1, x1 - hi_x[s]), 0.f), ey = fmaxf(fmaxf(lo_y[s] - y1, y1 - hi_y[s]), 0.f);
      const float ez = fmaxf(fmaxf(lo_z[s] - z1, z1 - hi_z[s]), 0.f);
      const float d_box = (ex * ex + ey * ey + ez * ez) * 0.99998f;
      const bool sit_out = bval[s] < 0 || d_box > __builtin_bit_cast(float, bval[s]);
      unsigned long long todo = __builtin_amdgcn_ballot_w64(!sit_out && s * 64 + lane < mine);
      swept |= todo != 0;

      // up to UU blocks per trip, all their loads in flight together (a trip is one round trip to L2; a short list is
      // padded with its last block: sweeping twice is idempotent)
      auto sweep = [&](auto ucount) {
        constexpr int UU = decltype(ucount)::value;
        int bi[UU], cnt = 0;
#pragma unroll
        for (int u = 0; u < UU; ++u) {
          const bool have = todo != 0;
          bi[u] = have ? (int)__builtin_ctzll(todo) : bi[u > 0 ? u - 1 : 0];
          if (have) { todo &= todo - 1; ++cnt; }
        }
        float4 p[UU][PPL];
        float t[UU][PPL];
#pragma unroll
        for (int u = 0; u < UU; ++u) {
          const size_t base = (size_t)((s * 64 + bi[u]) * NW + wave) * BP;
#pragma unroll
          for (int q = 0; q < PPL; ++q) {
            p[u][q] = rows[base + q * 64 + lane];
            t[u][q] = dist[base + q * 64 + lane];
          }
        }
#pragma unroll
        for (int u = 0; u < UU; ++u) {
          if (u < cnt) {
            const size_t base = (size_t)((s * 64 + bi[u]) * NW + wave) * BP;
#pragma unroll
            for (int q = 0; q < PPL; ++q) {
              const float d = sq_dist3(p[u][q].x, p[u][q].y, p[u][q].z, x1, y1, z1);
              const float tn = fminf(d, t[u][q]);
              if (tn != t[u][q]) dist[base + q * 64 + lane] = tn;
              t[u][q] = tn;
            }
            const Cand c = reduce_block(p[u], t[u]);
            if (lane == bi[u]) { bval[s] = c.v; bkey[s] = c.k; cx[s] = c.x; cy[s] = c.y; cz[s] = c.z; }
          }
        }
      };
      while (__builtin_popcountll(todo) > 2) sweep(std::integral_constant<int, U>{});
      if (todo) sweep(std::integral_constant<int, 2>{});
    }
    if (swept) refresh_candidate();

    int gvu;
    unsigned gku;
    if (NW == 1) {
      gvu = pub_v; gku = pub_k;
      x1 = pub_x; y1 = pub_y; z1 = pub_z;
    } else {
      const int par = j & 1;
      if (lane == 0) {
        s_part[par][wave][0] = pub_v;
        s_part[par][wave][1] = (int)pub_k;
        s_part[par][wave][2] = __builtin_bit_cast(int, pub_x);
        s_part[par][wave][3] = __builtin_bit_cast(int, pub_y);
        s_part[par][wave][4] = __builtin_bit_cast(int, pub_z);
      }
      // the barrier orders the LDS candidates only: running distances are private to their wave, and their stores
      // may still be in flight behind it
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");

      const int *sp = s_part[par][lane & (NW - 1)];
      const int ov = sp[0];
      const unsigned ok = (unsigned)sp[1];
      const float ox = __builtin_bit_cast(float, sp[2]), oy = __builtin_bit_cast(float, sp[3]), oz = __builtin_bit_cast(float, sp[4]);
      const int gv = row_allreduce_max_i32(ov);
      const unsigned gk = row_allreduce_min_u32(ov == gv ? ok : FPSB_NOKEY);
      gvu = __builtin_amdgcn_readfirstlane(gv);
      gku = (unsigned)__builtin_amdgcn_readfirstlane((int)gk);
      const int gl = (int)__builtin_ctzll(__builtin_amdgcn_ballot_w64(ov == gvu && ok == gku) | (1ull << 63));
      x1 = readlane_f32(ox, gl); y1 = readlane_f32(oy, gl); z1 = readlane_f32(oz, gl);
    }
    if (gvu < 0) {
      // nothing takes part (every point within the skip rule): the reference returns index 0 for every round
      for (int jj = j + tid; jj < m; jj += NW * 64) idxs[jj] = 0;
      return;
    }
    if (tid == 0) idxs[j] = (int)fps_unkey(gku, L);
  }
}

// rows of a scene in the block-list workspace: whole blocks, at most 64 * FPSB_NW of them
static int fpsb_points_per_lane(int n) { return n <= 65536 ? 1 : n <= 131072 ? 2 : 3; }
static long fpsb_padded(int n) {
  const long bp = 64L * fpsb_points_per_lane(n);
  return (n + bp - 1) / bp * bp;
}

// ---- gather_points: out[b,c,j] = points[b,c,idx[b,j]] ---------------------------------------
__global__ __launch_bounds__(256) void gather_points_kernel(int c, int n, int m,
                                                            const float *__restrict__ points,
                                                            const int *__restrict__ idx,
                                                            float *__restrict__ out) {
  const int bi = blockIdx.z;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= m) return;
  // clamped: a poisoned FPS result (-1, see fps_coop_kernel) must not become an out-of-bounds read
  const int a = min(max(idx[(size_t)bi * m + j], 0), n - 1);
  for (int l = blockIdx.y; l < c; l += gridDim.y)
    out[((size_t)bi * c + l) * m + j] = points[((size_t)bi * c + l) * n + a];
}

__global__ __launch_bounds__(256) void gather_points_grad_kernel(int c, int n, int m,
                                                                 const float *__restrict__ grad_out,
                                                                 const int *__restrict__ idx,
                                                                 float *__restrict__ grad_points) {
  const int bi = blockIdx.z;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= m) return;
  const int a = idx[(size_t)bi * m + j];
  for (int l = blockIdx.y; l < c; l += gridDim.y)
    unsafeAtomicAdd(grad_points + ((size_t)bi * c + l) * n + a,
                    grad_out[((size_t)bi * c + l) * m + j]);
}

// new_xyz[b,j,:] = xyz[b,idx[b,j],:] -- the (B,N,3)-layout gather that _PointnetSAModuleBase
// spells as transpose -> gather_operation -> transpose (pointnet2_modules.py:233-240)
__global__ __launch_bounds__(256) void gather_xyz_kernel(int n, int m, const float *__restrict__ xyz,
                                                         const int *__restrict__ idx,
                                                         float *__restrict__ out) {
  const int bi = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= m) return;
  const int a = min(max(idx[(size_t)bi * m + j], 0), n - 1);  // poisoned FPS indices stay in bounds
  const float *src = xyz + ((size_t)bi * n + a) * 3;
  float *dst = out + ((size_t)bi * m + j) * 3;
  dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];
}

// cuda_utils.h:13-19 of the reference (host libm, double log ratio truncated)
int ref_opt_n_threads_log2(int work_size) {
  int pow_2 = (int)(log((double)work_size) / log(2.0));
  if (pow_2 > 9) pow_2 = 9;
  if (pow_2 < 0) pow_2 = 0;
  return pow_2;
}

}  // namespace

extern "C" int sig3d_furthest_point_sampling(int b, int n, int m, const float *dataset,
                                             float *temp, int *idxs, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SIG3D_REQUIRE(b >= 0 && n >= 0 && m >= 0, "negative size");
  if (b == 0 || m <= 0) return 0;  // sampling_gpu.cu:73
  SIG3D_REQUIRE(n >= 1, "furthest_point_sampling needs n >= 1");
  SIG3D_REQUIRE((long)n < (1L << 31) - 1024, "n too large");
  const int L = ref_opt_n_threads_log2(n);
  // NT must be a multiple of the reference block size 2^L (<= 512).
  if (n <= 256) return launch_fps<256, 1>(b, n, m, L, dataset, temp, idxs, stream);
  if (n < 512) return launch_fps<256, 2>(b, n, m, L, dataset, temp, idxs, stream);
  if (n <= 512) return launch_fps<512, 1>(b, n, m, L, dataset, temp, idxs, stream);
  if (n <= 1024) return launch_fps<512, 2>(b, n, m, L, dataset, temp, idxs, stream);
  if (n <= 2048) return launch_fps<512, 4>(b, n, m, L, dataset, temp, idxs, stream);
  if (n <= 4096) return launch_fps<512, 8>(b, n, m, L, dataset, temp, idxs, stream);
  if (n <= 8192) return launch_fps<1024, 8>(b, n, m, L, dataset, temp, idxs, stream);
  if (n <= 196608) {
    // cooperative kernel; at most 64 co-resident workgroups per launch (so that several
    // concurrent launches -- other streams, other processes -- can never starve each other)
    // 8 workgroups of 512 threads per scene: measured 1.76 us/round at n = 40 000 against 2.18 for
    // 4 x 1024 (half the points per SIMD between two exchanges, 8-wave barriers) and 2.07 for
    // 16 x 512 (more peers to wait for per hop); 8 scenes = 64 workgroups per launch
    // (4 x 1024: 2.18 us/round, 2 x 1024: 2.97 -- tools/fps_bench.py)
    const int chunk = 8;
    for (int s0 = 0; s0 < b; s0 += chunk) {
      const int bc = (b - s0) < chunk ? (b - s0) : chunk;
      const float *ds = dataset + (size_t)s0 * n * 3;
      float *tp = temp + (size_t)s0 * n;
      int *ix = idxs + (size_t)s0 * m;
      int rc;
      if (n <= 16384) rc = launch_fps_coop<512, 4, 8>(bc, n, m, L, ds, tp, ix, stream);
      else if (n <= 24576) rc = launch_fps_coop<512, 6, 8>(bc, n, m, L, ds, tp, ix, stream);
      else if (n <= 40960) {
        // 8 x 256 threads x 20 points, waves that own a compact block of the scene's Morton order and sit out the rounds
        // that cannot change it (fps_coop_kernel BLOCKED).  What was measured on the way (DESIGN.md sections 4e / 4i):
        // 8 x 512 x 10 unblocked 1.76 us / round alone (this: 1.71), 16 x 256 x 10 2.29, 32 x 128 x 10 3.03 (more peers
        // per hop), 4 x 256 x 40 / 8 x 128 x 40 2.18 / 2.15; beside the training step every thinner or fatter shape cost
        // +0.12 ... +2.3 ms, the blocked form -0.062 +- 0.008.  Those instances left the library in round 6.
        rc = launch_fps_coop<256, 20, 8, true>(bc, n, m, L, ds, tp, ix, stream);
      }
      else if (n <= 65536) rc = launch_fps_coop<512, 16, 8>(bc, n, m, L, ds, tp, ix, stream);
      else if (n <= 98304) rc = launch_fps_coop<512, 24, 8>(bc, n, m, L, ds, tp, ix, stream);
      else rc = launch_fps_coop<1024, 24, 8>(bc, n, m, L, ds, tp, ix, stream);
      if (rc) return rc;
    }
    return 0;
  }
  return launch_fps<1024, 24>(b, n, m, L, dataset, temp, idxs, stream);  // global-memory tail
}

// FPS with a caller-provided workspace (sig3d_fps_blocks_workspace_bytes): scenes of 8193 .. 196 608 points run the
// block-list kernel -- one workgroup per scene instead of the cooperative kernel's eight -- every other size the kernels
// of sig3d_furthest_point_sampling with the workspace as their `temp`.  Same indices for any input.
extern "C" long sig3d_fps_blocks_workspace_bytes(int b, int n) {
  if (b < 0 || n < 0) return -1;
  return (long)b * fpsb_padded(n) * (long)(sizeof(float4) + sizeof(float));
}

extern "C" int sig3d_furthest_point_sampling_blocks(int b, int n, int m, const float *dataset, void *work,
                                                    long work_bytes, int waves, int *idxs, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SIG3D_REQUIRE(b >= 0 && n >= 0 && m >= 0, "negative size");
  if (b == 0 || m <= 0) return 0;  // sampling_gpu.cu:73
  SIG3D_REQUIRE(n >= 1, "furthest_point_sampling needs n >= 1");
  SIG3D_REQUIRE(work != nullptr && work_bytes >= sig3d_fps_blocks_workspace_bytes(b, n),
                "workspace smaller than sig3d_fps_blocks_workspace_bytes(b, n)");
  SIG3D_REQUIRE(((uintptr_t)work & 15u) == 0, "workspace must be 16-byte aligned");
  SIG3D_REQUIRE(waves == 0 || waves == 4 || waves == 8 || waves == 16, "waves must be 0 (default), 4, 8 or 16");
  if (n <= 8192 || n > 196608)
    return sig3d_furthest_point_sampling(b, n, m, dataset, (float *)work, idxs, stream_);
  const int L = ref_opt_n_threads_log2(n);
  const int ppl = fpsb_points_per_lane(n);
  const long n_pad = fpsb_padded(n);
  const long nblocks = n_pad / (64 * ppl);
  float4 *rows = (float4 *)work;
  float *dist = (float *)(rows + (size_t)b * n_pad);
  hipLaunchKernelGGL(fps_blocks_sort_kernel, dim3(b), dim3(1024), 0, stream, n, (int)n_pad, L, dataset, rows);
  SIG3D_LAUNCH_CHECK("fps_blocks_sort_kernel");
  // waves per scene x managed blocks per lane x points per lane and block x blocks per trip.  `waves` is the caller's
  // choice between latency and footprint: 16 waves finish 2047 rounds over 40 000 points in 3.6 ms
  // but hold four wave slots per SIMD of their CU (+0.27 ms on a training step that runs beside them); 4 waves take
  // 6.1 ms and cost the step nothing.  0 = 4.
#define SIG3D_FPSB(NWV, MBV, PPLV, UV)                                                                              \
  do {                                                                                                              \
    SIG3D_REQUIRE(nblocks <= 64L * NWV * MBV, "block-list FPS: more blocks than managing lanes");                    \
    hipLaunchKernelGGL((fps_blocks_kernel<NWV, MBV, PPLV, UV>), dim3(b), dim3(NWV * 64), 0, stream, n, (int)n_pad,   \
                       m, L, dataset, rows, dist, idxs);                                                            \
  } while (0)
  if (ppl == 1 && waves == 16) SIG3D_FPSB(16, 1, 1, 4);
  else if (ppl == 1 && waves == 8) SIG3D_FPSB(8, 2, 1, 4);
  else if (ppl == 1) SIG3D_FPSB(4, 4, 1, 4);
  else if (ppl == 2) SIG3D_FPSB(16, 1, 2, 4);
  else SIG3D_FPSB(16, 1, 3, 4);
#undef SIG3D_FPSB
  SIG3D_LAUNCH_CHECK("fps_blocks_kernel");
  return 0;
}

// the segmented check from 256 rounds (8 threads per point); below, the plain one
static void launch_prefix_check(const FpsChain &ch, int b, const float *dataset, const float *r, int *flags,
                                hipStream_t stream) {
  const bool seg = ch.m[0] >= 256 && ch.m[0] <= 4000;   // <= 64 KB of LDS
  if (seg) {
    constexpr int S = 8;
    hipLaunchKernelGGL(fps_prefix_check_seg_kernel<S>, dim3(sig3d_ceil_div(ch.n[0], 256 / S), b, ch.levels), dim3(256),
                       sizeof(float) * (4 * (size_t)ch.m[0] + 256), stream, ch, dataset, r, flags);
  } else {
    hipLaunchKernelGGL(fps_prefix_check_kernel, dim3(sig3d_ceil_div(ch.n[0], 256), b, ch.levels), dim3(256),
                       sizeof(float) * 4 * (size_t)ch.m[0], stream, ch, dataset, r, flags);
  }
}

static int launch_fps_flagged(int b, int n, int m, int L, const float *dataset, float *temp, int *idxs,
                              hipStream_t stream, const int *flags) {
  if (n <= 256) return launch_fps<256, 1>(b, n, m, L, dataset, temp, idxs, stream, flags);
  if (n < 512) return launch_fps<256, 2>(b, n, m, L, dataset, temp, idxs, stream, flags);
  if (n <= 512) return launch_fps<512, 1>(b, n, m, L, dataset, temp, idxs, stream, flags);
  if (n <= 1024) return launch_fps<512, 2>(b, n, m, L, dataset, temp, idxs, stream, flags);
  if (n <= 2048) return launch_fps<512, 4>(b, n, m, L, dataset, temp, idxs, stream, flags);
  if (n <= 4096) return launch_fps<512, 8>(b, n, m, L, dataset, temp, idxs, stream, flags);
  return launch_fps<1024, 8>(b, n, m, L, dataset, temp, idxs, stream, flags);
}

// FPS of a cloud that is (expected to be) the FPS-ordered output of an earlier FPS: identical
// results to sig3d_furthest_point_sampling for ANY input, the dependent rounds only run for the
// scenes where the prefix property fails.  `flags`: b ints of scratch (1 = proven, 0 = computed).
extern "C" int sig3d_furthest_point_sampling_nested(int b, int n, int m, const float *dataset,
                                                    float *temp, int *idxs, int *flags,
                                                    void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SIG3D_REQUIRE(b >= 0 && n >= 0 && m >= 0, "negative size");
  if (b == 0 || m <= 0) return 0;
  SIG3D_REQUIRE(n >= 1, "furthest_point_sampling needs n >= 1");
  SIG3D_REQUIRE(flags != nullptr, "flags must not be null");
  // the check holds the first m points in LDS and pays off against the single-workgroup kernels
  if (m > n || m > 4096 || n > 8192)
    return sig3d_furthest_point_sampling(b, n, m, dataset, temp, idxs, stream_);
  const int L = ref_opt_n_threads_log2(n);
  FpsChain ch{};
  ch.levels = 1; ch.stride = n; ch.n[0] = n; ch.m[0] = m; ch.L[0] = L;
  hipLaunchKernelGGL(fps_prefix_radius_kernel, dim3(sig3d_ceil_div(m, 4), b), dim3(256), 0, stream,
                     n, m, 1, dataset, temp, flags);
  SIG3D_LAUNCH_CHECK("fps_prefix_radius_kernel");
  launch_prefix_check(ch, b, dataset, temp, flags, stream);
  SIG3D_LAUNCH_CHECK("fps_prefix_check_kernel");
  return launch_fps_flagged(b, n, m, L, dataset, temp, idxs, stream, flags);
}

// A chain of nested samplings in one proof: level l draws m[l] points from the output of level l-1 (n[0] = n0 points
// of `dataset`, n[l] = m[l-1]), exactly as calling sig3d_furthest_point_sampling_nested + sig3d_gather_xyz level by
// level -- but the prefix property of ALL levels is checked by one radius launch and one check launch over the same
// FPS-ordered array (level l's input is its first n[l] points as long as every level above was proven; a level that
// fails also fails the levels below it, which then run the dependent rounds on their true input).  The geometry
// chain ends with these levels, and the training step waits for the chain: three proofs back to back were ~250 us.
extern "C" int sig3d_fps_nested_chain(int b, int n0, int nlevels, const int *m, const float *dataset, float *temp,
                                      int *const *idxs, float *const *new_xyz, int *flags, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SIG3D_REQUIRE(b >= 0 && n0 >= 1 && nlevels >= 1 && nlevels <= 4, "1 to 4 levels");
  SIG3D_REQUIRE(m != nullptr && idxs != nullptr && new_xyz != nullptr && flags != nullptr, "null pointer");
  if (b == 0) return 0;
  FpsChain ch{};
  ch.levels = nlevels; ch.stride = n0;
  int n = n0;
  for (int l = 0; l < nlevels; ++l) {
    SIG3D_REQUIRE(m[l] >= 1 && m[l] <= n && m[l] <= 4096 && n <= 8192, "levels must shrink: 1 <= m[l] <= n[l] <= 8192, m[l] <= 4096");
    ch.n[l] = n; ch.m[l] = m[l]; ch.L[l] = ref_opt_n_threads_log2(n);
    n = m[l];
  }
  hipLaunchKernelGGL(fps_prefix_radius_kernel, dim3(sig3d_ceil_div(m[0], 4), b), dim3(256), 0, stream,
                     n0, m[0], nlevels, dataset, temp, flags);
  SIG3D_LAUNCH_CHECK("fps_prefix_radius_kernel");
  launch_prefix_check(ch, b, dataset, temp, flags, stream);
  SIG3D_LAUNCH_CHECK("fps_prefix_check_kernel");
  const float *cur = dataset;
  for (int l = 0; l < nlevels; ++l) {
    // temp's r[] has been consumed by the check; the flagged kernel may use it as its scratch
    const int rc = launch_fps_flagged(b, ch.n[l], ch.m[l], ch.L[l], cur, temp, idxs[l], stream, flags + (size_t)l * b);
    if (rc) return rc;
    const int rg = sig3d_gather_xyz(b, ch.n[l], ch.m[l], cur, idxs[l], new_xyz[l], stream_);
    if (rg) return rg;
    cur = new_xyz[l];
  }
  return 0;
}

extern "C" int sig3d_fps_timeout_count(unsigned *count, int reset) {
  SIG3D_REQUIRE(count != nullptr, "count must not be null");
  SIG3D_HIP_TRY(hipMemcpyFromSymbol(count, HIP_SYMBOL(g_fps_timeouts), sizeof(unsigned)));
  if (reset) {
    const unsigned zero = 0;
    SIG3D_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_fps_timeouts), &zero, sizeof(unsigned)));
  }
  return 0;
}

extern "C" int sig3d_gather_points(int b, int c, int n, int npoints, const float *points,
                                   const int *idx, float *out, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SIG3D_REQUIRE(b >= 0 && c >= 0 && n >= 0 && npoints >= 0, "negative size");
  if (b == 0 || c == 0 || npoints == 0) return 0;
  dim3 grid(sig3d_ceil_div(npoints, 256), c < 64 ? c : 64, b);
  hipLaunchKernelGGL(gather_points_kernel, grid, dim3(256), 0, stream, c, n, npoints, points, idx,
                     out);
  SIG3D_LAUNCH_CHECK("gather_points_kernel");
  return 0;
}

extern "C" int sig3d_gather_points_grad(int b, int c, int n, int npoints, const float *grad_out,
                                        const int *idx, float *grad_points, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SIG3D_REQUIRE(b >= 0 && c >= 0 && n >= 0 && npoints >= 0, "negative size");
  if (b == 0 || c == 0 || n == 0) return 0;
  SIG3D_HIP_TRY(hipMemsetAsync(grad_points, 0, sizeof(float) * (size_t)b * c * n, stream));
  if (npoints == 0) return 0;
  dim3 grid(sig3d_ceil_div(npoints, 256), c < 64 ? c : 64, b);
  hipLaunchKernelGGL(gather_points_grad_kernel, grid, dim3(256), 0, stream, c, n, npoints,
                     grad_out, idx, grad_points);
  SIG3D_LAUNCH_CHECK("gather_points_grad_kernel");
  return 0;
}

extern "C" int sig3d_gather_xyz(int b, int n, int m, const float *xyz, const int *idx, float *out,
                                void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SIG3D_REQUIRE(b >= 0 && n >= 0 && m >= 0, "negative size");
  if (b == 0 || m == 0) return 0;
  dim3 grid(sig3d_ceil_div(m, 256), b);
  hipLaunchKernelGGL(gather_xyz_kernel, grid, dim3(256), 0, stream, n, m, xyz, idx, out);
  SIG3D_LAUNCH_CHECK("gather_xyz_kernel");
  return 0;
}
