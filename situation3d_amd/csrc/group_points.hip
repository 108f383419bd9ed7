// group_points.hip -- neighbourhood gather / scatter-add for gfx950.
//
// Replaces lib/pointnet2/_ext_src/src/group_points_gpu.cu of the reference and fuses the
// follow-up elementwise passes of QueryAndGroup.forward (pointnet2_utils.py:348-359).
//
// These kernels are HBM-write bound: the grouped tensor (b, c, npoints, nsample) is the
// dominant traffic of the whole point encoder.  The reference launches ONE block per batch
// element; here the (npoints*nsample) axis is spread over the grid, each lane owns FOUR
// consecutive samples (16-byte idx load, 16-byte stores), and walks a slab of channels with
// its four neighbour indices held in registers, so idx is read once per slab instead of once
// per channel.  The gathered rows (n floats per channel) are L2-resident.
#include <cstdlib>

#include "sig3d_common.h"

namespace {

constexpr int GP_THREADS = 256;
constexpr int GP_CSLAB = 8;  // channels walked per workgroup (idx reuse factor)

// out[b,l,e] = points[b,l,idx[b,e]], e over npoints*nsample (vector path: total % 4 == 0)
template <bool VEC4>
__global__ __launch_bounds__(GP_THREADS) void group_points_kernel(
    int c, int n, long total, const float *__restrict__ points, const int *__restrict__ idx,
    float *__restrict__ out) {
  const int bi = blockIdx.z;
  const long e = ((long)blockIdx.x * GP_THREADS + threadIdx.x) * (VEC4 ? 4 : 1);
  if (e >= total) return;
  const int *ip = idx + (size_t)bi * total + e;
  const int l0 = blockIdx.y * GP_CSLAB;
  const int l1 = min(l0 + GP_CSLAB, c);
  if (VEC4) {
    const int4 ii = *reinterpret_cast<const int4 *>(ip);
    float4 v[GP_CSLAB];
#pragma unroll
    for (int t = 0; t < GP_CSLAB; ++t) {  // gathers first (channel clamped), stores afterwards
      const float *row = points + ((size_t)bi * c + min(l0 + t, c - 1)) * n;
      v[t].x = row[ii.x]; v[t].y = row[ii.y]; v[t].z = row[ii.z]; v[t].w = row[ii.w];
    }
#pragma unroll
    for (int t = 0; t < GP_CSLAB; ++t)
      if (l0 + t < l1) *reinterpret_cast<float4 *>(out + ((size_t)bi * c + l0 + t) * total + e) = v[t];
  } else {
    const int ii = *ip;
    for (int l = l0; l < l1; ++l)
      out[((size_t)bi * c + l) * total + e] = points[((size_t)bi * c + l) * n + ii];
  }
}

// grad_points[b,l,idx[b,e]] += grad_out[b,c_off+l,e]   (hardware f32 atomics in L2)
template <bool VEC4>
__global__ __launch_bounds__(GP_THREADS) void group_points_grad_kernel(
    int c, int n, long total, int c_total, int c_off, const float *__restrict__ grad_out,
    const int *__restrict__ idx, float *__restrict__ grad_points) {
  const int bi = blockIdx.z;
  const long e = ((long)blockIdx.x * GP_THREADS + threadIdx.x) * (VEC4 ? 4 : 1);
  if (e >= total) return;
  const int *ip = idx + (size_t)bi * total + e;
  const int l0 = blockIdx.y * GP_CSLAB;
  const int l1 = min(l0 + GP_CSLAB, c);
  if (VEC4) {
    const int4 ii = *reinterpret_cast<const int4 *>(ip);
    for (int l = l0; l < l1; ++l) {
      const float4 g = *reinterpret_cast<const float4 *>(
          grad_out + ((size_t)bi * c_total + c_off + l) * total + e);
      float *row = grad_points + ((size_t)bi * c + l) * n;
      unsafeAtomicAdd(row + ii.x, g.x);
      unsafeAtomicAdd(row + ii.y, g.y);
      unsafeAtomicAdd(row + ii.z, g.z);
      unsafeAtomicAdd(row + ii.w, g.w);
    }
  } else {
    const int ii = *ip;
    for (int l = l0; l < l1; ++l)
      unsafeAtomicAdd(grad_points + ((size_t)bi * c + l) * n + ii,
                      grad_out[((size_t)bi * c_total + c_off + l) * total + e]);
  }
}

// LDS-privatised scatter-add.  The reference's one-atomicAdd-per-element scheme
// (group_points_gpu.cu:59-60) makes npoints*nsample*c read-modify-writes on n*c addresses in L2
// (16 hits per address at SA2) and measured 0.05 TB/s on MI355X.  Here a workgroup owns a slab
// of `cslab` channels of ONE scene in LDS (cslab*n floats <= 64 KiB), streams its share of
// grad_out with 16-byte loads, accumulates with ds_add_f32, and flushes the slab once (plain
// stores when it saw the whole (npoints*nsample) range, one global atomic per address otherwise).
constexpr int GG_LDS_FLOATS = 16384;  // 64 KiB

template <bool VEC4>
__global__ __launch_bounds__(GP_THREADS) void group_points_grad_lds_kernel(
    int c, int n, long total, int c_total, int c_off, int cslab, long e_per_block, int atomic_out,
    const float *__restrict__ grad_out, const int *__restrict__ idx,
    float *__restrict__ grad_points) {
  extern __shared__ __attribute__((aligned(16))) float s_acc[];
  const int bi = blockIdx.z;
  const int l0 = blockIdx.y * cslab;
  const int nl = min(cslab, c - l0);
  for (int i = threadIdx.x; i < nl * n; i += GP_THREADS) s_acc[i] = 0.f;
  __syncthreads();
  const long e_begin = (long)blockIdx.x * e_per_block;
  const long e_end = min(total, e_begin + e_per_block);
  constexpr int V = VEC4 ? 4 : 1;
  for (long e = e_begin + (long)threadIdx.x * V; e < e_end; e += GP_THREADS * V) {
    const int *ip = idx + (size_t)bi * total + e;
    int ii[4];
    if (VEC4) {
      const int4 t = *reinterpret_cast<const int4 *>(ip);
      ii[0] = t.x; ii[1] = t.y; ii[2] = t.z; ii[3] = t.w;
    } else {
      ii[0] = *ip;
    }
    // four channels per round: their loads are issued together (unconditionally, channel index
    // clamped) before the first ds_add -- one load -> wait -> atomics per channel made the kernel a
    // chain of ~1.4 us round trips (0.6 TB/s at SA2, 64 workgroups x 128 rounds at SA4)
    for (int l = 0; l < nl; l += 4) {
      if (VEC4) {
        float4 g[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
          g[u] = *reinterpret_cast<const float4 *>(
              grad_out + ((size_t)bi * c_total + c_off + l0 + min(l + u, nl - 1)) * total + e);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (l + u < nl) {
            float *row = s_acc + (size_t)(l + u) * n;
            atomicAdd(row + ii[0], g[u].x);
            atomicAdd(row + ii[1], g[u].y);
            atomicAdd(row + ii[2], g[u].z);
            atomicAdd(row + ii[3], g[u].w);
          }
        }
      } else {
        float g[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
          g[u] = grad_out[((size_t)bi * c_total + c_off + l0 + min(l + u, nl - 1)) * total + e];
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (l + u < nl) atomicAdd(s_acc + (size_t)(l + u) * n + ii[0], g[u]);
      }
    }
  }
  __syncthreads();
  float *dst = grad_points + ((size_t)bi * c + l0) * n;
  for (int i = threadIdx.x; i < nl * n; i += GP_THREADS) {
    if (atomic_out) unsafeAtomicAdd(dst + i, s_acc[i]);
    else dst[i] = s_acc[i];
  }
}

// ---- scenes stay on one XCD ---------------------------------------------------------------------
// Workgroups are dealt to the 8 XCDs round-robin by their linear id, and every XCD has an L2 of its own (4 MiB).  With
// the scene in blockIdx.z every scene's workgroups are spread over all eight L2s and each L2 sees the source rows of
// all eight scenes (8.4 MB at SA3: PMC FETCH 3.7 x the algorithmic reads).  On a 1-D grid, workgroup L (XCD L % 8) takes
// scene L % 8 (+ 8 per further round of scenes; b = 1, 2, 4: 8 / b XCDs share a scene): a scene's sources are then
// fetched into ONE L2.  A pure speed choice (guide: placement may change speed only): any other dealing gives the same
// result.  w is the workgroup's index inside its scene (0 .. per_scene - 1).
__device__ __forceinline__ void xcd_local_scene(int b, int per_scene, int &scene, int &w, int L = blockIdx.x) {
  if ((b & 7) == 0) {
    const int k = L >> 3;
    scene = (L & 7) + 8 * (k / per_scene);
    w = k % per_scene;
  } else if (b < 8 && 8 % b == 0 && per_scene % (8 / b) == 0) {
    const int x = L & 7, r = 8 / b;
    scene = x % b;
    w = (L >> 3) * r + x / b;
  } else {
    scene = L / per_scene;
    w = L - scene * per_scene;
  }
}

// Fused QueryAndGroup tail (pointnet2_utils.py:348-359): channels [0,3) = (xyz[idx] - centre)
// [/ radius], channels [3, 3+c) = features[idx]; one pass, grouped tensor written once.
// blockIdx.y == 0 handles the xyz slab (when use_xyz), the others feature slabs.
typedef float gp_f32x4 __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ void gp_store4(float *o, float a, float b, float c, float d) {
  gp_f32x4 v = {a, b, c, d};
  if (NT) __builtin_nontemporal_store(v, reinterpret_cast<gp_f32x4 *>(o));
  else *reinterpret_cast<gp_f32x4 *>(o) = v;
}

// `wg`: the workgroup's number inside this problem's grid (blockIdx.x of a launch of its own; the multi-level launch
// below hands every level a range of its grid that starts at a multiple of 8, so the XCD dealing is the same)
template <bool VEC4, bool NT>
__device__ __forceinline__ void query_group_fused_body(
    int wg, int b, int n, int m, int c, int nsample, int use_xyz, int normalize_xyz, float radius,
    const float *__restrict__ xyz, const float *__restrict__ new_xyz,
    const float *__restrict__ features, const int *__restrict__ idx, float *__restrict__ out) {
  const long total = (long)m * nsample;
  const int tiles_e = (int)(((VEC4 ? total / 4 : total) + GP_THREADS - 1) / GP_THREADS);
  const int slabs = (use_xyz ? 1 : 0) + (c + GP_CSLAB - 1) / GP_CSLAB;
  int bi, w;
  xcd_local_scene(b, tiles_e * slabs, bi, w, wg);
  if (bi >= b || w >= tiles_e * slabs) return;          // (the padding of a level's range in the multi-level launch)
  const int by = w / tiles_e, bx = w - by * tiles_e;
  const long e = ((long)bx * GP_THREADS + threadIdx.x) * (VEC4 ? 4 : 1);
  if (e >= total) return;
  const int c_total = (use_xyz ? 3 : 0) + c;
  const int *ip = idx + (size_t)bi * total + e;
  int ii[4];
  if (VEC4) {
    const int4 t = *reinterpret_cast<const int4 *>(ip);
    ii[0] = t.x; ii[1] = t.y; ii[2] = t.z; ii[3] = t.w;
  } else {
    ii[0] = *ip;
  }
  constexpr int V = VEC4 ? 4 : 1;
  int slab = by;
  if (use_xyz) {
    if (slab == 0) {
      // nsample % 4 == 0 on the vector path, so the four samples share one centre
      const int j = (int)(e / nsample);
      const float *ctr = new_xyz + ((size_t)bi * m + j) * 3;
      const float *pts = xyz + (size_t)bi * n * 3;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        float v[4];
        const float ca = ctr[a];
#pragma unroll
        for (int q = 0; q < V; ++q) {
          float t = __fsub_rn(pts[3 * (size_t)ii[q] + a], ca);  // grouped_xyz -= new_xyz  (:349)
          if (normalize_xyz) t = __fdiv_rn(t, radius);           // grouped_xyz /= radius   (:351)
          v[q] = t;
        }
        float *o = out + ((size_t)bi * c_total + a) * total + e;
        if (VEC4) gp_store4<NT>(o, v[0], v[1], v[2], v[3]);
        else *o = v[0];
      }
      return;
    }
    slab -= 1;
  }
  const int l0 = slab * GP_CSLAB;
  const int c_off = use_xyz ? 3 : 0;
  // all GP_CSLAB x V gathers of the slab are issued before the first store (channel clamped, store
  // predicated): 32 independent loads per lane instead of 4 per dependent round
  float v[GP_CSLAB][V];
#pragma unroll
  for (int t = 0; t < GP_CSLAB; ++t) {
    const float *row = features + ((size_t)bi * c + min(l0 + t, c - 1)) * n;
#pragma unroll
    for (int q = 0; q < V; ++q) v[t][q] = row[ii[q]];
  }
#pragma unroll
  for (int t = 0; t < GP_CSLAB; ++t) {
    if (l0 + t < c) {
      float *o = out + ((size_t)bi * c_total + c_off + l0 + t) * total + e;
      if (VEC4) gp_store4<NT>(o, v[t][0], v[t][1], v[t][2], v[t][3]);
      else *o = v[t][0];
    }
  }
}

template <bool VEC4, bool NT>
__global__ __launch_bounds__(GP_THREADS) void query_group_fused_kernel(
    int b, int n, int m, int c, int nsample, int use_xyz, int normalize_xyz, float radius,
    const float *__restrict__ xyz, const float *__restrict__ new_xyz,
    const float *__restrict__ features, const int *__restrict__ idx, float *__restrict__ out) {
  query_group_fused_body<VEC4, NT>((int)blockIdx.x, b, n, m, c, nsample, use_xyz, normalize_xyz, radius, xyz, new_xyz,
                                   features, idx, out);
}

// ---- point-major variant of the fused grouping -------------------------------------------------
// The (b,c,n) layout makes every grouped element a 4-byte gather per channel.  With the features
// ALSO available point-major (b,n,ld) a neighbour's channels are one contiguous row: a workgroup takes
// 64 grouped elements x 128 channels, half-waves read whole 512-byte rows (16 B per lane), the tile is
// turned through LDS (16-byte slots XOR-swizzled by the row so that both the row-wise writes and the
// column-wise reads are conflict-free) and leaves as 256-byte runs of one channel.
constexpr int GPM_P = 64;    // grouped elements per tile
constexpr int GPM_C = 128;   // channels per tile

// Compact mode (centre_of / n_act given, compact.hip): idx holds the DISTINCT neighbours of a batch element
// back to back, centre_of their centres, n_act[b] how many there are; positions >= n_act[b] are not written.
template <bool COMPACT>
__device__ __forceinline__ void query_group_fused_pm_body(
    int wg, gp_f32x4 *s_tile, int b, int n, int m, int c, int ld, int nsample, int use_xyz, int normalize_xyz, float radius,
    const float *__restrict__ xyz, const float *__restrict__ new_xyz, const float *__restrict__ feat_pm,
    const int *__restrict__ idx, float *__restrict__ out, const int *__restrict__ centre_of,
    const int *__restrict__ n_act) {
  const int stride = m * nsample;                       // positions per (batch, channel) row of out
  const int tiles_e = (stride + GPM_P - 1) / GPM_P, tiles_c = (c + GPM_C - 1) / GPM_C;
  int bi, w;
  xcd_local_scene(b, tiles_e * tiles_c, bi, w, wg);
  if (bi >= b || w >= tiles_e * tiles_c) return;        // (workgroup-uniform: the padding of a level's range)
  const int by = w / tiles_e, bx = w - by * tiles_e;    // element tile fastest: neighbours share source rows
  const int c0 = by * GPM_C;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int total = COMPACT ? n_act[bi] : stride;       // positions that exist
  const int e0 = bx * GPM_P;
  if (e0 >= total) return;
  const int c_total = (use_xyz ? 3 : 0) + c;
  const int *ip = idx + (size_t)bi * stride;
  const float *fb = feat_pm + (size_t)bi * n * ld;
  // rows: wave w, round r, half h -> tile row p = r*8 + w*2 + h; the half-wave's 32 lanes cover 128 channels
  const int half = lane >> 5, j = lane & 31;
  const bool col_ok = c0 + 4 * j < c;
  gp_f32x4 v[GPM_P / 8];
  int pi[GPM_P / 8];
#pragma unroll
  for (int r = 0; r < GPM_P / 8; ++r) {
    const int e = e0 + r * 8 + wave * 2 + half;
    pi[r] = ip[e < total ? e : total - 1];
  }
#pragma unroll
  for (int r = 0; r < GPM_P / 8; ++r)
    v[r] = *reinterpret_cast<const gp_f32x4 *>(fb + (size_t)pi[r] * ld + (col_ok ? c0 + 4 * j : 0));
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int r = 0; r < GPM_P / 8; ++r) {
    const int p = r * 8 + wave * 2 + half;
    s_tile[p * (GPM_C / 4) + (j ^ (p & 7))] = v[r];
  }
  // the xyz slab rides along with the first channel tile: lane p of wave 0 handles element e0 + p
  if (use_xyz && by == 0 && wave == 0) {
    const int e = e0 + lane;
    if (e < total) {
      const int a = ip[e], jc = COMPACT ? centre_of[(size_t)bi * stride + e] : e / nsample;
      const float *pt = xyz + ((size_t)bi * n + a) * 3;
      const float *ctr = new_xyz + ((size_t)bi * m + jc) * 3;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        float t = __fsub_rn(pt[k], ctr[k]);
        if (normalize_xyz) t = __fdiv_rn(t, radius);
        __builtin_nontemporal_store(t, out + ((size_t)bi * c_total + k) * stride + e);
      }
    }
  }
  __syncthreads();
  // columns: lane = element p, wave w walks 16-byte slots w*8 .. w*8+7 (4 channels each)
  const int p = lane, e = e0 + p;
  const int c_off = use_xyz ? 3 : 0;
  gp_f32x4 w4[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) w4[q] = s_tile[p * (GPM_C / 4) + ((wave * 8 + q) ^ (p & 7))];
  if (e < total) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int cc = c0 + 4 * (wave * 8 + q);
      float *o = out + ((size_t)bi * c_total + c_off + cc) * stride + e;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (cc + k < c) __builtin_nontemporal_store(w4[q][k], o + (size_t)k * stride);
    }
  }
}

template <bool COMPACT>
__global__ __launch_bounds__(GP_THREADS) void query_group_fused_pm_kernel(
    int b, int n, int m, int c, int ld, int nsample, int use_xyz, int normalize_xyz, float radius,
    const float *__restrict__ xyz, const float *__restrict__ new_xyz, const float *__restrict__ feat_pm,
    const int *__restrict__ idx, float *__restrict__ out, const int *__restrict__ centre_of,
    const int *__restrict__ n_act) {
  __shared__ gp_f32x4 s_tile[GPM_P * (GPM_C / 4)];
  query_group_fused_pm_body<COMPACT>((int)blockIdx.x, s_tile, b, n, m, c, ld, nsample, use_xyz, normalize_xyz, radius, xyz,
                                     new_xyz, feat_pm, idx, out, centre_of, n_act);
}

// ---- QueryAndGroup of several set-abstraction levels in ONE launch (sig3d_query_group_levels) -----------------------
// The grouping of a level is a bandwidth-sized launch of 10-30 us; four of them in a row each pay their own ramp and
// tail (4.0 TB/s over the four against 4.9 for the largest alone).  Their neighbour lists all exist before the first
// of them runs (sig3d_ball_query_levels), so one grid takes all of them: level ranges back to back, the largest first,
// every range starting at a multiple of 8 workgroups so that a scene still meets one XCD's L2 (xcd_local_scene).
struct GroupLevels {
  int levels;
  int first[5];                       // first workgroup of level l (first[levels] = grid size)
  sig3d_group_level lv[4];
};

__global__ __launch_bounds__(GP_THREADS) void query_group_levels_kernel(GroupLevels g, int b) {
  __shared__ gp_f32x4 s_tile[GPM_P * (GPM_C / 4)];
  int l = 0;
  while (l + 1 < g.levels && (int)blockIdx.x >= g.first[l + 1]) ++l;
  const sig3d_group_level &v = g.lv[l];
  const int wg = (int)blockIdx.x - g.first[l];
  if (v.point_major)
    query_group_fused_pm_body<false>(wg, s_tile, b, v.n, v.m, v.c, v.ld, v.nsample, v.use_xyz, v.normalize_xyz, v.radius,
                                     v.xyz, v.new_xyz, v.features, v.idx, v.out, nullptr, nullptr);
  else
    query_group_fused_body<true, true>(wg, b, v.n, v.m, v.c, v.nsample, v.use_xyz, v.normalize_xyz, v.radius, v.xyz,
                                       v.new_xyz, v.features, v.idx, v.out);
}

// (b,c,n) -> (b,n,c): 32x32 tiles through LDS, both sides coalesced
__global__ __launch_bounds__(256) void transpose_cn_kernel(int c, int n, const float *__restrict__ in,
                                                           float *__restrict__ out) {
  __shared__ float s_t[32][33];
  const int bi = blockIdx.z, n0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int cc = c0 + ty + 8 * k, nn = n0 + tx;
    s_t[ty + 8 * k][tx] = (cc < c && nn < n) ? in[((size_t)bi * c + cc) * n + nn] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int nn = n0 + ty + 8 * k, cc = c0 + tx;
    if (cc < c && nn < n) out[((size_t)bi * n + nn) * c + cc] = s_t[tx][ty + 8 * k];
  }
}

// ---- point-major scatter-add (backward of the fused grouping, wide levels) --------------------------
// grad_out (b,c_total,m,ns) is read in 256-byte runs per channel, turned through the same swizzled LDS
// tile, and every grouped element then adds ONE contiguous row of the point-major gradient (b,n,ld).
// A half-wave walks 8 consecutive elements and merges runs of equal indices in registers first: ball
// query pads a short neighbour list with its first hit, so most of a list is one repeated index.
__global__ __launch_bounds__(GP_THREADS) void group_points_grad_pm_kernel(
    int b, int n, int c, int ld, int stride, int c_total, int c_off, const float *__restrict__ grad_out,
    const int *__restrict__ idx, float *__restrict__ grad_pm, const int *__restrict__ n_act) {
  __shared__ gp_f32x4 s_tile[GPM_P * (GPM_C / 4)];
  const int tiles_e = (stride + GPM_P - 1) / GPM_P, tiles_c = (c + GPM_C - 1) / GPM_C;
  int bi, w;
  xcd_local_scene(b, tiles_e * tiles_c, bi, w);    // a scene's scattered rows are added in ONE L2
  const int by = w / tiles_e, bx = w - by * tiles_e;
  const int c0 = by * GPM_C;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int e0 = bx * GPM_P;
  const int total = n_act ? n_act[bi] : stride;   // compact mode: only the distinct neighbours exist
  if (e0 >= total) return;
  {
    const int p = lane, e = e0 + p;
    const bool ok = e < total;
    const float *g = grad_out + ((size_t)bi * c_total + c_off) * stride + (ok ? e : total - 1);
    float v[8][4];
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int cc = c0 + 4 * (wave * 8 + q) + k;
        v[q][k] = g[(size_t)(cc < c ? cc : c - 1) * stride];
      }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int cc = c0 + 4 * (wave * 8 + q);
      gp_f32x4 t = {(ok && cc + 0 < c) ? v[q][0] : 0.f, (ok && cc + 1 < c) ? v[q][1] : 0.f,
                    (ok && cc + 2 < c) ? v[q][2] : 0.f, (ok && cc + 3 < c) ? v[q][3] : 0.f};
      s_tile[p * (GPM_C / 4) + ((wave * 8 + q) ^ (p & 7))] = t;
    }
  }
  __syncthreads();
  const int hw = wave * 2 + (lane >> 5), j = lane & 31;
  const int *ip = idx + (size_t)bi * stride;
  int a[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int e = e0 + hw * 8 + t;
    a[t] = e < total ? ip[e] : -1;
  }
  const float *sf = reinterpret_cast<const float *>(s_tile);
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  int cur = -1;
  auto flush = [&]() {
    if (cur < 0) return;
    float *dst = grad_pm + ((size_t)bi * n + cur) * ld + c0 + j;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (c0 + j + 32 * k < c) unsafeAtomicAdd(dst + 32 * k, acc[k]);
  };
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int p = hw * 8 + t;
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {  // lane j holds channels j, j+32, j+64, j+96: 128-byte atomics runs
      const int cc = j + 32 * k;
      v[k] = sf[(p * (GPM_C / 4) + ((cc >> 2) ^ (p & 7))) * 4 + (cc & 3)];
    }
    if (a[t] != cur) {
      flush();
      cur = a[t];
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] = v[k];
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] += v[k];
    }
  }
  flush();
}

// Compact-mode grouping for narrow levels (c < 32, e.g. SA1's 3 colour channels): lane = distinct neighbour.
__global__ __launch_bounds__(GP_THREADS) void query_group_compact_kernel(
    int n, int m, int c, int stride, int use_xyz, int normalize_xyz, float radius,
    const float *__restrict__ xyz, const float *__restrict__ new_xyz, const float *__restrict__ features,
    const int *__restrict__ cidx, const int *__restrict__ centre_of, const int *__restrict__ n_act,
    float *__restrict__ out) {
  const int bi = blockIdx.y, e = blockIdx.x * GP_THREADS + threadIdx.x;
  if (e >= n_act[bi]) return;
  const int a = cidx[(size_t)bi * stride + e], jc = centre_of[(size_t)bi * stride + e];
  const int c_total = (use_xyz ? 3 : 0) + c;
  float *o = out + (size_t)bi * c_total * stride + e;
  if (use_xyz) {
    const float *pt = xyz + ((size_t)bi * n + a) * 3;
    const float *ctr = new_xyz + ((size_t)bi * m + jc) * 3;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float t = __fsub_rn(pt[k], ctr[k]);
      if (normalize_xyz) t = __fdiv_rn(t, radius);
      o[(size_t)k * stride] = t;
    }
    o += (size_t)3 * stride;
  }
  const float *f = features + (size_t)bi * c * n + a;
  for (int l = 0; l < c; ++l) o[(size_t)l * stride] = f[(size_t)l * n];
}

// ... and its backward: one atomic per (distinct neighbour, channel) into the channel-major gradient
__global__ __launch_bounds__(GP_THREADS) void query_group_compact_grad_kernel(
    int n, int c, int stride, int c_total, int c_off, const float *__restrict__ grad_out,
    const int *__restrict__ cidx, const int *__restrict__ n_act, float *__restrict__ grad_features) {
  const int bi = blockIdx.y, e = blockIdx.x * GP_THREADS + threadIdx.x;
  if (e >= n_act[bi]) return;
  const int a = cidx[(size_t)bi * stride + e];
  const float *g = grad_out + ((size_t)bi * c_total + c_off) * stride + e;
  float *d = grad_features + (size_t)bi * c * n + a;
  for (int l = 0; l < c; ++l) unsafeAtomicAdd(d + (size_t)l * n, g[(size_t)l * stride]);
}

}  // namespace

extern "C" int sig3d_group_points(int b, int c, int n, int npoints, int nsample,
                                  const float *points, const int *idx, float *out,
                                  void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SIG3D_REQUIRE(b >= 0 && c >= 0 && n >= 0 && npoints >= 0 && nsample >= 0, "negative size");
  const long total = (long)npoints * nsample;
  if (b == 0 || c == 0 || total == 0) return 0;
  SIG3D_REQUIRE(n >= 1, "group_points: n must be >= 1 when idx is non-empty");
  const bool vec = (total % 4 == 0);
  dim3 grid(sig3d_ceil_div(vec ? total / 4 : total, GP_THREADS), sig3d_ceil_div(c, GP_CSLAB), b);
  if (vec)
    hipLaunchKernelGGL((group_points_kernel<true>), grid, dim3(GP_THREADS), 0, stream, c, n, total,
                       points, idx, out);
  else
    hipLaunchKernelGGL((group_points_kernel<false>), grid, dim3(GP_THREADS), 0, stream, c, n, total,
                       points, idx, out);
  SIG3D_LAUNCH_CHECK("group_points_kernel");
  return 0;
}

static int launch_group_grad(int b, int c, int n, long total, int c_total, int c_off,
                             const float *grad_out, const int *idx, float *grad_points,
                             hipStream_t stream) {
  const bool vec = (total % 4 == 0);
  if (total > 0 && n <= GG_LDS_FLOATS) {
    int cslab = GG_LDS_FLOATS / n;
    if (cslab > c) cslab = c;
    // narrower channel slabs (down to one 4-channel round) until >= 1024 workgroups exist: idx is
    // re-read once per slab (4 B/element, cached), the flush traffic does not change
    while (cslab > 4 && (long)sig3d_ceil_div(c, cslab) * b < 1024) cslab = (cslab / 2 + 3) / 4 * 4;
    const int slabs = sig3d_ceil_div(c, cslab);
    // then split the (npoints*nsample) range, keeping at least 4 sweeps of the workgroup per block
    const long per_sweep = (long)GP_THREADS * (vec ? 4 : 1);
    long max_splits = total / (4 * per_sweep);
    if (max_splits < 1) max_splits = 1;
    long splits = (1024 + (long)slabs * b - 1) / ((long)slabs * b);
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    long e_per_block = (total + splits - 1) / splits;
    e_per_block = (e_per_block + per_sweep - 1) / per_sweep * per_sweep;  // keep 16-B alignment
    splits = (total + e_per_block - 1) / e_per_block;
    if (splits > 1)
      SIG3D_HIP_TRY(hipMemsetAsync(grad_points, 0, sizeof(float) * (size_t)b * c * n, stream));
    dim3 grid((unsigned)splits, slabs, b);
    const size_t lds = sizeof(float) * (size_t)cslab * n;
    if (vec)
      hipLaunchKernelGGL((group_points_grad_lds_kernel<true>), grid, dim3(GP_THREADS), lds, stream,
                         c, n, total, c_total, c_off, cslab, e_per_block, splits > 1 ? 1 : 0,
                         grad_out, idx, grad_points);
    else
      hipLaunchKernelGGL((group_points_grad_lds_kernel<false>), grid, dim3(GP_THREADS), lds, stream,
                         c, n, total, c_total, c_off, cslab, e_per_block, splits > 1 ? 1 : 0,
                         grad_out, idx, grad_points);
    SIG3D_LAUNCH_CHECK("group_points_grad_lds_kernel");
    return 0;
  }
  SIG3D_HIP_TRY(hipMemsetAsync(grad_points, 0, sizeof(float) * (size_t)b * c * n, stream));
  if (total == 0) return 0;
  dim3 grid(sig3d_ceil_div(vec ? total / 4 : total, GP_THREADS), sig3d_ceil_div(c, GP_CSLAB), b);
  if (vec)
    hipLaunchKernelGGL((group_points_grad_kernel<true>), grid, dim3(GP_THREADS), 0, stream, c, n,
                       total, c_total, c_off, grad_out, idx, grad_points);
  else
    hipLaunchKernelGGL((group_points_grad_kernel<false>), grid, dim3(GP_THREADS), 0, stream, c, n,
                       total, c_total, c_off, grad_out, idx, grad_points);
  SIG3D_LAUNCH_CHECK("group_points_grad_kernel");
  return 0;
}

extern "C" int sig3d_group_points_grad(int b, int c, int n, int npoints, int nsample,
                                       const float *grad_out, const int *idx,
                                       float *grad_points, void *stream_) {
  SIG3D_REQUIRE(b >= 0 && c >= 0 && n >= 0 && npoints >= 0 && nsample >= 0, "negative size");
  if (b == 0 || c == 0 || n == 0) return 0;
  return launch_group_grad(b, c, n, (long)npoints * nsample, c, 0, grad_out, idx, grad_points,
                           (hipStream_t)stream_);
}

extern "C" int sig3d_query_group_fused(int b, int n, int m, int c, int nsample, int use_xyz,
                                       int normalize_xyz, float radius, const float *xyz,
                                       const float *new_xyz, const float *features,
                                       const int *idx, float *out, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SIG3D_REQUIRE(b >= 0 && c >= 0 && n >= 0 && m >= 0 && nsample >= 0, "negative size");
  SIG3D_REQUIRE(use_xyz || c > 0, "Cannot have not features and not use xyz as a feature!");
  SIG3D_REQUIRE(c == 0 || features != nullptr, "features pointer is NULL with c > 0");
  const long total = (long)m * nsample;
  if (b == 0 || total == 0) return 0;
  SIG3D_REQUIRE(n >= 1, "query_group_fused: n must be >= 1 when idx is non-empty");
  const bool vec = (nsample % 4 == 0);
  const long blocks = (long)sig3d_ceil_div(vec ? total / 4 : total, GP_THREADS) *
                      ((use_xyz ? 1 : 0) + sig3d_ceil_div(c, GP_CSLAB)) * b;
  SIG3D_REQUIRE(blocks < (1L << 31), "query_group_fused: more than 2^31 - 1 workgroups in the 1-D grid");
  dim3 grid((unsigned)blocks);   // 1-D: xcd_local_scene
  // the vector path streams the grouped tensor out with nontemporal stores (+4 % on this kernel)
  if (vec)
    hipLaunchKernelGGL((query_group_fused_kernel<true, true>), grid, dim3(GP_THREADS), 0, stream, b, n, m, c,
                       nsample, use_xyz, normalize_xyz, radius, xyz, new_xyz, features, idx, out);
  else
    hipLaunchKernelGGL((query_group_fused_kernel<false, false>), grid, dim3(GP_THREADS), 0, stream, b, n, m, c,
                       nsample, use_xyz, normalize_xyz, radius, xyz, new_xyz, features, idx, out);
  SIG3D_LAUNCH_CHECK("query_group_fused_kernel");
  return 0;
}

extern "C" int sig3d_transpose_cn(int b, int c, int n, const float *in, float *out, void *stream_) {
  SIG3D_REQUIRE(b >= 0 && c >= 0 && n >= 0, "negative size");
  if (b == 0 || c == 0 || n == 0) return 0;
  hipLaunchKernelGGL(transpose_cn_kernel, dim3(sig3d_ceil_div(n, 32), sig3d_ceil_div(c, 32), b), dim3(256), 0,
                     (hipStream_t)stream_, c, n, in, out);
  SIG3D_LAUNCH_CHECK("transpose_cn_kernel");
  return 0;
}

static int launch_group_pm(int b, int n, int m, int c, int ld, int nsample, int use_xyz, int normalize_xyz,
                           float radius, const float *xyz, const float *new_xyz, const float *features_pm,
                           const int *idx, float *out, const int *centre_of, const int *n_act, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SIG3D_REQUIRE(b >= 0 && c >= 4 && n >= 0 && m >= 0 && nsample >= 0, "bad size (c >= 4)");
  SIG3D_REQUIRE(c % 4 == 0 && ld % 4 == 0 && ld >= c, "point-major rows: c and ld multiples of 4, ld >= c");
  SIG3D_REQUIRE(((uintptr_t)features_pm & 15) == 0, "features_pm must be 16-byte aligned");
  const long total = (long)m * nsample;
  SIG3D_REQUIRE(total < (1L << 31) - GPM_P, "m * nsample too large");
  if (b == 0 || total == 0) return 0;
  SIG3D_REQUIRE(n >= 1, "query_group_fused_pm: n must be >= 1 when idx is non-empty");
  const long blocks = (long)sig3d_ceil_div(total, GPM_P) * sig3d_ceil_div(c, GPM_C) * b;
  SIG3D_REQUIRE(blocks < (1L << 31), "query_group_fused_pm: more than 2^31 - 1 workgroups in the 1-D grid");
  dim3 grid((unsigned)blocks);   // 1-D: xcd_local_scene
  if (n_act != nullptr)
    hipLaunchKernelGGL(query_group_fused_pm_kernel<true>, grid, dim3(GP_THREADS), 0, stream, b, n, m, c, ld, nsample,
                       use_xyz, normalize_xyz, radius, xyz, new_xyz, features_pm, idx, out, centre_of, n_act);
  else
    hipLaunchKernelGGL(query_group_fused_pm_kernel<false>, grid, dim3(GP_THREADS), 0, stream, b, n, m, c, ld, nsample,
                       use_xyz, normalize_xyz, radius, xyz, new_xyz, features_pm, idx, out, centre_of, n_act);
  SIG3D_LAUNCH_CHECK("query_group_fused_pm_kernel");
  return 0;
}

extern "C" int sig3d_query_group_fused_pm(int b, int n, int m, int c, int ld, int nsample, int use_xyz,
                                          int normalize_xyz, float radius, const float *xyz,
                                          const float *new_xyz, const float *features_pm, const int *idx,
                                          float *out, void *stream_) {
  return launch_group_pm(b, n, m, c, ld, nsample, use_xyz, normalize_xyz, radius, xyz, new_xyz, features_pm, idx,
                         out, nullptr, nullptr, stream_);
}

extern "C" int sig3d_query_group_compact(int b, int n, int m, int c, int ld, int nsample, int use_xyz,
                                         int normalize_xyz, float radius, const float *xyz, const float *new_xyz,
                                         const float *features, const float *features_pm, const int *cidx,
                                         const int *centre_of, const int *n_act, float *out, void *stream_) {
  SIG3D_REQUIRE(cidx && centre_of && n_act, "compact lists missing (sig3d_compact_neighbour_lists)");
  if (features_pm != nullptr)
    return launch_group_pm(b, n, m, c, ld, nsample, use_xyz, normalize_xyz, radius, xyz, new_xyz, features_pm, cidx,
                           out, centre_of, n_act, stream_);
  SIG3D_REQUIRE(b >= 0 && c >= 0 && n >= 0 && m >= 0 && nsample >= 0, "negative size");
  SIG3D_REQUIRE(use_xyz || c > 0, "Cannot have not features and not use xyz as a feature!");
  SIG3D_REQUIRE(c == 0 || features != nullptr, "features pointer is NULL with c > 0");
  const long total = (long)m * nsample;
  if (b == 0 || total == 0) return 0;
  hipLaunchKernelGGL(query_group_compact_kernel, dim3(sig3d_ceil_div(total, GP_THREADS), b), dim3(GP_THREADS), 0,
                     (hipStream_t)stream_, n, m, c, (int)total, use_xyz, normalize_xyz, radius, xyz, new_xyz,
                     features, cidx, centre_of, n_act, out);
  SIG3D_LAUNCH_CHECK("query_group_compact_kernel");
  return 0;
}

static int launch_group_grad_pm(int b, int n, int m, int c, int ld, int nsample, int c_total, int c_off,
                                const float *grad_out, const int *idx, float *grad_features_pm,
                                const int *n_act, void *stream_, bool zeroed = false) {
  hipStream_t stream = (hipStream_t)stream_;
  SIG3D_REQUIRE(b >= 0 && c >= 1 && n >= 0 && m >= 0 && nsample >= 0 && ld >= c, "bad size");
  SIG3D_REQUIRE(c_off >= 0 && c_off + c <= c_total, "channel window out of range");
  const long total = (long)m * nsample;
  SIG3D_REQUIRE(total < (1L << 31) - GPM_P, "m * nsample too large");
  if (b == 0 || n == 0) return 0;
  if (!zeroed) SIG3D_HIP_TRY(hipMemsetAsync(grad_features_pm, 0, sizeof(float) * (size_t)b * n * ld, stream));
  if (total == 0) return 0;
  const long blocks = (long)sig3d_ceil_div(total, GPM_P) * sig3d_ceil_div(c, GPM_C) * b;
  SIG3D_REQUIRE(blocks < (1L << 31), "group_points_grad_pm: more than 2^31 - 1 workgroups in the 1-D grid");
  dim3 grid((unsigned)blocks);   // 1-D: xcd_local_scene
  hipLaunchKernelGGL(group_points_grad_pm_kernel, grid, dim3(GP_THREADS), 0, stream, b, n, c, ld, (int)total, c_total,
                     c_off, grad_out, idx, grad_features_pm, n_act);
  SIG3D_LAUNCH_CHECK("group_points_grad_pm_kernel");
  return 0;
}

extern "C" int sig3d_query_group_levels(int b, int nlevels, const sig3d_group_level *levels, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SIG3D_REQUIRE(b >= 0 && nlevels >= 1 && nlevels <= 4 && levels != nullptr, "1 to 4 levels");
  if (b == 0) return 0;
  long blocks[4];
  int order[4];
  for (int l = 0; l < nlevels; ++l) {
    const sig3d_group_level &v = levels[l];
    SIG3D_REQUIRE(v.n >= 1 && v.m >= 0 && v.c >= 0 && v.nsample >= 0, "bad level size");
    SIG3D_REQUIRE(v.use_xyz || v.c > 0, "Cannot have not features and not use xyz as a feature!");
    SIG3D_REQUIRE(v.c == 0 || v.features != nullptr, "features pointer is NULL with c > 0");
    const long total = (long)v.m * v.nsample;
    SIG3D_REQUIRE(total < (1L << 31) - GPM_P, "m * nsample too large");
    if (v.point_major) {
      SIG3D_REQUIRE(v.c >= 4 && v.c % 4 == 0 && v.ld % 4 == 0 && v.ld >= v.c && ((uintptr_t)v.features & 15) == 0,
                    "point-major rows: c and ld multiples of 4, ld >= c, 16-byte aligned");
      blocks[l] = (long)sig3d_ceil_div(total, GPM_P) * sig3d_ceil_div(v.c, GPM_C) * b;
    } else {
      SIG3D_REQUIRE(v.nsample % 4 == 0, "channel-major levels of the multi-level launch need nsample % 4 == 0");
      blocks[l] = (long)sig3d_ceil_div(total / 4, GP_THREADS) * ((v.use_xyz ? 1 : 0) + sig3d_ceil_div(v.c, GP_CSLAB)) * b;
    }
    order[l] = l;
  }
  for (int i = 1; i < nlevels; ++i)          // largest level first: the small ones fill its tail
    for (int j = i; j > 0 && blocks[order[j]] > blocks[order[j - 1]]; --j) { const int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t; }
  GroupLevels g{};
  long at = 0;
  for (int k = 0; k < nlevels; ++k) {
    g.first[k] = (int)at;
    g.lv[k] = levels[order[k]];
    at += (blocks[order[k]] + 7) / 8 * 8;
    SIG3D_REQUIRE(at < (1L << 31), "sig3d_query_group_levels: more than 2^31 - 1 workgroups");
  }
  g.levels = nlevels;
  g.first[nlevels] = (int)at;
  if (at == 0) return 0;
  hipLaunchKernelGGL(query_group_levels_kernel, dim3((unsigned)at), dim3(GP_THREADS), 0, stream, g, b);
  SIG3D_LAUNCH_CHECK("query_group_levels_kernel");
  return 0;
}

extern "C" int sig3d_query_group_fused_grad_pm(int b, int n, int m, int c, int ld, int nsample, int c_total,
                                               int c_off, const float *grad_out, const int *idx,
                                               float *grad_features_pm, void *stream_) {
  return launch_group_grad_pm(b, n, m, c, ld, nsample, c_total, c_off, grad_out, idx, grad_features_pm, nullptr,
                              stream_);
}

// grad_features_pm zeroed by the caller (scratch.py: one fill per step)
extern "C" int sig3d_query_group_fused_grad_pm_z(int b, int n, int m, int c, int ld, int nsample, int c_total,
                                                 int c_off, const float *grad_out, const int *idx,
                                                 float *grad_features_pm, void *stream_) {
  return launch_group_grad_pm(b, n, m, c, ld, nsample, c_total, c_off, grad_out, idx, grad_features_pm, nullptr,
                              stream_, true);
}

// compact mode: grad_out holds one (duplicate-summed) gradient column per distinct neighbour.
// point_major != 0: -> grad (b,n,ld) point-major (zeroed here); else -> grad (b,c,n) channel-major (zeroed here)
extern "C" int sig3d_query_group_compact_grad(int b, int n, int m, int c, int ld, int nsample, int c_total,
                                              int c_off, const float *grad_out, const int *cidx,
                                              const int *n_act, int point_major, float *grad, void *stream_) {
  SIG3D_REQUIRE(cidx && n_act, "compact lists missing (sig3d_compact_neighbour_lists)");
  if (point_major)
    return launch_group_grad_pm(b, n, m, c, ld, nsample, c_total, c_off, grad_out, cidx, grad, n_act, stream_);
  hipStream_t stream = (hipStream_t)stream_;
  SIG3D_REQUIRE(b >= 0 && c >= 1 && n >= 0 && m >= 0 && nsample >= 0, "bad size");
  SIG3D_REQUIRE(c_off >= 0 && c_off + c <= c_total, "channel window out of range");
  const long total = (long)m * nsample;
  if (b == 0 || n == 0) return 0;
  SIG3D_HIP_TRY(hipMemsetAsync(grad, 0, sizeof(float) * (size_t)b * c * n, stream));
  if (total == 0) return 0;
  hipLaunchKernelGGL(query_group_compact_grad_kernel, dim3(sig3d_ceil_div(total, GP_THREADS), b), dim3(GP_THREADS), 0,
                     stream, n, c, (int)total, c_total, c_off, grad_out, cidx, n_act, grad);
  SIG3D_LAUNCH_CHECK("query_group_compact_grad_kernel");
  return 0;
}

extern "C" int sig3d_query_group_fused_grad(int b, int n, int m, int c, int nsample,
                                            int c_total, int c_off, const float *grad_out,
                                            const int *idx, float *grad_features,
                                            void *stream_) {
  SIG3D_REQUIRE(b >= 0 && c >= 0 && n >= 0 && m >= 0 && nsample >= 0, "negative size");
  SIG3D_REQUIRE(c_off >= 0 && c_off + c <= c_total, "channel window out of range");
  if (b == 0 || c == 0 || n == 0) return 0;
  return launch_group_grad(b, c, n, (long)m * nsample, c_total, c_off, grad_out, idx,
                           grad_features, (hipStream_t)stream_);
}
