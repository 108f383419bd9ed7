// voxelize.hip -- scene voxelisation / de-duplication on the GPU.
//
// What the reference does per sample in its DataLoader workers (numpy, float64):
//   lib/sepdataset.py:286-295   optional augmentation rotations  p <- p . R^T   (one np.dot per axis)
//   lib/sepdataset.py:298-300   p <- p - p.min(0)
//   lib/openscene/voxelizer_dev.py:35-46   q = floor([p,1] @ diag(1/voxel_size)^T)
//   lib/openscene/voxelization_utils.py:9-24   key = FNV-64 over the three uint64 cells
//   lib/openscene/voxelization_utils.py:133    _, inds, inverse = np.unique(key, return_index, return_inverse)
//   voxelizer_dev.py:48         coords[inds], feats[inds], labels[inds]
// np.unique sorts the keys (unsigned), `inds` is the FIRST point of every distinct key in ascending key
// order and `inverse` the rank of every point's key.
//
// Here a whole batch of ragged scenes is one flat array with device-side offsets, and the pipeline is
//   min (two levels) -> key -> 8 x stable LSD radix pass on (key, point index) -> head flags -> ranks.
// The sort is stable and starts from ascending point indices, so the first element of every run of
// equal keys IS the first occurrence; results are bit-identical to the numpy path.
// Everything is 8..20 bytes per point per pass: HBM / launch-latency sized, no MFMA shaped work.
#include "sig3d_common.h"

namespace {

typedef unsigned long long u64;

constexpr int VX_THREADS = 256;
constexpr int VX_WAVES = VX_THREADS / 64;
constexpr int VX_ITEMS = 8;
constexpr int VX_TILE = VX_THREADS * VX_ITEMS;  // 2048 points per workgroup
constexpr int VX_MAX_ROT = 8;

struct VoxParams {
  const void *coords;      // flat (total,3), f32 or f64
  const double *rot;       // (b, n_rot, 9) row-major 3x3, applied in order as p <- R p, or NULL
  const int *offsets;      // (b+1) first point of every scene
  const int *flips;        // (b) bit a set: negate axis a before anything else (mirror augmentation), or NULL
  int is_f64, n_rot, shift_min, divide;
  double q0, q1, q2;       // per-axis scale (divide == 0) or cell size (divide == 1)
};

// One point through the reference's float pipeline.  np.dot of an (N,3) block with a 3x3 matrix is
// spelled as x*r0 then two fused multiply-adds (what the BLAS micro-kernels on the reference's hosts
// do); without rotations a float32 scene stays float32 until the voxeliser's float64 matmul.
struct VoxPoint {
  double x, y, z;
  bool f32;  // still float32-valued (x,y,z hold exact float32 values)
};

__device__ __forceinline__ VoxPoint vox_load(const VoxParams &p, int scene, long gi) {
  VoxPoint v;
  if (p.is_f64) {
    const double *c = (const double *)p.coords + 3 * gi;
    v.x = c[0]; v.y = c[1]; v.z = c[2];
    v.f32 = false;
  } else {
    const float *c = (const float *)p.coords + 3 * gi;
    v.x = (double)c[0]; v.y = (double)c[1]; v.z = (double)c[2];
    v.f32 = true;
  }
  if (p.flips) {  // sepdataset.py:246,255: exact sign change, the scene keeps its dtype
    const int f = p.flips[scene];
    v.x = (f & 1) ? -v.x : v.x; v.y = (f & 2) ? -v.y : v.y; v.z = (f & 4) ? -v.z : v.z;
  }
  for (int k = 0; k < p.n_rot; ++k) {
    const double *r = p.rot + ((long)scene * p.n_rot + k) * 9;
    const double nx = __fma_rn(v.z, r[2], __fma_rn(v.y, r[1], __dmul_rn(v.x, r[0])));
    const double ny = __fma_rn(v.z, r[5], __fma_rn(v.y, r[4], __dmul_rn(v.x, r[3])));
    const double nz = __fma_rn(v.z, r[8], __fma_rn(v.y, r[7], __dmul_rn(v.x, r[6])));
    v.x = nx; v.y = ny; v.z = nz;
    v.f32 = false;
  }
  return v;
}

__device__ __forceinline__ double vox_cell(double v, double mn, bool f32, const VoxParams &p, double q) {
  if (p.shift_min) v = f32 ? (double)__fsub_rn((float)v, (float)mn) : __dsub_rn(v, mn);
  return floor(p.divide ? __ddiv_rn(v, q) : __dmul_rn(v, q));
}

// voxelization_utils.py:9-24 -- multiply first, then xor (the reference's "FNV64-1A")
__device__ __forceinline__ u64 vox_fnv3(long a, long b, long c) {
  u64 h = 14695981039346656037ull;
  h *= 1099511628211ull; h ^= (u64)a;
  h *= 1099511628211ull; h ^= (u64)b;
  h *= 1099511628211ull; h ^= (u64)c;
  return h;
}

__device__ __forceinline__ double wave_min_f64(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_xor(v, off));
  return v;
}

// ---- per-scene minimum of the (rotated) coordinates: tile partials, then one workgroup per scene ----
__global__ __launch_bounds__(VX_THREADS) void vox_min_tile_kernel(VoxParams p, int tiles,
                                                                  double *__restrict__ part) {
  __shared__ double s_m[VX_WAVES][3];
  const int scene = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x;
  const int seg = p.offsets[scene], n = p.offsets[scene + 1] - seg;
  double mx = INFINITY, my = INFINITY, mz = INFINITY;
  for (int j = 0; j < VX_ITEMS; ++j) {
    const int e = tile * VX_TILE + j * VX_THREADS + tid;
    if (e < n) {
      const VoxPoint v = vox_load(p, scene, (long)seg + e);
      mx = fmin(mx, v.x); my = fmin(my, v.y); mz = fmin(mz, v.z);
    }
  }
  mx = wave_min_f64(mx); my = wave_min_f64(my); mz = wave_min_f64(mz);
  if ((tid & 63) == 0) { s_m[tid >> 6][0] = mx; s_m[tid >> 6][1] = my; s_m[tid >> 6][2] = mz; }
  __syncthreads();
  if (tid < 3) {
    double m = s_m[0][tid];
    for (int w = 1; w < VX_WAVES; ++w) m = fmin(m, s_m[w][tid]);
    part[((long)scene * tiles + tile) * 3 + tid] = m;
  }
}

__global__ __launch_bounds__(64) void vox_min_scene_kernel(int tiles, const double *__restrict__ part,
                                                           double *__restrict__ mins) {
  const int scene = blockIdx.x, lane = threadIdx.x;
  double m[3] = {INFINITY, INFINITY, INFINITY};
  for (int t = lane; t < tiles; t += 64)
    for (int a = 0; a < 3; ++a) m[a] = fmin(m[a], part[((long)scene * tiles + t) * 3 + a]);
  for (int a = 0; a < 3; ++a) {
    const double r = wave_min_f64(m[a]);
    if (lane == 0) mins[scene * 3 + a] = r;
  }
}

// ---- keys ------------------------------------------------------------------------------------
__global__ __launch_bounds__(VX_THREADS) void vox_key_kernel(VoxParams p, const double *__restrict__ mins,
                                                             u64 *__restrict__ keys, int *__restrict__ vals) {
  const int scene = blockIdx.y;
  const int seg = p.offsets[scene], n = p.offsets[scene + 1] - seg;
  const int e = blockIdx.x * VX_THREADS + threadIdx.x;
  if (e >= n) return;
  const VoxPoint v = vox_load(p, scene, (long)seg + e);
  const double m0 = p.shift_min ? mins[scene * 3 + 0] : 0.0, m1 = p.shift_min ? mins[scene * 3 + 1] : 0.0,
               m2 = p.shift_min ? mins[scene * 3 + 2] : 0.0;
  const long a = (long)vox_cell(v.x, m0, v.f32, p, p.q0);
  const long b = (long)vox_cell(v.y, m1, v.f32, p, p.q1);
  const long c = (long)vox_cell(v.z, m2, v.f32, p, p.q2);
  keys[(long)seg + e] = vox_fnv3(a, b, c);
  vals[(long)seg + e] = e;
}

// voxelization_utils.py:9-24 on its own: (n, d) int64 cells -> uint64 keys
__global__ __launch_bounds__(256) void fnv_hash_vec_kernel(long n, int d, const long *__restrict__ arr,
                                                           u64 *__restrict__ out) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  u64 h = 14695981039346656037ull;
  for (int j = 0; j < d; ++j) { h *= 1099511628211ull; h ^= (u64)arr[i * d + j]; }
  out[i] = h;
}

// ---- stable LSD radix sort, 8-bit digits, segmented by scene ---------------------------------------
// element order inside a tile: wave-major, then round, then lane (coalesced 512-byte rounds)
__device__ __forceinline__ int vx_elem(int tile, int wave, int r, int lane) {
  return tile * VX_TILE + wave * (64 * VX_ITEMS) + r * 64 + lane;
}

__global__ __launch_bounds__(VX_THREADS) void vox_hist_kernel(const int *__restrict__ offsets, int tiles, int shift,
                                                              const u64 *__restrict__ keys,
                                                              int *__restrict__ counts) {
  __shared__ int s_hist[256];
  const int scene = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int seg = offsets[scene], n = offsets[scene + 1] - seg;
  s_hist[tid] = 0;
  __syncthreads();
  u64 k[VX_ITEMS];
#pragma unroll
  for (int r = 0; r < VX_ITEMS; ++r) {
    const int e = vx_elem(tile, wave, r, lane);
    k[r] = keys[(long)seg + (e < n ? e : 0)];
  }
#pragma unroll
  for (int r = 0; r < VX_ITEMS; ++r)
    if (vx_elem(tile, wave, r, lane) < n) atomicAdd(&s_hist[(int)((k[r] >> shift) & 255u)], 1);
  __syncthreads();
  counts[((long)scene * 256 + tid) * tiles + tile] = s_hist[tid];
}

// one wave per (scene, digit): exclusive scan over the tiles of that digit's row, in place, and the
// row total; the scatter kernel turns the 256 totals into digit bases itself
__global__ __launch_bounds__(64) void vox_rowscan_kernel(int tiles, int *__restrict__ counts,
                                                         int *__restrict__ digit_tot) {
  const int row = blockIdx.x, lane = threadIdx.x;  // row = scene * 256 + digit
  int *c = counts + (long)row * tiles;
  int carry = 0;
  for (int t0 = 0; t0 < tiles; t0 += 64) {
    const int t = t0 + lane;
    const int v = t < tiles ? c[t] : 0;
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int o = __shfl_up(incl, off);
      if (lane >= off) incl += o;
    }
    if (t < tiles) c[t] = carry + incl - v;
    carry += __shfl(incl, 63);
  }
  if (lane == 0) digit_tot[row] = carry;
}

__global__ __launch_bounds__(VX_THREADS) void vox_scatter_kernel(const int *__restrict__ offsets, int tiles,
                                                                 int shift, const u64 *__restrict__ keys_in,
                                                                 const int *__restrict__ vals_in,
                                                                 const int *__restrict__ starts,
                                                                 const int *__restrict__ digit_tot,
                                                                 u64 *__restrict__ keys_out,
                                                                 int *__restrict__ vals_out) {
  __shared__ int s_cnt[VX_WAVES][256];
  __shared__ int s_base[VX_WAVES][256];
  __shared__ int s_dig[VX_WAVES];
  const int scene = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int seg = offsets[scene], n = offsets[scene + 1] - seg;
  if (tile * VX_TILE >= n) return;  // uniform
  for (int w = 0; w < VX_WAVES; ++w) s_cnt[w][tid] = 0;
  u64 k[VX_ITEMS];
  int v[VX_ITEMS], rank[VX_ITEMS];
#pragma unroll
  for (int r = 0; r < VX_ITEMS; ++r) {
    const int e = vx_elem(tile, wave, r, lane);
    const long g = (long)seg + (e < n ? e : 0);
    k[r] = keys_in[g];
    v[r] = vals_in[g];
  }
  __syncthreads();
  volatile int *cnt = s_cnt[wave];
  const u64 lt = (1ull << lane) - 1ull;
#pragma unroll
  for (int r = 0; r < VX_ITEMS; ++r) {
    const bool valid = vx_elem(tile, wave, r, lane) < n;
    const int d = (int)((k[r] >> shift) & 255u);
    u64 peers = __ballot(valid);
#pragma unroll
    for (int bit = 0; bit < 8; ++bit) {
      const bool on = (d >> bit) & 1;
      const u64 bal = __ballot(on);
      peers &= on ? bal : ~bal;
    }
    const int before = __popcll(peers & lt);
    const int pre = cnt[d];                 // every peer reads the running count ...
    __builtin_amdgcn_wave_barrier();
    if (valid && before == 0) cnt[d] = pre + __popcll(peers);  // ... then the run's first lane bumps it
    __builtin_amdgcn_wave_barrier();
    rank[r] = pre + before;
  }
  // digit base = totals of all smaller digits (block scan of the 256 row totals)
  const int dtot = digit_tot[scene * 256 + tid];
  int dincl = dtot;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int o = __shfl_up(dincl, off);
    if (lane >= off) dincl += o;
  }
  if (lane == 63) s_dig[wave] = dincl;
  __syncthreads();
  {
    int run = starts[((long)scene * 256 + tid) * tiles + tile] + dincl - dtot;
    for (int w = 0; w < wave; ++w) run += s_dig[w];
    for (int w = 0; w < VX_WAVES; ++w) {
      s_base[w][tid] = run;
      run += s_cnt[w][tid];
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < VX_ITEMS; ++r) {
    if (vx_elem(tile, wave, r, lane) < n) {
      const int d = (int)((k[r] >> shift) & 255u);
      const long g = (long)seg + s_base[wave][d] + rank[r];
      keys_out[g] = k[r];
      vals_out[g] = v[r];
    }
  }
}

// ---- runs of equal keys -> first occurrences and ranks -----------------------------------------
__device__ __forceinline__ bool vx_head(const u64 *keys, long seg, int e) {
  return e == 0 || keys[seg + e] != keys[seg + e - 1];
}

__global__ __launch_bounds__(VX_THREADS) void vox_heads_kernel(const int *__restrict__ offsets, int tiles,
                                                               const u64 *__restrict__ keys,
                                                               int *__restrict__ tile_heads) {
  __shared__ int s_w[VX_WAVES];
  const int scene = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x;
  const int seg = offsets[scene], n = offsets[scene + 1] - seg;
  int c = 0;
#pragma unroll
  for (int j = 0; j < VX_ITEMS; ++j) {
    const int e = tile * VX_TILE + j * VX_THREADS + tid;
    if (e < n && vx_head(keys, seg, e)) ++c;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
  if ((tid & 63) == 0) s_w[tid >> 6] = c;
  __syncthreads();
  if (tid == 0) {
    int t = 0;
    for (int w = 0; w < VX_WAVES; ++w) t += s_w[w];
    tile_heads[(long)scene * tiles + tile] = t;
  }
}

__global__ __launch_bounds__(64) void vox_tile_scan_kernel(int tiles, int *__restrict__ tile_heads,
                                                           int *__restrict__ num_unique) {
  const int scene = blockIdx.x, lane = threadIdx.x;
  int *h = tile_heads + (long)scene * tiles;
  int carry = 0;
  for (int t0 = 0; t0 < tiles; t0 += 64) {
    const int t = t0 + lane;
    const int v = t < tiles ? h[t] : 0;
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int o = __shfl_up(incl, off);
      if (lane >= off) incl += o;
    }
    if (t < tiles) h[t] = carry + incl - v;
    carry += __shfl(incl, 63);
  }
  if (lane == 0) num_unique[scene] = carry;
}

__global__ __launch_bounds__(VX_THREADS) void vox_emit_kernel(
    VoxParams p, int tiles, const double *__restrict__ mins, const u64 *__restrict__ keys,
    const int *__restrict__ vals, const int *__restrict__ tile_off, int c_feat, const float *__restrict__ feats,
    const int *__restrict__ labels, int *__restrict__ inds, int *__restrict__ inverse, int *__restrict__ vox,
    float *__restrict__ feats_out, int *__restrict__ labels_out) {
  __shared__ int s_cnt[VX_ITEMS][VX_WAVES];
  const int scene = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int seg = p.offsets[scene], n = p.offsets[scene + 1] - seg;
  if (tile * VX_TILE >= n) return;
  // element order: round-major, then thread (coalesced); all key / index loads first
  u64 k[VX_ITEMS], kp[VX_ITEMS];
  int src[VX_ITEMS];
#pragma unroll
  for (int r = 0; r < VX_ITEMS; ++r) {
    const int e = tile * VX_TILE + r * VX_THREADS + tid;
    const long g = (long)seg + (e < n ? e : 0);
    k[r] = keys[g];
    kp[r] = keys[g > seg ? g - 1 : g];
    src[r] = vals[g];
  }
  __builtin_amdgcn_sched_barrier(0);
  const u64 lt = (1ull << lane) - 1ull;
  bool head[VX_ITEMS];
  int before[VX_ITEMS];
#pragma unroll
  for (int r = 0; r < VX_ITEMS; ++r) {
    const int e = tile * VX_TILE + r * VX_THREADS + tid;
    head[r] = e < n && (e == 0 || k[r] != kp[r]);
    const u64 bal = __ballot(head[r]);
    before[r] = __popcll(bal & lt);
    if (lane == 0) s_cnt[r][wave] = __popcll(bal);
  }
  __syncthreads();
  const double m0 = p.shift_min ? mins[scene * 3 + 0] : 0.0, m1 = p.shift_min ? mins[scene * 3 + 1] : 0.0,
               m2 = p.shift_min ? mins[scene * 3 + 2] : 0.0;
  int rank[VX_ITEMS];
  {
    int run = tile_off[(long)scene * tiles + tile];
#pragma unroll
    for (int r = 0; r < VX_ITEMS; ++r) {
      int mine = run;
#pragma unroll
      for (int w = 0; w < VX_WAVES; ++w) {
        const int c = s_cnt[r][w];
        mine += w < wave ? c : 0;
        run += c;
      }
      rank[r] = mine + before[r] + (head[r] ? 1 : 0) - 1;  // heads up to and including e, minus one
    }
  }
  // gathers of the kept points: loads of all rounds before any store
  VoxPoint pt[VX_ITEMS];
  int lab[VX_ITEMS];
#pragma unroll
  for (int r = 0; r < VX_ITEMS; ++r) {
    const long gs = (long)seg + (head[r] ? src[r] : 0);
    if (vox) pt[r] = vox_load(p, scene, gs);
    lab[r] = labels_out ? labels[gs] : 0;
  }
#pragma unroll
  for (int r = 0; r < VX_ITEMS; ++r) {
    const int e = tile * VX_TILE + r * VX_THREADS + tid;
    if (e < n) inverse[(long)seg + src[r]] = rank[r];
    if (head[r]) {
      const long o = (long)seg + rank[r];
      inds[o] = src[r];
      if (vox) {
        vox[3 * o + 0] = (int)(long)vox_cell(pt[r].x, m0, pt[r].f32, p, p.q0);
        vox[3 * o + 1] = (int)(long)vox_cell(pt[r].y, m1, pt[r].f32, p, p.q1);
        vox[3 * o + 2] = (int)(long)vox_cell(pt[r].z, m2, pt[r].f32, p, p.q2);
      }
      if (labels_out) labels_out[o] = lab[r];
    }
  }
  if (feats_out) {
#pragma unroll
    for (int r = 0; r < VX_ITEMS; ++r) {
      if (head[r]) {
        const float *fs = feats + ((long)seg + src[r]) * c_feat;
        float *fd = feats_out + ((long)seg + rank[r]) * c_feat;
        if (c_feat == 3) {
          const float a = fs[0], b2 = fs[1], c2 = fs[2];
          fd[0] = a; fd[1] = b2; fd[2] = c2;
        } else {
          for (int c = 0; c < c_feat; ++c) fd[c] = fs[c];
        }
      }
    }
  }
}

struct VoxWorkspace {
  u64 *keys[2];
  int *vals[2];
  int *counts, *digit_tot, *tile_heads;
  double *min_part;
  size_t bytes;
};

VoxWorkspace vox_layout(void *base, long b, long total, int tiles) {
  VoxWorkspace w;
  char *p = (char *)base;
  auto take = [&](size_t nbytes) {
    char *r = p;
    p += (nbytes + 255) & ~(size_t)255;
    return r;
  };
  w.keys[0] = (u64 *)take(sizeof(u64) * total);
  w.keys[1] = (u64 *)take(sizeof(u64) * total);
  w.vals[0] = (int *)take(sizeof(int) * total);
  w.vals[1] = (int *)take(sizeof(int) * total);
  w.counts = (int *)take(sizeof(int) * b * 256 * tiles);
  w.digit_tot = (int *)take(sizeof(int) * b * 256);
  w.tile_heads = (int *)take(sizeof(int) * b * tiles);
  w.min_part = (double *)take(sizeof(double) * b * tiles * 3);
  w.bytes = (size_t)(p - (char *)base);
  return w;
}

}  // namespace

extern "C" long sig3d_voxelize_workspace_bytes(int b, long total, int max_n) {
  if (b <= 0 || total <= 0 || max_n <= 0) return 256;
  return (long)vox_layout(nullptr, b, total, sig3d_ceil_div(max_n, VX_TILE)).bytes;
}

extern "C" int sig3d_fnv_hash_vec(long n, int d, const long *arr, unsigned long long *out, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SIG3D_REQUIRE(n >= 0 && d >= 1, "bad shape");
  if (n == 0) return 0;
  hipLaunchKernelGGL(fnv_hash_vec_kernel, dim3(sig3d_ceil_div(n, 256)), dim3(256), 0, stream, n, d, arr, out);
  SIG3D_LAUNCH_CHECK("fnv_hash_vec_kernel");
  return 0;
}

extern "C" int sig3d_voxelize(int b, int max_n, const int *offsets, const void *coords, int coords_f64,
                              int n_rot, const double *rot, const int *flips, int shift_min, int divide,
                              const double *quant, int c_feat, const float *feats, const int *labels, int *inds, int *inverse,
                              int *num_unique, int *vox, float *feats_out, int *labels_out, double *mins,
                              void *workspace, long workspace_bytes, long total, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SIG3D_REQUIRE(b >= 0 && max_n >= 0 && total >= 0, "negative size");
  SIG3D_REQUIRE(n_rot >= 0 && n_rot <= VX_MAX_ROT && (n_rot == 0 || rot), "0..8 rotations, rot required when n_rot > 0");
  SIG3D_REQUIRE(total < (1L << 31) - VX_TILE, "more than 2^31 points");
  SIG3D_REQUIRE(c_feat >= 0 && (!feats_out || feats) && (!labels_out || labels), "gather sources missing");
  if (b == 0) return 0;
  if (total == 0 || max_n == 0) {
    SIG3D_HIP_TRY(hipMemsetAsync(num_unique, 0, sizeof(int) * b, stream));
    return 0;
  }
  SIG3D_REQUIRE(quant && offsets && coords && inds && inverse && num_unique && mins, "null argument");
  const int tiles = sig3d_ceil_div(max_n, VX_TILE);
  const VoxWorkspace w = vox_layout(workspace, b, total, tiles);
  SIG3D_REQUIRE(workspace && (size_t)workspace_bytes >= w.bytes, "workspace too small (sig3d_voxelize_workspace_bytes)");

  VoxParams p;
  p.coords = coords; p.rot = rot; p.offsets = offsets; p.flips = flips;
  p.is_f64 = coords_f64; p.n_rot = n_rot; p.shift_min = shift_min; p.divide = divide;
  p.q0 = quant[0]; p.q1 = quant[1]; p.q2 = quant[2];
  const dim3 tgrid(tiles, b), blk(VX_THREADS);
  // the minimum is reported even when it is not subtracted (voxelizer_dev.py:45 asserts on it)
  hipLaunchKernelGGL(vox_min_tile_kernel, tgrid, blk, 0, stream, p, tiles, w.min_part);
  hipLaunchKernelGGL(vox_min_scene_kernel, dim3(b), dim3(64), 0, stream, tiles, w.min_part, mins);
  hipLaunchKernelGGL(vox_key_kernel, dim3(tiles * VX_ITEMS, b), blk, 0, stream, p, mins, w.keys[0], w.vals[0]);
  int cur = 0;
  for (int pass = 0; pass < 8; ++pass) {
    const int shift = 8 * pass;
    hipLaunchKernelGGL(vox_hist_kernel, tgrid, blk, 0, stream, offsets, tiles, shift, w.keys[cur], w.counts);
    hipLaunchKernelGGL(vox_rowscan_kernel, dim3(b * 256), dim3(64), 0, stream, tiles, w.counts, w.digit_tot);
    hipLaunchKernelGGL(vox_scatter_kernel, tgrid, blk, 0, stream, offsets, tiles, shift, w.keys[cur], w.vals[cur],
                       w.counts, w.digit_tot, w.keys[cur ^ 1], w.vals[cur ^ 1]);
    cur ^= 1;
  }
  hipLaunchKernelGGL(vox_heads_kernel, tgrid, blk, 0, stream, offsets, tiles, w.keys[cur], w.tile_heads);
  hipLaunchKernelGGL(vox_tile_scan_kernel, dim3(b), dim3(64), 0, stream, tiles, w.tile_heads, num_unique);
  hipLaunchKernelGGL(vox_emit_kernel, tgrid, blk, 0, stream, p, tiles, mins, w.keys[cur], w.vals[cur], w.tile_heads,
                     c_feat, feats, labels, inds, inverse, vox, feats_out, labels_out);
  SIG3D_LAUNCH_CHECK("voxelize kernels");
  return 0;
}
