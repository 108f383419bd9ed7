// ball_query.hip -- radius neighbour search for gfx950.
//
// Replaces lib/pointnet2/_ext_src/src/ball_query_gpu.cu of the reference: for every centre,
// the first `nsample` points IN INDEX ORDER with d2 < radius*radius (strict, f32), the row
// padded with the first hit, all-zero when there is none (ball_query_gpu.cu:27-42,
// ball_query.cpp:19-21).
//
// The reference gives one THREAD a centre and lets it walk all n points serially.  Here a
// WAVEFRONT walks the scene 64 points at a time (one point per lane, centre coordinates in
// SGPRs), a 64-bit ballot marks the hits of the chunk, and popcount of the lower lanes gives
// each hit its slot -- which preserves index order by construction.  Each wave serves CPW
// centres from the same 64-point register tile, so the scene is read from L2 once per CPW
// centres instead of once per centre.  The kernel writes every element of idx itself
// (padding and the no-hit zero row included): no memset pass is needed.
#include "sig3d_common.h"

namespace {

constexpr int BQ_WAVES = 4;  // waves per workgroup

// One wave, CPW centres from c0 on, the scene's points 64 at a time from `xyz` (global memory, or an LDS copy of
// the scene: the caller's pointer decides after inlining); new_xyz / idx are the scene's centres and rows.
template <int CPW>
__device__ __forceinline__ void bq_scan_wave(int n, int m, float radius2, int nsample,
                                             const float *__restrict__ new_xyz, const float *__restrict__ xyz,
                                             int *__restrict__ idx, int c0, int lane, int *tests = nullptr) {
  float cx[CPW], cy[CPW], cz[CPW];
  int cnt[CPW], first[CPW];
#pragma unroll
  for (int c = 0; c < CPW; ++c) {
    const int j = min(c0 + c, m - 1);  // clamp: out-of-range slots repeat the last centre
    cx[c] = new_xyz[3 * j + 0];
    cy[c] = new_xyz[3 * j + 1];
    cz[c] = new_xyz[3 * j + 2];
    cnt[c] = (c0 + c < m) ? 0 : nsample;  // out-of-range slots start "full"
    first[c] = 0;
  }

  for (int k0 = 0; k0 < n; k0 += 64) {
    const int k = k0 + lane;
    const bool inb = k < n;
    const int kk = inb ? k : n - 1;
    const float x = xyz[3 * kk + 0], y = xyz[3 * kk + 1], z = xyz[3 * kk + 2];
    int open = 0;
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
      // ball_query_gpu.cu:31-32: (new_x - x)^2 + (new_y - y)^2 + (new_z - z)^2, unfused
      const float d2 = sq_dist3(cx[c], cy[c], cz[c], x, y, z);
      const bool hit = inb && (d2 < radius2);
      const unsigned long long mask = __ballot(hit);
      if (mask != 0ull && cnt[c] < nsample) {  // wave-uniform branch
        const int before = __builtin_popcountll(mask & ((1ull << lane) - 1ull));
        const int pos = cnt[c] + before;
        if (hit && pos < nsample) idx[(size_t)(c0 + c) * nsample + pos] = k;
        if (cnt[c] == 0) first[c] = k0 + __builtin_ctzll(mask);
        cnt[c] += __builtin_popcountll(mask);
      }
      open |= (cnt[c] < nsample);
    }
    if (tests) *tests += CPW;   // per lane: CPW tests of this 64-point step
    if (!open) break;  // every centre of this wave is full (ball_query_gpu.cu:27 cnt < nsample)
  }

  // padding with the first hit (ball_query_gpu.cu:34-38) / zero row when no hit
#pragma unroll
  for (int c = 0; c < CPW; ++c) {
    if (c0 + c >= m) break;
    const int have = min(cnt[c], nsample);
    const int fill = (cnt[c] == 0) ? 0 : first[c];
    for (int l = have + lane; l < nsample; l += 64) idx[(size_t)(c0 + c) * nsample + l] = fill;
  }
}

template <int CPW>
__global__ __launch_bounds__(BQ_WAVES * 64) void ball_query_scan_kernel(
    int n, int m, float radius2, int nsample, const float *__restrict__ new_xyz_all,
    const float *__restrict__ xyz_all, int *__restrict__ idx_all) {
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int bi = blockIdx.y;
  const int c0 = (blockIdx.x * BQ_WAVES + wave) * CPW;  // first centre of this wave
  if (c0 >= m) return;
  bq_scan_wave<CPW>(n, m, radius2, nsample, new_xyz_all + (size_t)bi * m * 3, xyz_all + (size_t)bi * n * 3,
                    idx_all + (size_t)bi * m * nsample, c0, lane_id());
}


// ---- centres binned into cells, points streamed once (all sizes; several levels per launch) -------------
// Brute force costs n distance tests per centre (655 M at SA1, 0.4 ms).  Round 1 hashed the POINTS into a
// grid (histogram, two scan passes, scatter, a 27-bucket walk per centre: five launches, 43 MB of scratch
// written and the sorted copy re-read seven times over).  The centres are the small side (2048 of 40 000 at
// SA1), so they are binned instead and every point is read exactly once, in index order, coalesced:
//   scatter : a workgroup builds the table of ITS scene's centres in LDS -- cells of edge e = 2.02 r hashed
//             into H >= 2m buckets, counting sort of {x, y, z, centre} records (96 KB at most) -- and streams
//             a chunk of the scene's points past it: a point within r of a centre lies in the 2 x 2 x 2 cell
//             block [cell(p - 1.005 r), cell(p + 1.005 r)] (2 * 1.005 r < e, so each axis spans one or two cells;
//             the 0.5 % margin absorbs the rounding of v / e for |v| up to ~4e4 radii), so it tests the
//             centres of at most eight buckets (colliding cells only add candidates, buckets met twice are
//             walked once) with the reference's strict f32 test and APPENDS itself to every centre it hits
//             (hits counted per centre in LDS, one global atomic per centre and chunk reserves the list range,
//             a second pass places the hits; hits beyond BQC_CAP are only counted);
//   rank    : 16 lanes per centre put the (unordered) hit list back into INDEX ORDER by counting smaller
//             indices -- indices are unique -- keep the first nsample, pad with the smallest, write the zero
//             row of a centre without hits; a centre with more than BQC_CAP hits (dense clusters, zero-padded
//             tails) runs the ordered 64-point scan of the brute-force kernel.  Bit-identical output.
// Several (xyz, new_xyz, radius, nsample) problems -- the levels of a set-abstraction stack, or blocks of
// 4096 centres of one large problem -- share ONE scatter and ONE rank launch (+ one memset of the counters):
// geometry.GeometryPlan issues the ball queries of SA1-4 as three graph nodes instead of fourteen.
constexpr int BQC_CAP = 256;       // hits kept per centre before falling back to the ordered scan
constexpr int BQC_MAXM = 4096;     // centres per table: 64 KB of records + 32 KB of bucket ends + 32 KB of counters in LDS
constexpr int BQC_THREADS = 512;   // scatter kernel
constexpr int BQC_CPT = BQC_MAXM / BQC_THREADS;
constexpr int BQC_MAXLV = 16;      // problems (levels x centre blocks) per launch
constexpr int BQC_HITS = 2048;      // hits a cell-role workgroup notes in LDS (8 KB) before it falls back to a second pass
constexpr int BQC_PPT = 2;          // points per thread of a cell-role workgroup (1024 points per workgroup)
constexpr int BQC_SCAN_MAXN = 4096; // scenes up to this size take the ordered scan from an LDS copy (48 KB)
constexpr int BQC_RANK_CENTRES = 16;   // per rank workgroup: 4 waves x 4 groups of 16 lanes

// Phase timing for tools/bq_timing.py (compiled in only with -DSIG3D_BQ_TIMING): wave 0 of the FIRST cell-role
// workgroup stores the 100 MHz real-time counter at the marks.
#ifdef SIG3D_BQ_TIMING
__device__ unsigned long long g_bq_marks[16];
#define BQ_MARK(id)                                                                            \
  do {                                                                                         \
    if (local == 0 && threadIdx.x == 0) g_bq_marks[id] = __builtin_amdgcn_s_memrealtime();     \
  } while (0)
#else
#define BQ_MARK(id) do { } while (0)
#endif

struct BqcLevel {
  const float *xyz;       // (b, n, 3)
  const float *new_xyz;   // (b, m_total, 3)
  int *idx;               // (b, m_total, nsample)
  int n, m, m_total, c_off, nsample, hsize;
  float radius2, inv_e, reach;
  int kind;               // 0: cell table + point stream (+ rank kernel);  1: ordered scan from an LDS copy of the scene
  int cpw;                // kind 1: centres per wave
  int chunks;             // scatter workgroups per scene
  int wg_begin;           // first scatter workgroup of the level
  int rank_wg_begin;      // first rank workgroup of the level
  int ctr_begin;          // first slot of the level in cnt / list (slot = bi * m + j)
};
struct BqcParams {
  int nlevels, b;
  unsigned long long *stats;   // or null: [0] += distance tests of the scatter kernel (bench.py's roofline_ball_query)
  BqcLevel lv[BQC_MAXLV];
};

__device__ __forceinline__ int bqc_cell(float v, float inv_e) {
  const float f = floorf(v * inv_e);
  return (int)fminf(fmaxf(f, -2097152.f), 2097152.f);
}
__device__ __forceinline__ unsigned bqc_hash(int ix, int iy, int iz, unsigned hmask) {
  return ((unsigned)ix * 73856093u ^ (unsigned)iy * 19349663u ^ (unsigned)iz * 83492791u) & hmask;
}

// The table of a scene's centres: hashed cells of edge 2.02 r, a counting sort of {x, y, z, centre} records in LDS
// (s_end[h] = END of bucket h = start of bucket h + 1 afterwards).  Ends with a barrier.
__device__ __forceinline__ void bqc_build_table(int *s_end, float4 *s_ctr, int *s_wave, const float *__restrict__ ctr,
                                                int m, int H, float inv_e, unsigned hmask, int tid, int lane, int wave) {
  for (int i = tid; i < H; i += BQC_THREADS) s_end[i] = 0;
  __syncthreads();
  float cx[BQC_CPT], cy[BQC_CPT], cz[BQC_CPT];
  unsigned hs[BQC_CPT];
#pragma unroll
  for (int c = 0; c < BQC_CPT; ++c) {
    const int j = tid + c * BQC_THREADS;
    if (j < m) {
      cx[c] = ctr[3 * j + 0];
      cy[c] = ctr[3 * j + 1];
      cz[c] = ctr[3 * j + 2];
      hs[c] = bqc_hash(bqc_cell(cx[c], inv_e), bqc_cell(cy[c], inv_e), bqc_cell(cz[c], inv_e), hmask);
      atomicAdd(&s_end[hs[c]], 1);
    }
  }
  __syncthreads();
  {  // exclusive prefix of the H counts, in place: a contiguous run per thread, wave scan, wave offsets
    const int per = (H + BQC_THREADS - 1) / BQC_THREADS;   // <= 16 (H <= 8192)
    const int e0 = min(tid * per, H), e1 = min(e0 + per, H);
    int tot = 0, cv[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) cv[i] = s_end[min(e0 + i, H - 1)];   // unconditional: sixteen reads in flight
#pragma unroll
    for (int i = 0; i < 16; ++i) tot += (e0 + i < e1) ? cv[i] : 0;
    int incl = tot;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int o = __shfl_up(incl, off);
      if (lane >= off) incl += o;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int run = incl - tot;
    for (int w = 0; w < wave; ++w) run += s_wave[w];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (e0 + i < e1) s_end[e0 + i] = run;
      run += (e0 + i < e1) ? cv[i] : 0;
    }
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < BQC_CPT; ++c) {
    const int j = tid + c * BQC_THREADS;
    if (j < m) {
      const int pos = atomicAdd(&s_end[hs[c]], 1);   // leaves s_end[h] = END of bucket h = start of bucket h + 1
      s_ctr[pos] = make_float4(cx[c], cy[c], cz[c], __builtin_bit_cast(float, j));
    }
  }
  __syncthreads();
}

__global__ __launch_bounds__(BQC_THREADS) void bqc_scatter_kernel(BqcParams P, int *__restrict__ cnt,
                                                                  int *__restrict__ list) {
  extern __shared__ __attribute__((aligned(16))) int bqc_smem[];
  __shared__ int s_wave[BQC_THREADS / 64];
  __shared__ int s_nhit;
  int li = 0;
  for (int i = 1; i < P.nlevels; ++i)
    if ((int)blockIdx.x >= P.lv[i].wg_begin) li = i;
  const BqcLevel &L = P.lv[li];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int local = (int)blockIdx.x - L.wg_begin;
  const int bi = local / L.chunks, chunk = local % L.chunks;
  const int n = L.n, m = L.m, H = L.hsize;
  const float radius2 = L.radius2;
  if (L.kind == 1) {
    // Small scenes (n <= BQC_SCAN_MAXN: the deeper levels, where a ball holds a tenth of the scene and cells
    // prune nothing): the ordered 64-point scan, but from an LDS copy of the scene -- the n / 64 dependent steps
    // of a wave wait for LDS instead of L2 (13 -> ~4 us at the SA2 shape) -- inside the same launch.
    float *s_pts = reinterpret_cast<float *>(bqc_smem);
    const float *xyz = L.xyz + (size_t)bi * n * 3;
    for (int i = tid; i < 3 * n; i += BQC_THREADS) s_pts[i] = xyz[i];
    __syncthreads();
    const int c0 = (chunk * (BQC_THREADS / 64) + wave) * L.cpw;
    if (c0 >= m) return;
    const float *cs = L.new_xyz + ((size_t)bi * L.m_total + L.c_off) * 3;
    int *rows = L.idx + ((size_t)bi * L.m_total + L.c_off) * L.nsample;
    int tests = 0;
    if (L.cpw == 8) bq_scan_wave<8>(n, m, radius2, L.nsample, cs, s_pts, rows, c0, lane, &tests);
    else if (L.cpw == 4) bq_scan_wave<4>(n, m, radius2, L.nsample, cs, s_pts, rows, c0, lane, &tests);
    else if (L.cpw == 2) bq_scan_wave<2>(n, m, radius2, L.nsample, cs, s_pts, rows, c0, lane, &tests);
    else bq_scan_wave<1>(n, m, radius2, L.nsample, cs, s_pts, rows, c0, lane, &tests);
    if (P.stats && lane == 0) atomicAdd(P.stats, (unsigned long long)tests * 64ull);
    return;
  }
  const unsigned hmask = (unsigned)H - 1u;
  const float inv_e = L.inv_e, reach = L.reach;
  // this thread's points (BQC_PPT of the workgroup's BQC_PPT * 512), requested before the table is built
  float px[BQC_PPT], py[BQC_PPT], pz[BQC_PPT];
  int pk[BQC_PPT];
  {
    const float *xyz = L.xyz + (size_t)bi * n * 3;
#pragma unroll
    for (int t = 0; t < BQC_PPT; ++t) {
      pk[t] = (chunk * BQC_PPT + t) * BQC_THREADS + tid;
      const int kk = min(pk[t], n - 1);
      px[t] = xyz[3 * kk + 0];
      py[t] = xyz[3 * kk + 1];
      pz[t] = xyz[3 * kk + 2];
    }
  }
  int *s_end = bqc_smem;                                         // [H] counts -> starts -> ends
  float4 *s_ctr = reinterpret_cast<float4 *>(bqc_smem + H);     // [m] {x, y, z, centre}
  const float *ctr = L.new_xyz + ((size_t)bi * L.m_total + L.c_off) * 3;

  bqc_build_table(s_end, s_ctr, s_wave, ctr, m, H, inv_e, hmask, tid, lane, wave);

  // The chunk's points against the table, TWICE.  A returning global atomic per hit inside these divergent
  // loops made every loop trip a memory round trip (measured: 105 us at SA3, where 1024 points find 11 centres
  // each).  Pass A only counts a centre's hits in LDS; then ONE global atomic per centre reserves the chunk's
  // range of the centre's list (issued by m threads at once: one round trip); pass B repeats the tests and
  // places every hit with a returning LDS atomic.  The distance tests are a few VALU instructions per candidate.
  BQ_MARK(4);
  int *cnt_l = cnt + L.ctr_begin + (size_t)bi * m;
  int *list_l = list + ((size_t)L.ctr_begin + (size_t)bi * m) * BQC_CAP;
  int *s_hits = reinterpret_cast<int *>(s_ctr + m);   // [m] hits of this chunk per centre, then running position
  int *s_base = s_hits + m;                            // [m] first list slot of this chunk's hits
  for (int i = tid; i < m; i += BQC_THREADS) s_hits[i] = 0;
  // the (at most eight) buckets of a point's 2 x 2 x 2 cell block
  int p0[BQC_PPT][8], p1[BQC_PPT][8];
#pragma unroll
  for (int t = 0; t < BQC_PPT; ++t) {
    const float x = px[t], y = py[t], z = pz[t];
    const int lx = bqc_cell(x - reach, inv_e), ly = bqc_cell(y - reach, inv_e), lz = bqc_cell(z - reach, inv_e);
    const bool tx = bqc_cell(x + reach, inv_e) > lx, ty = bqc_cell(y + reach, inv_e) > ly,
               tz = bqc_cell(z + reach, inv_e) > lz;
    unsigned hh[8];
    bool ok[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      ok[q] = pk[t] < n && ((q & 1) == 0 || tx) && ((q & 2) == 0 || ty) && ((q & 4) == 0 || tz);
      hh[q] = bqc_hash(lx + (q & 1), ly + ((q >> 1) & 1), lz + (q >> 2), hmask);
#pragma unroll
      for (int p = 0; p < q; ++p) ok[q] = ok[q] && !(ok[p] && hh[p] == hh[q]);   // a bucket is walked once
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {   // unconditional reads (a read under `ok ?` is a branch and a wait each), masked after
      const int e1v = s_end[hh[q]], e0v = s_end[hh[q] ? hh[q] - 1 : 0];
      p1[t][q] = ok[q] ? e1v : 0;
      p0[t][q] = ok[q] ? (hh[q] ? e0v : 0) : 0;
    }
  }
  __syncthreads();
  BQ_MARK(5);
  // The points against their candidates.  A returning global atomic per hit inside these divergent loops made every
  // loop trip a memory round trip (measured: 105 us at the SA3 shape).  So the pass only counts a centre's hits in
  // LDS and notes every hit (centre, point) in a workgroup list; then ONE global atomic per centre reserves this
  // workgroup's range of the centre's list (issued by m threads at once: one round trip) and the noted hits are
  // placed by a dense loop (a returning LDS atomic each).  A workgroup with more than BQC_HITS hits (dense
  // clusters) repeats the tests instead of reading the list.
  int *s_list = s_base + m;                // [BQC_HITS] centre << 10 | point of the chunk
  if (tid == 0) s_nhit = 0;
  __syncthreads();
  int ntests = 0;
#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {
    if (pass == 1 && s_nhit <= BQC_HITS) {   // uniform: every hit is in the list
      const int nh = s_nhit;
      for (int hq = tid; hq < nh; hq += BQC_THREADS) {
        const int ent = s_list[hq], j = ent >> 10, pl = ent & 1023;
        const int slot = s_base[j] + atomicAdd(&s_hits[j], 1);
        if (slot < BQC_CAP) list_l[(size_t)j * BQC_CAP + slot] = chunk * (BQC_PPT * BQC_THREADS) + pl;
      }
      break;
    }
    // (Round 4 tried requesting the first two records of four buckets at a time, eight LDS reads in flight, with only
    // the 1.4 % of buckets that hold more than two centres left in a dependent loop: 42 us instead of 39 for the four
    // levels at 128 registers, 61 us at the 171 the compiler wants -- with four waves per SIMD the walk's LDS round
    // trips were hidden already, and the second workgroup per CU is worth more than the reads in flight.)
#pragma unroll
    for (int t = 0; t < BQC_PPT; ++t) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        for (int p = p0[t][q]; p < p1[t][q]; ++p) {
          const float4 C = s_ctr[p];
          ++ntests;
          if (sq_dist3(C.x, C.y, C.z, px[t], py[t], pz[t]) < radius2) {   // ball_query_gpu.cu:31-33
            const int j = __builtin_bit_cast(int, C.w);
            if (pass == 0) {
              atomicAdd(&s_hits[j], 1);
              // one list-cursor atomic per wave and loop trip (the lanes that hit right now), not one per hit
              const unsigned long long act = __ballot(1);
              const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(act >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)act, 0));
              int base = 0;
              if (rank == 0) base = atomicAdd(&s_nhit, __builtin_popcountll(act));
              const int hq = __builtin_amdgcn_readfirstlane(base) + rank;
              if (hq < BQC_HITS) s_list[hq] = (j << 10) | (t * BQC_THREADS + tid);
            } else {
              const int slot = s_base[j] + atomicAdd(&s_hits[j], 1);
              if (slot < BQC_CAP) list_l[(size_t)j * BQC_CAP + slot] = pk[t];
            }
          }
        }
      }
    }
    __syncthreads();
    BQ_MARK(6 + 2 * pass);
    if (pass == 0) {
      for (int j = tid; j < m; j += BQC_THREADS) {
        const int h = s_hits[j];
        s_base[j] = h ? atomicAdd(&cnt_l[j], h) : 0;
        s_hits[j] = 0;
      }
      __syncthreads();
      BQ_MARK(7);
    }
  }
  BQ_MARK(8);
  if (P.stats) {
    const int wsum = (int)wave_allreduce_sum_f32((float)ntests);   // < 2^24 per wave: exact
    if (lane == 0) atomicAdd(P.stats, (unsigned long long)wsum);
  }
}

__global__ __launch_bounds__(256) void bqc_rank_kernel(BqcParams P, int *__restrict__ cnt,
                                                       const int *__restrict__ list) {
  __shared__ int s_list[BQC_RANK_CENTRES][BQC_CAP];
  int li = 0;
  for (int i = 1; i < P.nlevels; ++i)
    if ((int)blockIdx.x >= P.lv[i].rank_wg_begin) li = i;
  const BqcLevel &L = P.lv[li];
  const int lane = lane_id(), l16 = lane & 15, grp = (int)(threadIdx.x >> 4);
  const int m = L.m, nsample = L.nsample;
  const int slot = ((int)blockIdx.x - L.rank_wg_begin) * BQC_RANK_CENTRES + grp;   // bi * m + j
  const bool valid = slot < P.b * m;
  const int bi = valid ? slot / m : 0, j = valid ? slot % m : 0;
  const int c = valid ? cnt[L.ctr_begin + slot] : 0;
  if (valid && l16 == 0) cnt[L.ctr_begin + slot] = 0;   // the counters leave as they must arrive: zero (SIG3D_BQ_CLEAN)
  int *row = L.idx + ((size_t)bi * L.m_total + L.c_off + j) * nsample;
  const int *lst = list + ((size_t)L.ctr_begin + slot) * BQC_CAP;
  int *s = s_list[grp];
  const bool ranked = valid && c > 0 && c <= BQC_CAP;
  if (ranked)
    for (int i = l16; i < c; i += 16) s[i] = lst[i];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  if (valid && c == 0)   // ball_query.cpp:19-21: the row stays zero
    for (int l = l16; l < nsample; l += 16) row[l] = 0;
  unsigned lo = 0x7FFFFFFFu;
  if (ranked) {
    // rank of a hit = number of hits with a smaller index (indices are unique): index order restored
    for (int i = l16; i < c; i += 16) {
      const int v = s[i];
      int rank = 0;
      for (int q = 0; q < c; ++q) rank += s[q] < v;
      if (rank < nsample) row[rank] = v;
      lo = min(lo, (unsigned)v);
    }
  }
  lo = row_allreduce_min_u32(lo);   // 16-lane DPP row == this centre's group: first hit in index order
  if (ranked)
    for (int l = c + l16; l < nsample; l += 16) row[l] = (int)lo;   // ball_query_gpu.cu:34-38
  // dense neighbourhoods: ordered scan of the whole scene by the full wave, one overflowing centre at a time
  // (same code path as the brute-force kernel with one centre)
  const unsigned long long over = __ballot(valid && c > BQC_CAP);
  if (over == 0ull) return;
  const float *ctr = L.new_xyz + ((size_t)bi * L.m_total + L.c_off + j) * 3;
  const float mx = valid ? ctr[0] : 0.f, my = valid ? ctr[1] : 0.f, mz = valid ? ctr[2] : 0.f;
  const int n = L.n;
  const float radius2 = L.radius2;
  for (int g = 0; g < 4; ++g) {
    if (((over >> (16 * g)) & 1ull) == 0ull) continue;
    const int src = 16 * g;
    const float cx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mx), src));
    const float cy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my), src));
    const float cz = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mz), src));
    const int gbi = __builtin_amdgcn_readlane(bi, src), gj = __builtin_amdgcn_readlane(j, src);
    const float *xyz = L.xyz + (size_t)gbi * n * 3;
    int *grow = L.idx + ((size_t)gbi * L.m_total + L.c_off + gj) * nsample;
    int have = 0, first = 0;
    for (int k0 = 0; k0 < n && have < nsample; k0 += 64) {
      const int k = k0 + lane;
      const bool inb = k < n;
      const int kk = inb ? k : n - 1;
      const bool hit = inb && (sq_dist3(cx, cy, cz, xyz[3 * kk + 0], xyz[3 * kk + 1], xyz[3 * kk + 2]) < radius2);
      const unsigned long long mask = __ballot(hit);
      if (mask != 0ull) {
        const int at = have + __builtin_popcountll(mask & ((1ull << lane) - 1ull));
        if (hit && at < nsample) grow[at] = k;
        if (have == 0) first = k0 + __builtin_ctzll(mask);
        have += __builtin_popcountll(mask);
      }
    }
    for (int l = min(have, nsample) + lane; l < nsample; l += 64) grow[l] = first;
  }
}

// Host side of a multi-level launch: fills P (splitting problems with more than BQC_MAXM centres into blocks)
// and returns the workspace it needs, or -1 when the problems do not fit one launch.
static long bqc_plan(int b, int nlevels, const sig3d_bq_level *levels, BqcParams *P, int *scatter_wgs,
                     int *rank_wgs, size_t *lds_bytes) {
  int nl = 0, wg = 0, rwg = 0;
  long slots = 0;
  size_t lds = 0;
  for (int i = 0; i < nlevels; ++i) {
    const sig3d_bq_level &q = levels[i];
    if (q.m <= 0 || q.nsample <= 0) continue;
    for (int c0 = 0; c0 < q.m; c0 += BQC_MAXM) {
      if (nl == BQC_MAXLV) return -1;
      BqcLevel &L = P->lv[nl++];
      L.xyz = q.xyz; L.new_xyz = q.new_xyz; L.idx = q.idx;
      L.n = q.n; L.m = q.m - c0 < BQC_MAXM ? q.m - c0 : BQC_MAXM; L.m_total = q.m; L.c_off = c0;
      L.nsample = q.nsample;
      int h = 64;
      while (h < 2 * L.m) h <<= 1;
      L.hsize = h;
      L.radius2 = q.radius * q.radius;        // ball_query_gpu.cu:22, f32 product on the host
      L.inv_e = 1.f / (q.radius * 2.02f);
      L.reach = q.radius * 1.005f;
      L.kind = q.n <= BQC_SCAN_MAXN ? 1 : 0;
      if (L.kind == 1) {   // centres per wave as in sig3d_ball_query: fewer until there are ~4096 waves
        const long centres = (long)b * L.m;
        L.cpw = centres >= 8L * 4096 ? 8 : centres >= 4L * 4096 ? 4 : centres >= 2L * 4096 ? 2 : 1;
        L.chunks = q.n > 0 ? sig3d_ceil_div(L.m, (BQC_THREADS / 64) * L.cpw) : 0;
      } else {
        L.cpw = 0;
        L.chunks = sig3d_ceil_div(q.n, BQC_THREADS * BQC_PPT);
      }
      L.wg_begin = wg; L.rank_wg_begin = rwg; L.ctr_begin = (int)slots;
      wg += b * L.chunks;
      // kind 1 writes its rows itself -- except for an empty scene (no workgroup at all): the rank kernel then
      // writes the zero rows from the zeroed counters
      const bool ranked = L.kind == 0 || L.chunks == 0;
      rwg += ranked ? sig3d_ceil_div((long)b * L.m, BQC_RANK_CENTRES) : 0;
      slots += ranked ? (long)b * L.m : 0;
      const size_t need = L.kind == 1 ? sizeof(float) * 3 * (size_t)q.n
                                      : sizeof(int) * ((size_t)h + 2 * (size_t)L.m + BQC_HITS) + sizeof(float4) * (size_t)L.m;
      if (need > lds) lds = need;
    }
  }
  P->nlevels = nl;
  P->b = b;
  if (scatter_wgs) *scatter_wgs = wg;
  if (rank_wgs) *rank_wgs = rwg;
  if (lds_bytes) *lds_bytes = lds;
  if (slots >= (1L << 22)) return -1;   // slot * BQC_CAP stays inside 31 bits
  return slots * (long)sizeof(int) * (1 + BQC_CAP);
}

}  // namespace

#ifdef SIG3D_BQ_TIMING
extern "C" int sig3d_debug_bq_marks(unsigned long long *host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_bq_marks), sizeof(unsigned long long) * 16);
}
#endif

extern "C" int sig3d_ball_query(int b, int n, int m, float radius, int nsample,
                                const float *new_xyz, const float *xyz, int *idx,
                                void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SIG3D_REQUIRE(b >= 0 && n >= 0 && m >= 0 && nsample >= 0, "negative size");
  if (b == 0 || m == 0 || nsample == 0) return 0;
  if (n == 0) {  // no points: all rows stay zero
    SIG3D_HIP_TRY(hipMemsetAsync(idx, 0, sizeof(int) * (size_t)b * m * nsample, stream));
    return 0;
  }
  const float radius2 = radius * radius;  // ball_query_gpu.cu:22, f32 product on the host
  // Centres per wave: 8 amortise the point loads best, but the scan is a chain of n / 64 dependent steps per
  // wave, and the small levels (SA2-4: 8192 / 4096 / 2048 centres in all) then leave most of the chip idle
  // behind 1024 / 512 / 256 long-running waves (20 us each for 2 MB of work).  Fewer centres per wave until
  // there are ~4096 waves: the scene (24 KB at SA2) is cache-resident anyway.
  const long centres = (long)b * m;
  const int cpw = centres >= 8L * 4096 ? 8 : centres >= 4L * 4096 ? 4 : centres >= 2L * 4096 ? 2 : 1;
  dim3 grid(sig3d_ceil_div(m, BQ_WAVES * cpw), b);
#define SIG3D_BQ_SCAN(CPW)                                                                                  \
  hipLaunchKernelGGL((ball_query_scan_kernel<CPW>), grid, dim3(BQ_WAVES * 64), 0, stream, n, m, radius2,    \
                     nsample, new_xyz, xyz, idx)
  if (cpw == 8) SIG3D_BQ_SCAN(8);
  else if (cpw == 4) SIG3D_BQ_SCAN(4);
  else if (cpw == 2) SIG3D_BQ_SCAN(2);
  else SIG3D_BQ_SCAN(1);
#undef SIG3D_BQ_SCAN
  SIG3D_LAUNCH_CHECK("ball_query_scan_kernel");
  return 0;
}

extern "C" long sig3d_ball_query_levels_workspace_bytes(int b, int nlevels, const sig3d_bq_level *levels) {
  if (b <= 0 || nlevels <= 0 || levels == nullptr) return 0;
  BqcParams P;
  P.stats = nullptr;
  return bqc_plan(b, nlevels, levels, &P, nullptr, nullptr, nullptr);     // counters + lists
}

extern "C" int sig3d_ball_query_levels(int b, int nlevels, const sig3d_bq_level *levels, void *workspace,
                                       long workspace_bytes, void *stream_) {
  return sig3d_ball_query_levels_ex(b, nlevels, levels, workspace, workspace_bytes, 0, stream_);
}

extern "C" int sig3d_ball_query_levels_ex(int b, int nlevels, const sig3d_bq_level *levels, void *workspace,
                                          long workspace_bytes, int flags, void *stream_) {
  return sig3d_ball_query_levels_stats(b, nlevels, levels, workspace, workspace_bytes, flags, nullptr, stream_);
}

extern "C" int sig3d_ball_query_levels_stats(int b, int nlevels, const sig3d_bq_level *levels, void *workspace,
                                             long workspace_bytes, int flags, unsigned long long *stats,
                                             void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SIG3D_REQUIRE(b >= 0 && nlevels >= 0 && (nlevels == 0 || levels != nullptr), "bad arguments");
  if (b == 0 || nlevels == 0) return 0;
  for (int i = 0; i < nlevels; ++i) {
    SIG3D_REQUIRE(levels[i].n >= 0 && levels[i].m >= 0 && levels[i].nsample >= 0, "negative size");
    SIG3D_REQUIRE(levels[i].radius > 0.f, "sig3d_ball_query_levels needs radius > 0 (use sig3d_ball_query)");
    SIG3D_REQUIRE((long)b * levels[i].m * levels[i].nsample < (1L << 31) && (long)b * levels[i].n * 3 < (1L << 31),
                  "problem too large for 32-bit indexing");
  }
  BqcParams P;
  int wgs = 0, rwgs = 0;
  size_t lds = 0;
  const long need = bqc_plan(b, nlevels, levels, &P, &wgs, &rwgs, &lds);
  SIG3D_REQUIRE(need >= 0, "too many problems / centres for one launch (16 blocks of 4096 centres)");
  P.stats = stats;
  if (P.nlevels == 0) return 0;
  SIG3D_REQUIRE(need == 0 || (workspace != nullptr && workspace_bytes >= need),
                "workspace too small: see sig3d_ball_query_levels_workspace_bytes");
  const long slots = need / (long)(sizeof(int) * (1 + BQC_CAP));
  int *cnt = (int *)workspace;
  int *list = cnt + slots;
  // the rank kernel zeroes every counter it reads, so a workspace that has been through one call (same problem
  // list) is clean: SIG3D_BQ_CLEAN lets the caller vouch for that and saves the memset node
  if (slots > 0 && !(flags & SIG3D_BQ_CLEAN)) SIG3D_HIP_TRY(hipMemsetAsync(cnt, 0, sizeof(int) * (size_t)slots, stream));
  if (wgs > 0) {
    if (lds > 48 * 1024)
      SIG3D_HIP_TRY(hipFuncSetAttribute((const void *)bqc_scatter_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)lds));
    // (Round 6 tried the cell-role level and the ordered-scan levels as two launches with their own LDS sizes -- every
    // workgroup of one launch is given the largest level's 74 KB, so a scan workgroup shares its CU with one other
    // instead of three: 50.6 us for the stack against 39.3 in one launch.  The scan workgroups fill the cell-role
    // workgroups' tail; a launch boundary between them costs more than the occupancy returns.)
    hipLaunchKernelGGL(bqc_scatter_kernel, dim3(wgs), dim3(BQC_THREADS), lds, stream, P, cnt, list);
  }
  if (rwgs > 0) hipLaunchKernelGGL(bqc_rank_kernel, dim3(rwgs), dim3(256), 0, stream, P, cnt, list);
  SIG3D_LAUNCH_CHECK("ball query (cell-binned centres)");
  return 0;
}

extern "C" int sig3d_ball_query_grid(int b, int n, int m, float radius, int nsample,
                                     const float *new_xyz, const float *xyz, int *idx, void *workspace,
                                     long workspace_bytes, void *stream_) {
  SIG3D_REQUIRE(b >= 0 && n >= 0 && m >= 0 && nsample >= 0, "negative size");
  if (b == 0 || m == 0 || nsample == 0) return 0;
  if (n < 256 || !(radius > 0.f))  // tiny scenes / degenerate radius: the ordered scan is the right tool
    return sig3d_ball_query(b, n, m, radius, nsample, new_xyz, xyz, idx, stream_);
  sig3d_bq_level q;
  q.n = n; q.m = m; q.nsample = nsample; q.radius = radius; q.xyz = xyz; q.new_xyz = new_xyz; q.idx = idx;
  if (sig3d_ball_query_levels_workspace_bytes(b, 1, &q) < 0)   // more than 16 x 4096 centres
    return sig3d_ball_query(b, n, m, radius, nsample, new_xyz, xyz, idx, stream_);
  return sig3d_ball_query_levels(b, 1, &q, workspace, workspace_bytes, stream_);
}
