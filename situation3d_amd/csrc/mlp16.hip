// mlp16.hip -- a SharedMLP layer (1x1 Conv2d as a GEMM over positions) on v_mfma_f32_16x16x4_f32 with ds_read_b128
// operand fragments: the core the weight gradients got in round 4 (gemm16_core.h), for the forward and input-gradient
// products of lib/pointnet2/pytorch_utils.py:11-36.
//
//   y[b][co][e] = sum_k W[co][k] * act(x[b][k][e])        act = identity, or relu(x * pscale[k] + pshift[k])
//
// What the round-1 kernel (mlp_layer_fwd_kernel, 32x32x2 MFMA, one ds_read_b32 per MFMA for the weight operand, the
// activation operand straight from global memory into the MFMA) pays per MFMA in LDS reads, this one pays per sixteen:
//   * a workgroup of 8 waves owns TM = 64 or 128 output channels x 256 positions; a wave all TM channels x 32 positions
//     (AB x 2 blocks of 16 x 16: 10 operand fragments of 16 bytes feed 64 MFMAs at AB = 8);
//   * the weight tile (TM x K, K <= 288) is staged ONCE per workgroup as [K / 32][TM][32] with the 16-byte-slot XOR swizzle
//     of gemm16_core.h (conflict-free ds_read_b128 by 16 rows x 4 slots); workgroups are persistent over position tiles;
//   * the activation tile streams through two LDS stages of [256 positions][32 k]: channel-major activations
//     (position-contiguous rows) are read as 16-byte runs of four positions and turned by their 4-byte LDS stores, the
//     previous layer's BatchNorm + ReLU applied on the way (one scale / shift pair per thread and chunk); a GATHERING
//     first layer reads 16-byte runs of a neighbour's point-major feature row (k-contiguous: no turn) through the
//     ball-query list and forms the xyz offset chunk (xyz[idx] - centre, / radius: QueryAndGroup's arithmetic,
//     pointnet2_utils.py:348-359) in registers;
//   * the epilogue stores 64-byte runs per channel row and keeps the per-channel sum / sum of squares of y (weighted by
//     the multiplicity of a compact position) in registers across all tiles of the workgroup: one DPP row reduction, one
//     LDS add and one f64 atomic per channel and workgroup at the end;
//   * input-gradient form (w_t): the stored forward weight is staged transposed, once.
// Sums are f32 in another order than the round-1 kernel's (far inside the 1e-4 bar; same tests).
#include "mlp16.h"

#include "sig3d_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int M16_TN = 256;      // positions per workgroup tile
constexpr int M16_THREADS = 512;

// float offset of (row, 16-byte slot q) inside a [rows][32] tile (gemm16_core.h: slot_off)
__device__ __forceinline__ int m16_slot(int row, int q) { return row * 32 + ((q ^ ((row >> 1) & 7)) << 2); }

template <int AB, bool PRO, bool GATHER>
__global__ __launch_bounds__(M16_THREADS) void mlp16_kernel(const Mlp16Args A, const int kpad, const int tiles_per_wg) {
  constexpr int TM = 16 * AB;
  extern __shared__ __attribute__((aligned(16))) float m16_smem[];
  const int nck = kpad / 32;
  float *s_w = m16_smem;                       // [nck][TM][32]
  float *s_b = s_w + (size_t)nck * TM * 32;    // [2][M16_TN][32]
  float *s_ps = s_b + 2 * M16_TN * 32;         // [kpad] scale, [kpad] shift
  float *s_pb = s_ps + kpad;
  float *s_stat = s_pb + kpad;                 // [8 waves][2][TM]: per-wave channel sums, folded in a fixed order

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  const int m0 = blockIdx.x * TM, bi = blockIdx.z;
  const long E = A.E;
  const long En = A.n_act ? (long)A.n_act[bi] : E;
  const int n_tiles = (int)((En + M16_TN - 1) / M16_TN);
  if ((int)blockIdx.y >= n_tiles) return;      // (uniform: no barrier has been reached)
  const int cin = A.cin, cout = A.cout;

  // ---- weights -> LDS, once.  LDS column kc of chunk c stands for reduction index k = 32 c + kc.
  if (A.w_t) {
    // w is (cin, cout) row-major: W^T[m][k] = w[k * cout + m]; lanes along m (coalesced), 4-byte LDS stores
    for (int k = wv; k < kpad; k += M16_THREADS / 64) {
      for (int m = lane; m < TM; m += 64) {
        const float v = (k < cin && m0 + m < cout) ? A.w[(size_t)k * cout + m0 + m] : 0.f;
        s_w[(k >> 5) * TM * 32 + m16_slot(m, (k & 31) >> 2) + (k & 3)] = v;
      }
    }
  } else if (GATHER) {
    // reduction slot k < C: feature channel k = column 3 + k of the layer's weight; C <= k < C + 3: xyz channel k - C
    const int C = A.gC;
    for (int i = tid; i < TM * kpad; i += M16_THREADS) {
      const int m = i / kpad, k = i - m * kpad;
      const int col = k < C ? k + 3 : k - C;
      const float v = (k < C + 3 && m0 + m < cout) ? A.w[(size_t)(m0 + m) * cin + col] : 0.f;
      s_w[(k >> 5) * TM * 32 + m16_slot(m, (k & 31) >> 2) + (k & 3)] = v;
    }
  } else {
    // (cout, cin) rows, k contiguous, cin % 4 == 0: 16-byte loads and stores
    const int per_row = kpad / 4;
    for (int i = tid; i < TM * per_row; i += M16_THREADS) {
      const int m = i / per_row, q4 = i - m * per_row, k = 4 * q4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (m0 + m < cout) v = *reinterpret_cast<const f32x4 *>(A.w + (size_t)(m0 + m) * cin + k);
      *reinterpret_cast<f32x4 *>(s_w + (k >> 5) * TM * 32 + m16_slot(m, (k & 31) >> 2)) = v;
    }
  }
  if (PRO)
    for (int i = tid; i < kpad; i += M16_THREADS) {
      s_ps[i] = i < cin ? A.pscale[i] : 0.f;
      s_pb[i] = i < cin ? A.pshift[i] : 0.f;
    }
  __syncthreads();

  // ---- this thread's units of an activation chunk
  // channel-major: wave-instruction wi = wv + 8 i covers 16 k x 16 positions: k = 16 (wi & 1) + lane / 4,
  // positions 16 (wi / 2) + 4 (lane % 4) .. + 3 (one 16-byte load; four 4-byte LDS stores into [position][k])
  // gathering: unit u = tid + 512 i is (position u / 8, slot u % 8) of the chunk (one 16-byte load and store)
  // (unit i + 1 sits 64 positions behind unit i in both forms: row + 64 leaves the swizzle term (row >> 1) & 7 alone)
  const int u_k = GATHER ? 4 * (tid & 7) : 16 * (wv & 1) + (lane >> 2);   // k inside the chunk of the unit's first element
  const int u_pos0 = GATHER ? (tid >> 3) : 16 * (wv >> 1) + 4 * (lane & 3);
  const int u_lds0 = GATHER ? m16_slot(tid >> 3, tid & 7) : m16_slot(u_pos0, u_k >> 2) + (u_k & 3);      // rows n, n + 1
  const int u_lds1 = GATHER ? 0 : m16_slot(u_pos0 + 2, u_k >> 2) + (u_k & 3);   // rows n + 2, n + 3 (the swizzle term moves on by one)
  const float *xb = A.x + (size_t)bi * cin * E;
  float *yb = A.y + (size_t)bi * cout * E;
  const float *mb = A.mult ? A.mult + (size_t)bi * E : nullptr;
  const bool stats = A.stat_sum != nullptr;

  float s1[AB][4], s2[AB][4];
#pragma unroll
  for (int a = 0; a < AB; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) { s1[a][r] = 0.f; s2[a][r] = 0.f; }

  // fragment read offsets of this lane (gemm16_core.h): row lr of a block, slots lg (half 0) and lg + 4 (half 1)
  const int swz[2] = {((lg) ^ (lr >> 1)) << 2, ((lg + 4) ^ (lr >> 1)) << 2};
  const int offA = lr * 32, offB = (wv * 32 + lr) * 32;

  for (int t = 0; t < tiles_per_wg; ++t) {
    const int tile = (int)blockIdx.y + t * (int)gridDim.y;
    if (tile >= n_tiles) break;                                  // (uniform)
    const long e0 = (long)tile * M16_TN;

    // gathering: the neighbour of each of my four positions, and the xyz offset chunk of the slot-0 units
    int gi[4];
    f32x4 gx[4];
    if (GATHER) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const long pos = min(e0 + (u_pos0 + 64 * i), En - 1);
        gi[i] = A.g_idx[(size_t)bi * E + pos];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        gx[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if ((tid & 7) == 0) {
          const long pos = min(e0 + (u_pos0 + 64 * i), En - 1);
          const int ctr = A.g_centre_of ? A.g_centre_of[(size_t)bi * E + pos] : (int)(pos / A.gS);
          const float *pt = A.g_xyz + ((size_t)bi * A.gN + gi[i]) * 3;
          const float *cc = A.g_centre + ((size_t)bi * A.gP + ctr) * 3;
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            float d = __fsub_rn(pt[j], cc[j]);
            if (A.g_normalize) d = __fdiv_rn(d, A.g_radius);
            gx[i][j] = d;
          }
        }
      }
    }
    auto load_chunk = [&](int c, f32x4 (&r)[4]) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (GATHER) {
          if (c * 32 < A.gC)
            r[i] = *reinterpret_cast<const f32x4 *>(A.g_feat + ((size_t)bi * A.gN + gi[i]) * A.gC + c * 32 + u_k);
          else
            r[i] = gx[i];
        } else {
          const long pos = min(e0 + (u_pos0 + 64 * i), E - 4);
          r[i] = *reinterpret_cast<const f32x4 *>(xb + (size_t)(c * 32 + u_k) * E + pos);
        }
      }
    };
    auto store_chunk = [&](float *sb, int c, const f32x4 (&r)[4]) {
      float sc = 1.f, sh = 0.f;
      if (PRO) { sc = s_ps[c * 32 + u_k]; sh = s_pb[c * 32 + u_k]; }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f32x4 v = r[i];
        if (PRO) {
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = fmaxf(0.f, fmaf(v[q], sc, sh));
        }
        if (GATHER) {
          *reinterpret_cast<f32x4 *>(sb + (u_lds0 + 2048 * i)) = v;
        } else {
          sb[(u_lds0 + 2048 * i)] = v[0];
          sb[(u_lds0 + 2048 * i) + 32] = v[1];
          sb[(u_lds1 + 2048 * i)] = v[2];
          sb[(u_lds1 + 2048 * i) + 32] = v[3];
        }
      }
    };

    f32x4 acc[AB][2];
#pragma unroll
    for (int a = 0; a < AB; ++a) { acc[a][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[a][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    f32x4 ring[4];
    load_chunk(0, ring);
    store_chunk(s_b, 0, ring);
    __syncthreads();
    for (int c = 0; c < nck; ++c) {
      const bool more = c + 1 < nck;
      if (more) load_chunk(c + 1, ring);
      const float *sa = s_w + (size_t)c * TM * 32 + offA;
      const float *sb = s_b + (c & 1) * (M16_TN * 32) + offB;
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        f32x4 fb[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) fb[b] = *reinterpret_cast<const f32x4 *>(sb + b * 16 * 32 + swz[hh]);
#pragma unroll
        for (int a4 = 0; a4 < AB; a4 += 4) {       // four row blocks at a time: 16 fragment registers live, not 32
          f32x4 fa[4];
#pragma unroll
          for (int a = 0; a < 4; ++a) fa[a] = *reinterpret_cast<const f32x4 *>(sa + (a4 + a) * 16 * 32 + swz[hh]);
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
              for (int b = 0; b < 2; ++b)
                acc[a4 + a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[a][j], fb[b][j], acc[a4 + a][b], 0, 0, 0);
        }
      }
      if (more) store_chunk(s_b + ((c + 1) & 1) * (M16_TN * 32), c + 1, ring);
      __syncthreads();
    }

    // ---- epilogue.  C/D map of the 16x16 MFMA: column (position) = lane & 15, row (channel) = 4 (lane >> 4) + register
    long pos[2];
    bool ok[2];
    float mw[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      pos[b] = e0 + wv * 32 + b * 16 + lr;
      ok[b] = pos[b] < En;
      mw[b] = ok[b] ? (mb ? mb[pos[b]] : 1.f) : 0.f;
    }
#pragma unroll
    for (int a = 0; a < AB; ++a) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = m0 + a * 16 + 4 * lg + r;
        if (co < cout) {
          float *yrow = yb + (size_t)co * E;
#pragma unroll
          for (int b = 0; b < 2; ++b)
            if (ok[b]) yrow[pos[b]] = acc[a][b][r];
        }
        if (stats) {
          const float v0 = acc[a][0][r], v1 = acc[a][1][r];
          s1[a][r] += mw[0] * v0 + mw[1] * v1;
          s2[a][r] += mw[0] * v0 * v0 + mw[1] * v1 * v1;
        }
      }
    }
  }

  if (!stats) return;
  // per-channel sums: the 16 lanes of a DPP row hold 16 positions of the same four channels; every (wave, channel) slot
  // has one writer, and the waves' slots are folded in a fixed order in f64 (no float atomics: the statistics of a
  // workgroup do not depend on the order its waves arrive in)
#pragma unroll
  for (int a = 0; a < AB; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float t1 = row_allreduce_sum_f32(s1[a][r]), t2 = row_allreduce_sum_f32(s2[a][r]);
      if (lr == 0) {
        s_stat[(wv * 2 + 0) * TM + a * 16 + 4 * lg + r] = t1;
        s_stat[(wv * 2 + 1) * TM + a * 16 + 4 * lg + r] = t2;
      }
    }
  __syncthreads();
  for (int i = tid; i < 2 * TM; i += M16_THREADS) {
    const int which = i / TM, ch = i - which * TM, co = m0 + ch;
    if (co < cout) {
      double tot = 0.0;
#pragma unroll
      for (int w8 = 0; w8 < M16_THREADS / 64; ++w8) tot += (double)s_stat[(w8 * 2 + which) * TM + ch];
      unsafeAtomicAdd((which ? A.stat_sq : A.stat_sum) + co, tot);
    }
  }
}

size_t m16_lds_bytes(int tm, int kpad) {
  return sizeof(float) * ((size_t)kpad * tm + 2 * M16_TN * 32 + 2 * (size_t)kpad + (M16_THREADS / 64) * 2 * (size_t)tm);
}

int m16_kpad(const Mlp16Args &a) { return a.gather ? a.gC + 32 : (a.cin + 31) / 32 * 32; }

int m16_tile_rows(const Mlp16Args &a) {
  const int kpad = m16_kpad(a);
  if (a.cout % 128 == 0 && m16_lds_bytes(128, kpad) <= 160 * 1024) return 128;
  if (a.cout % 64 == 0 && m16_lds_bytes(64, kpad) <= 160 * 1024) return 64;
  return 0;
}

template <int AB, bool PRO, bool GATHER>
int m16_go(const Mlp16Args &a, hipStream_t stream) {
  constexpr int TM = 16 * AB;
  const int kpad = m16_kpad(a);
  const size_t lds = m16_lds_bytes(TM, kpad);
  static sig3d_once_per_device attr_done;   // per template instance
  if (attr_done.pending()) {
    SIG3D_HIP_TRY(hipFuncSetAttribute((const void *)mlp16_kernel<AB, PRO, GATHER>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      160 * 1024));
    attr_done.done();
  }
  // one workgroup per CU (LDS): a persistent grid of ~256 workgroups over (channel block, tile range, batch)
  const int cblocks = a.cout / TM;
  const long tiles = (a.E + M16_TN - 1) / M16_TN;          // per batch element (fewer with compact lists: they leave early)
  long gy = (256L + (long)a.b * cblocks - 1) / ((long)a.b * cblocks);
  if (gy > tiles) gy = tiles;
  if (gy < 1) gy = 1;
  const int tpw = (int)((tiles + gy - 1) / gy);
  hipLaunchKernelGGL((mlp16_kernel<AB, PRO, GATHER>), dim3(cblocks, (unsigned)gy, a.b), dim3(M16_THREADS), lds, stream, a, kpad, tpw);
  SIG3D_LAUNCH_CHECK("mlp16_kernel");
  return 0;
}

}  // namespace

bool sig3d_mlp16_applies(const Mlp16Args &a) {
  if (a.b <= 0 || a.E <= 0 || a.cout < 64 || a.cout % 64 != 0) return false;
  if (a.gather) {
    if (a.w_t || a.pscale != nullptr || a.gC < 32 || a.gC % 32 != 0 || a.gC > 256 || a.cin != a.gC + 3) return false;
    if (((uintptr_t)a.g_feat & 15) != 0) return false;
  } else {
    if (a.cin < 32 || a.cin % 32 != 0 || a.cin > 288 || a.E % 4 != 0 || ((uintptr_t)a.x & 15) != 0) return false;
    if (!a.w_t && ((uintptr_t)a.w & 15) != 0) return false;
  }
  return m16_tile_rows(a) != 0;
}

int sig3d_mlp16_launch(const Mlp16Args &a, hipStream_t stream) {
  const int tm = m16_tile_rows(a);
  SIG3D_REQUIRE(tm != 0, "mlp16: no tile for this shape (sig3d_mlp16_applies)");
  const bool pro = a.pscale != nullptr;
  if (a.gather) return tm == 128 ? m16_go<8, false, true>(a, stream) : m16_go<4, false, true>(a, stream);
  if (pro) return tm == 128 ? m16_go<8, true, false>(a, stream) : m16_go<4, true, false>(a, stream);
  return tm == 128 ? m16_go<8, false, false>(a, stream) : m16_go<4, false, false>(a, stream);
}
