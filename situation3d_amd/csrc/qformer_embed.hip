// qformer_embed.hip -- BertEmbeddings of the Q-Former (Qformer.py:70-98) and its additive masks as row kernels.
//
//   embeddings = word_embeddings(input_ids) + position_embeddings(position_ids)        (:84-89)
//   embeddings = cat(query_embeds, embeddings, dim=1)                                  (:91-92)
//   embeddings = dropout(LayerNorm(embeddings))                                        (:96-97)
//   extended mask = (1 - mask) * -10000                                                (:729-731)
//
// torch runs this as ~22 launches forward (two gathers, adds, fills, two cats, layer norm, dropout, four mask ops)
// and 13 backward for 416 rows of 768 floats -- 0.15 ms of a step whose whole Q-Former forward is 1.5 ms.  Here:
// ONE forward launch (a wave per output row: gathers its source row(s), normalises, drops, writes the row in the
// plain (B, Q+T, C) order or in the two-segment row layout of qformer.BertLayer.forward_segmented) and ONE backward
// launch (a workgroup per token index j, a wave per batch element: LayerNorm / dropout backward per row, then the
// sums over the batch that position j needs -- the shared query row or the position-table row -- meet in LDS; the
// word-table rows are scattered with float atomics into the zeroed dense gradient, or stored as rows for the
// data-parallel row exchange) + the fold of the per-workgroup gamma / beta partials (sig3d_column_sum).
#include "sig3d_common.h"

namespace {

constexpr int QE_MAX_PER_LANE = 16;  // hidden sizes up to 1024

__device__ __forceinline__ unsigned qe_mix32(unsigned x) {
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  return x;
}

struct QeShape {
  int b, q, t, cols, seg_rows;   // seg_rows > 0: two-segment layout, 2*seg_rows output rows
  int vocab, pos_rows, pos_off, pad_id;
};

// output row of token (bi, j), j < q: query token j, else text token j - q
__device__ __forceinline__ int qe_out_row(const QeShape &s, int bi, int j) {
  if (s.seg_rows > 0) return j < s.q ? bi * s.q + j : s.seg_rows + bi * s.t + (j - s.q);
  return bi * (s.q + s.t) + j;
}

template <int PER_LANE>
__global__ __launch_bounds__(256) void qe_fwd_kernel(QeShape s, const float *__restrict__ query, long query_bstride,
                                                     const long long *__restrict__ ids,
                                                     const float *__restrict__ word, const float *__restrict__ pos,
                                                     const float *__restrict__ gamma, const float *__restrict__ beta,
                                                     float eps, float p_drop, unsigned call_id,
                                                     const unsigned *__restrict__ rng_counter,
                                                     float *__restrict__ out, float *__restrict__ v_out,
                                                     float *__restrict__ mean_out, float *__restrict__ rstd_out,
                                                     unsigned short *__restrict__ mask_out) {
  const int lane = lane_id();
  const int rows = s.seg_rows > 0 ? 2 * s.seg_rows : s.b * (s.q + s.t);
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int cols = s.cols;
  // which token lives in this row?
  int bi = -1, j = 0;
  if (s.seg_rows > 0) {
    if (row < s.seg_rows) {
      if (row < s.b * s.q) { bi = row / s.q; j = row - bi * s.q; }
    } else {
      const int r2 = row - s.seg_rows;
      if (r2 < s.b * s.t) { bi = r2 / s.t; j = s.q + (r2 - bi * s.t); }
    }
  } else {
    bi = row / (s.q + s.t);
    j = row - bi * (s.q + s.t);
  }
  if (bi < 0) {  // padding row of the two-segment layout: finite (zero), never read by a live row
#pragma unroll
    for (int i = 0; i < PER_LANE; ++i)
      if (lane + 64 * i < cols) out[(size_t)row * cols + lane + 64 * i] = 0.f;
    if (lane == 0) { mean_out[row] = 0.f; rstd_out[row] = 0.f; }
    return;
  }
  const float *src0, *src1 = nullptr;
  if (j < s.q) {
    src0 = query + (size_t)bi * query_bstride + (size_t)j * cols;
  } else {
    long long id = ids[(size_t)bi * s.t + (j - s.q)];
    id = id < 0 ? 0 : (id >= s.vocab ? s.vocab - 1 : id);      // torch asserts; never read out of bounds here
    int p = s.pos_off + (j - s.q);
    p = p < s.pos_rows ? p : s.pos_rows - 1;
    src0 = word + (size_t)id * cols;
    src1 = pos + (size_t)p * cols;
  }
  float v[PER_LANE], w[PER_LANE], gam[PER_LANE], bet[PER_LANE];
#pragma unroll
  for (int i = 0; i < PER_LANE; ++i) {   // unconditional clamped loads (see rowops.hip)
    const int cc = min(lane + 64 * i, cols - 1);
    v[i] = src0[cc];
    w[i] = src1 ? src1[cc] : 0.f;
    gam[i] = gamma[cc];
    bet[i] = beta[cc];
  }
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < PER_LANE; ++i) {
    v[i] = (lane + 64 * i < cols) ? v[i] + w[i] : 0.f;
    sum += v[i];
  }
  const float mean = wave_allreduce_sum_f32(sum) / cols;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < PER_LANE; ++i)
    if (lane + 64 * i < cols) sq += (v[i] - mean) * (v[i] - mean);
  const float rstd = rsqrtf(wave_allreduce_sum_f32(sq) / cols + eps);
  const unsigned seed = qe_mix32((rng_counter ? *rng_counter : 0u) * 0x9E3779B9u + call_id);
  const unsigned thresh = (unsigned)((double)p_drop * 4294967296.0);
  const float keep_scale = 1.f / (1.f - p_drop);
  unsigned keep_bits = 0;
#pragma unroll
  for (int i = 0; i < PER_LANE; ++i) {
    const int c = lane + 64 * i;
    const unsigned idx = (unsigned)row * (unsigned)cols + (unsigned)c;
    const bool keep = (p_drop > 0.f) ? (qe_mix32(seed ^ idx * 0x9E3779B9u) >= thresh) : true;
    keep_bits |= (keep ? 1u : 0u) << i;
    if (c < cols) {
      const float y = (v[i] - mean) * rstd * gam[i] + bet[i];
      out[(size_t)row * cols + c] = keep ? (p_drop > 0.f ? y * keep_scale : y) : 0.f;
      v_out[(size_t)row * cols + c] = v[i];
    }
  }
  if (mask_out) mask_out[(size_t)row * 64 + lane] = (unsigned short)keep_bits;
  if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
}

// One workgroup per token index j, NW waves walking the batch.
//   dquery: query_shared ? (q, cols) summed over the batch : (b, q, cols)
//   dpos  : (pos_rows, cols), EVERY row written (rows no token of this batch sits on: zeros)
//   dword : (vocab, cols) zeroed by the caller, rows added with float atomics -- or, rows_out != NULL, the text
//           rows' gradients as (b*t, cols) rows in (b, t) order (pad-id rows zero) for the row exchange
//   partial: (q + t, 2*cols) per-workgroup [d gamma | d beta]
template <int PER_LANE, int NW>
__global__ __launch_bounds__(64 * NW) void qe_bwd_kernel(
    QeShape s, int query_shared, const long long *__restrict__ ids, const float *__restrict__ dy,
    const float *__restrict__ v, const float *__restrict__ mean, const float *__restrict__ rstd,
    const float *__restrict__ gamma, const unsigned short *__restrict__ mask, float p_drop,
    float *__restrict__ dquery, float *__restrict__ dpos, float *__restrict__ dword, float *__restrict__ rows_out,
    float *__restrict__ partial) {
  __shared__ float part[NW - 1][3][64 * PER_LANE];
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  const int j = blockIdx.x, cols = s.cols;
  const bool is_query = j < s.q;
  const float keep_scale = 1.f / (1.f - p_drop);
  float ag[PER_LANE], ab[PER_LANE], ax[PER_LANE], gam[PER_LANE];
#pragma unroll
  for (int i = 0; i < PER_LANE; ++i) {
    ag[i] = ab[i] = ax[i] = 0.f;
    gam[i] = gamma[min(lane + 64 * i, cols - 1)];
  }
  for (int bi = wave; bi < s.b; bi += NW) {
    const int row = qe_out_row(s, bi, j);
    const float mu = mean[row], rs = rstd[row];
    const unsigned keep_bits = (p_drop > 0.f) ? mask[(size_t)row * 64 + lane] : 0xFFFFu;
    const float *dyr = dy + (size_t)row * cols, *vr = v + (size_t)row * cols;
    float g[PER_LANE], xh[PER_LANE];
#pragma unroll
    for (int i = 0; i < PER_LANE; ++i) {
      const int cc = min(lane + 64 * i, cols - 1);
      g[i] = dyr[cc];
      xh[i] = vr[cc];
    }
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int i = 0; i < PER_LANE; ++i) {
      const bool live = lane + 64 * i < cols;
      float d = live ? g[i] : 0.f;
      d = ((keep_bits >> i) & 1u) ? (p_drop > 0.f ? d * keep_scale : d) : 0.f;   // dropout sits AFTER the norm here
      xh[i] = live ? (xh[i] - mu) * rs : 0.f;
      g[i] = d * gam[i];
      c1 += g[i];
      c2 += g[i] * xh[i];
      ag[i] += d * xh[i];
      ab[i] += d;
    }
    c1 = wave_allreduce_sum_f32(c1) / cols;
    c2 = wave_allreduce_sum_f32(c2) / cols;
    long long id = 0;
    if (!is_query) {
      id = ids[(size_t)bi * s.t + (j - s.q)];
      id = id < 0 ? 0 : (id >= s.vocab ? s.vocab - 1 : id);
    }
#pragma unroll
    for (int i = 0; i < PER_LANE; ++i) {
      const int c = lane + 64 * i;
      const float dv = rs * (g[i] - c1 - xh[i] * c2);
      ax[i] += dv;
      if (c < cols) {
        if (is_query) {
          if (!query_shared) dquery[((size_t)bi * s.q + j) * cols + c] = dv;
        } else if (rows_out) {
          rows_out[((size_t)bi * s.t + (j - s.q)) * cols + c] = (id == s.pad_id) ? 0.f : dv;
        } else if (id != s.pad_id) {      // nn.Embedding(padding_idx): that row gets no gradient
          atomicAdd(dword + (size_t)id * cols + c, dv);
        }
      }
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int i = 0; i < PER_LANE; ++i) {
      part[wave - 1][0][lane + 64 * i] = ag[i];
      part[wave - 1][1][lane + 64 * i] = ab[i];
      part[wave - 1][2][lane + 64 * i] = ax[i];
    }
  }
  __syncthreads();
  if (wave == 0) {
    float *dst = partial + (size_t)j * 2 * cols;
    int p = s.pos_off + (j - s.q);
    p = p < s.pos_rows ? p : s.pos_rows - 1;
#pragma unroll
    for (int i = 0; i < PER_LANE; ++i) {
      const int c = lane + 64 * i;
      if (c < cols) {
        float sg = ag[i], sb = ab[i], sx = ax[i];
#pragma unroll
        for (int w = 0; w < NW - 1; ++w) { sg += part[w][0][c]; sb += part[w][1][c]; sx += part[w][2][c]; }
        dst[c] = sg;
        dst[cols + c] = sb;
        if (is_query) {
          if (query_shared) dquery[(size_t)j * cols + c] = sx;
        } else {
          dpos[(size_t)p * cols + c] = sx;
        }
      }
    }
  }
  // the rows of the position table no token of this batch sits on: zeros, spread over the workgroups
  const int ntok = s.q + s.t;
  for (int r = j; r < s.pos_rows; r += ntok) {
    if (r >= s.pos_off && r < s.pos_off + s.t) continue;
    for (int c = threadIdx.x; c < cols; c += 64 * NW) dpos[(size_t)r * cols + c] = 0.f;
  }
}

// (1 - mask) * -10000 on a 0/1 mask of any of the dtypes a caller passes (Qformer.py:729-731)
template <typename T>
__global__ __launch_bounds__(256) void additive_mask_kernel(long n, const T *__restrict__ m, float *__restrict__ out) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = (1.f - (float)m[i]) * -10000.f;
}

// column sums of the (rows, cols) partial rows: small, one workgroup per 64 columns (the rowops fold kernel lives
// in another translation unit)
__global__ __launch_bounds__(256) void qe_fold_kernel(int rows, int cols, const float *__restrict__ x,
                                                      float *__restrict__ out) {
  __shared__ float part[4][64];
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  float acc = 0.f;
  if (c < cols)
    for (int r = wave; r < rows; r += 4) acc += x[(size_t)r * cols + c];
  part[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && c < cols) out[c] = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
}

}  // namespace

static int qe_check(int b, int q, int t, int cols, int seg_rows, int vocab, int pos_rows, int pos_off) {
  SIG3D_REQUIRE(b >= 0 && q >= 0 && t >= 0 && cols >= 1 && cols <= 64 * QE_MAX_PER_LANE, "hidden size must be <= 1024");
  SIG3D_REQUIRE(seg_rows == 0 || (seg_rows >= b * q && seg_rows >= b * t), "seg_rows must hold either segment");
  SIG3D_REQUIRE(t == 0 || (vocab >= 1 && pos_off >= 0 && pos_off + t <= pos_rows), "text tokens need the tables");
  const long rows = seg_rows > 0 ? 2L * seg_rows : (long)b * (q + t);
  SIG3D_REQUIRE(rows * cols < (1L << 32), "rows*cols must fit 32 bits (dropout hash index)");
  return 0;
}

extern "C" int sig3d_qformer_embed_fwd(int b, int q, int t, int cols, int seg_rows, const float *query,
                                       long query_bstride, const long long *ids, const float *word, int vocab,
                                       const float *pos, int pos_rows, int pos_off, const float *gamma,
                                       const float *beta, float eps, float p_drop, unsigned call_id,
                                       const unsigned *rng_counter, float *out, float *v, float *mean, float *rstd,
                                       unsigned short *mask, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (qe_check(b, q, t, cols, seg_rows, vocab, pos_rows, pos_off)) return 1;
  SIG3D_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "dropout probability must be in [0, 1)");
  SIG3D_REQUIRE(p_drop == 0.f || mask != nullptr, "a mask buffer is required when p_drop > 0");
  const int rows = seg_rows > 0 ? 2 * seg_rows : b * (q + t);
  if (rows == 0) return 0;
  QeShape s{b, q, t, cols, seg_rows, vocab, pos_rows, pos_off, -1};
  const dim3 grid(sig3d_ceil_div(rows, 4));
  if (cols <= 64 * 12)
    hipLaunchKernelGGL(qe_fwd_kernel<12>, grid, dim3(256), 0, stream, s, query, query_bstride, ids, word, pos, gamma,
                       beta, eps, p_drop, call_id, rng_counter, out, v, mean, rstd, mask);
  else
    hipLaunchKernelGGL(qe_fwd_kernel<QE_MAX_PER_LANE>, grid, dim3(256), 0, stream, s, query, query_bstride, ids, word,
                       pos, gamma, beta, eps, p_drop, call_id, rng_counter, out, v, mean, rstd, mask);
  SIG3D_LAUNCH_CHECK("qe_fwd_kernel");
  return 0;
}

extern "C" int sig3d_qformer_embed_bwd(int b, int q, int t, int cols, int seg_rows, int query_shared,
                                       const long long *ids, int vocab, int pos_rows, int pos_off, int pad_id,
                                       const float *dy, const float *v, const float *mean, const float *rstd,
                                       const float *gamma, const unsigned short *mask, float p_drop, float *dquery,
                                       float *dpos, float *dword, float *rows_out, float *dgamma_dbeta,
                                       float *workspace, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (qe_check(b, q, t, cols, seg_rows, vocab, pos_rows, pos_off)) return 1;
  SIG3D_REQUIRE(p_drop == 0.f || mask != nullptr, "the forward's mask buffer is required when p_drop > 0");
  SIG3D_REQUIRE(dgamma_dbeta != nullptr && workspace != nullptr, "dgamma_dbeta (2*cols) and workspace ((q+t)*2*cols) are required");
  SIG3D_REQUIRE(t == 0 || (dpos != nullptr && (dword != nullptr || rows_out != nullptr)), "text tokens need gradient buffers");
  SIG3D_REQUIRE(q == 0 || dquery != nullptr, "dquery is required");
  if (b == 0 || q + t == 0) {
    SIG3D_HIP_TRY(hipMemsetAsync(dgamma_dbeta, 0, sizeof(float) * 2 * cols, stream));
    return 0;
  }
  QeShape s{b, q, t, cols, seg_rows, vocab, pos_rows, pos_off, pad_id};
  if (cols <= 64 * 12)
    hipLaunchKernelGGL((qe_bwd_kernel<12, 8>), dim3(q + t), dim3(64 * 8), 0, stream, s, query_shared, ids, dy, v,
                       mean, rstd, gamma, mask, p_drop, dquery, dpos, dword, rows_out, workspace);
  else
    hipLaunchKernelGGL((qe_bwd_kernel<QE_MAX_PER_LANE, 4>), dim3(q + t), dim3(64 * 4), 0, stream, s, query_shared,
                       ids, dy, v, mean, rstd, gamma, mask, p_drop, dquery, dpos, dword, rows_out, workspace);
  SIG3D_LAUNCH_CHECK("qe_bwd_kernel");
  hipLaunchKernelGGL(qe_fold_kernel, dim3(sig3d_ceil_div(2 * cols, 64)), dim3(256), 0, stream, q + t, 2 * cols,
                     workspace, dgamma_dbeta);
  SIG3D_LAUNCH_CHECK("qe_fold_kernel");
  return 0;
}

extern "C" int sig3d_additive_mask(long n, const void *mask, int kind, float *out, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SIG3D_REQUIRE(n >= 0 && (n == 0 || (mask != nullptr && out != nullptr)), "bad arguments");
  if (n == 0) return 0;
  const dim3 grid((unsigned)((n + 255) / 256));
  switch (kind) {
    case 0: hipLaunchKernelGGL(additive_mask_kernel<float>, grid, dim3(256), 0, stream, n, (const float *)mask, out); break;
    case 1: hipLaunchKernelGGL(additive_mask_kernel<long long>, grid, dim3(256), 0, stream, n, (const long long *)mask, out); break;
    case 2: hipLaunchKernelGGL(additive_mask_kernel<int>, grid, dim3(256), 0, stream, n, (const int *)mask, out); break;
    case 3: hipLaunchKernelGGL(additive_mask_kernel<unsigned char>, grid, dim3(256), 0, stream, n, (const unsigned char *)mask, out); break;
    default: SIG3D_REQUIRE(false, "kind: 0 f32, 1 i64, 2 i32, 3 u8 / bool");
  }
  SIG3D_LAUNCH_CHECK("additive_mask_kernel");
  return 0;
}
