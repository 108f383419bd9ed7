// tools/micro/gemm_variants.hip -- design-space probe for csrc/gemm.hip (not product code).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o gemm_variants gemm_variants.hip && ./gemm_variants
// C(m,n) = X(m,k) W(n,k)^T in exact-f32 MFMA, interior tiles only, variants over:
//   WM x WN wave tiles (workgroup tile 64WM x 64WN), K chunk GK, global prefetch distance PF (chunks),
//   operand-read interleave SCHED (sched_group_barrier), workgroups per CU via launch bounds.
// Weights rotate over enough buffers to stay HBM-cold.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int TT, int GK>
struct Tile {  // k-contiguous global operand -> LDS [k][m], odd row stride
  static constexpr int PER = TT * GK / 4 / 256;
  static constexpr int LD = TT + 1;
  f32x4 r[PER];
  __device__ __forceinline__ void load(const float *__restrict__ p, int ld, int m0, int k0, int tid) {
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int u = tid + 256 * q;
      const int m = m0 + u / (GK / 4), k = k0 + 4 * (u % (GK / 4));
      r[q] = *reinterpret_cast<const f32x4 *>(p + (size_t)m * ld + k);
    }
  }
  __device__ __forceinline__ void store(float *__restrict__ s, int tid) const {
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int u = tid + 256 * q;
      const int m = u / (GK / 4), k = 4 * (u % (GK / 4));
#pragma unroll
      for (int e = 0; e < 4; ++e) s[(k + e) * LD + m] = r[q][e];
    }
  }
};

template <int WM, int WN, int GK, int PF, int SCHED, int OCC, int KNOCK = 0>
__global__ __launch_bounds__(256, OCC) void gemm_nt(int M, int N, int K, int ksplit, const float *__restrict__ A,
                                                    const float *__restrict__ B, float *__restrict__ C, int ntm) {
  constexpr int TM = 64 * WM, TN = 64 * WN;
  typedef Tile<TM, GK> TA;
  typedef Tile<TN, GK> TB;
  constexpr int LDA = TA::LD, LDB = TB::LD;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *const s_a = smem;
  float *const s_b = smem + 2 * GK * LDA;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int T = gridDim.x;
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  int base = 0;
  for (int y = 0; y < xcd; ++y) base += (T - y + 7) >> 3;
  const int id = base + local;
  const int tn = id / ntm, tm = id - tn * ntm;
  const int m0 = tm * TM, n0 = tn * TN;
  const int kz = blockIdx.z;
  const int nall = K / GK;
  const int c_lo = nall * kz / ksplit, c_hi = nall * (kz + 1) / ksplit;
  const int nchunks = c_hi - c_lo;
  f32x16 acc[WM][WN];
#pragma unroll
  for (int a = 0; a < WM; ++a)
#pragma unroll
    for (int b = 0; b < WN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  TA ta[PF];
  TB tb[PF];
  // prologue: chunk 0 -> LDS stage 0, chunks 1..PF-1 (PF = 2) in flight
  ta[0].load(A, K, m0, c_lo * GK, tid);
  tb[0].load(B, K, n0, c_lo * GK, tid);
  if (PF == 2) {
    ta[1].load(A, K, m0, (c_lo + min(1, nchunks - 1)) * GK, tid);
    tb[1].load(B, K, n0, (c_lo + min(1, nchunks - 1)) * GK, tid);
  }
  ta[0].store(s_a, tid);
  tb[0].store(s_b, tid);
  __syncthreads();

  auto body = [&](int c, auto par_) {
    constexpr int par = decltype(par_)::value;   // register buffer that holds chunk c + 1 (PF = 2) / receives it (PF = 1)
    const int st = c & 1;
    if (KNOCK & 2) {
    } else if (PF == 1) {
      // UNCONDITIONAL (clamped) loads: a load under `if` ends its basic block, hipcc's s_waitcnt pass then
      // merges the pending-load state conservatively and drains vmcnt(0) at the next use
      const int cn = min(c + 1, nchunks - 1);
      ta[0].load(A, K, m0, (c_lo + cn) * GK, tid);
      tb[0].load(B, K, n0, (c_lo + cn) * GK, tid);
    } else {
      // chunk c+2 into the buffer chunk c was in (already in LDS)
      const int cn = min(c + 2, nchunks - 1);
      ta[par ^ 1].load(A, K, m0, (c_lo + cn) * GK, tid);
      tb[par ^ 1].load(B, K, n0, (c_lo + cn) * GK, tid);
    }
    __builtin_amdgcn_sched_barrier(0);   // the loads stay HERE: the scheduler otherwise sinks them to the stores
    const float *sa = s_a + st * (GK * LDA) + wm * (32 * WM) + l31 + half * LDA;
    const float *sb = s_b + st * (GK * LDB) + wn * (32 * WN) + l31 + half * LDB;
    constexpr int GS = 4, NG = GK / 2 / GS;
    float av[2][GS][WM], bv[2][GS][WN];
#pragma unroll
    for (int t = 0; t < GS; ++t) {
#pragma unroll
      for (int a = 0; a < WM; ++a) av[0][t][a] = (KNOCK & 1) ? (float)(lane + t) : sa[(2 * t) * LDA + 32 * a];
#pragma unroll
      for (int b = 0; b < WN; ++b) bv[0][t][b] = (KNOCK & 1) ? (float)(lane - t) : sb[(2 * t) * LDB + 32 * b];
    }
    if (SCHED) __builtin_amdgcn_sched_group_barrier(0x100, GS * (WM + WN), 0);
#pragma unroll
    for (int gq = 0; gq < NG; ++gq) {
      if (gq + 1 < NG) {
#pragma unroll
        for (int t = 0; t < GS; ++t) {
#pragma unroll
          for (int a = 0; a < WM; ++a) av[(gq + 1) & 1][t][a] = (KNOCK & 1) ? (float)(lane + t + gq) : sa[(2 * ((gq + 1) * GS + t)) * LDA + 32 * a];
#pragma unroll
          for (int b = 0; b < WN; ++b) bv[(gq + 1) & 1][t][b] = (KNOCK & 1) ? (float)(lane - t - gq) : sb[(2 * ((gq + 1) * GS + t)) * LDB + 32 * b];
        }
      }
#pragma unroll
      for (int t = 0; t < GS; ++t) {
#pragma unroll
        for (int a = 0; a < WM; ++a)
#pragma unroll
          for (int b = 0; b < WN; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[gq & 1][t][a], bv[gq & 1][t][b], acc[a][b], 0, 0, 0);
        if (SCHED) {
          __builtin_amdgcn_sched_group_barrier(0x8, WM * WN, 0);
          if (gq + 1 < NG) __builtin_amdgcn_sched_group_barrier(0x100, WM + WN, 0);
        }
      }
    }
    if (!(KNOCK & 2)) {   // unconditional too (the last chunk's store is never read)
      const int src = (PF == 1) ? 0 : par;
      ta[src].store(s_a + (st ^ 1) * (GK * LDA), tid);
      tb[src].store(s_b + (st ^ 1) * (GK * LDB), tid);
    }
    if (!(KNOCK & 4)) __syncthreads();
  };
  for (int c = 0; c < nchunks; c += 2) {
    body(c, std::integral_constant<int, 1>());
    if (c + 1 < nchunks) body(c + 1, std::integral_constant<int, 0>());
  }
#pragma unroll
  for (int a = 0; a < WM; ++a)
#pragma unroll
    for (int b = 0; b < WN; ++b) {
      const int col = n0 + wn * (32 * WN) + 32 * b + l31;
      const int row0 = m0 + wm * (32 * WM) + 32 * a + 4 * half;
      float *cp = C + (size_t)row0 * N + col;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int dr = (r & 3) + 8 * (r >> 2);
        if (ksplit > 1) unsafeAtomicAdd(cp + (size_t)dr * N, acc[a][b][r]);
        else cp[(size_t)dr * N] = acc[a][b][r];
      }
    }
}

struct Shape { const char *name; int m, n, k; };

template <int WM, int WN, int GK, int PF, int SCHED, int OCC, int KNOCK = 0>
void run(const Shape &s, int ksplit, const float *A, std::vector<float *> &W, float *C, const std::vector<float> &ref) {
  constexpr int TM = 64 * WM, TN = 64 * WN;
  if (s.m % TM || s.n % TN || s.k % GK || (s.k / GK) < ksplit) return;
  const size_t lds = sizeof(float) * 2 * GK * (TM + 1 + TN + 1);
  auto kern = gemm_nt<WM, WN, GK, PF, SCHED, OCC, KNOCK>;
  CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const int ntm = s.m / TM, ntn = s.n / TN;
  dim3 grid(ntm * ntn, 1, ksplit);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int reps = W.size() > 1 ? (int)W.size() : 20;
  for (int warm = 0; warm < 2; ++warm) {
    if (warm) CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) {
      hipLaunchKernelGGL(kern, grid, dim3(256), lds, 0, s.m, s.n, s.k, ksplit, A, W[i % W.size()], C, ntm);
    }
    if (warm) CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
  }
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  // correctness (ksplit == 1 overwrites; ksplit > 1 accumulated reps + warm-up times: check by re-running into zeros)
  CHECK(hipMemset(C, 0, sizeof(float) * s.m * s.n));
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, 0, s.m, s.n, s.k, ksplit, A, W[0], C, ntm);
  std::vector<float> out(256);
  CHECK(hipMemcpy(out.data(), C, sizeof(float) * 256, hipMemcpyDeviceToHost));
  double err = 0;
  for (int i = 0; i < 256; ++i) err = fmax(err, fabs(out[i] - ref[i]));
  const double us = ms * 1e3 / reps;
  if (KNOCK) printf("  [knock-out %d: %s%s%s]", KNOCK, KNOCK & 1 ? "no LDS operand reads " : "", KNOCK & 2 ? "no global loads / LDS stores " : "", KNOCK & 4 ? "no barrier" : "");
  printf("  %dx%d gk%-2d pf%d s%d occ%d ks%-2d (%4d WGs, %5.1f KB lds): %7.1f us %6.1f TF  err %.1e\n", TM, TN, GK, PF, SCHED,
         OCC, ksplit, grid.x * ksplit, lds / 1024.0, us, 2.0 * s.m * s.n * s.k / us / 1e6, err);
}

int main(int argc, char **argv) {
  // ./gemm_variants            : every shape, every variant
  // ./gemm_variants S KS       : shape S only, split KS only (for rocprofv3 --pmc runs)
  const int only_shape = argc > 1 ? atoi(argv[1]) : -1, only_ks = argc > 2 ? atoi(argv[2]) : -1;
  const int hot = argc > 3 ? atoi(argv[3]) : 0;   // 1: one weight buffer (L2 / MALL-hot) instead of rotating HBM-cold ones
  const Shape shapes[] = {{"ffn-up (one half x2 rows)", 512, 3072, 768}, {"ffn-down", 512, 768, 3072}, {"qkv", 512, 2304, 768},
                          {"out-proj", 512, 768, 768}, {"cross-kv x6", 2048, 9216, 256}};
  int shape_idx = -1;
  for (const Shape &s : shapes) {
    if (++shape_idx != only_shape && only_shape >= 0) continue;
    printf("%s  m %d n %d k %d  (%.2f GFLOP, %.1f us at 157 TF)\n", s.name, s.m, s.n, s.k, 2e-9 * s.m * s.n * s.k,
           2.0 * s.m * s.n * s.k / 157e6);
    std::vector<float> hA((size_t)s.m * s.k), hW((size_t)s.n * s.k);
    for (auto &v : hA) v = (float)rand() / RAND_MAX - 0.5f;
    for (auto &v : hW) v = (float)rand() / RAND_MAX - 0.5f;
    std::vector<float> ref(256);
    for (int j = 0; j < 256; ++j) { double t = 0; for (int k = 0; k < s.k; ++k) t += (double)hA[k] * hW[(size_t)j * s.k + k]; ref[j] = (float)t; }
    float *A, *C;
    CHECK(hipMalloc(&A, sizeof(float) * hA.size()));
    CHECK(hipMalloc(&C, sizeof(float) * s.m * s.n));
    CHECK(hipMemcpy(A, hA.data(), sizeof(float) * hA.size(), hipMemcpyHostToDevice));
    const int nbuf = hot ? 1 : (int)(400e6 / (4.0 * s.n * s.k)) + 2;
    std::vector<float *> W(nbuf);
    for (auto &w : W) { CHECK(hipMalloc(&w, sizeof(float) * hW.size())); CHECK(hipMemcpy(w, hW.data(), sizeof(float) * hW.size(), hipMemcpyHostToDevice)); }
    for (int ks : {1, 2, 3, 4, 6, 8}) {
      if (only_ks >= 0 && ks != only_ks) continue;
      if (argc > 4) {   // knock-out experiments only
        run<1, 1, 32, 1, 1, 1>(s, ks, A, W, C, ref);
        run<1, 1, 32, 2, 1, 1>(s, ks, A, W, C, ref);
        run<1, 1, 64, 2, 1, 1>(s, ks, A, W, C, ref);
        run<1, 2, 32, 2, 1, 1>(s, ks, A, W, C, ref);
        run<2, 1, 32, 2, 1, 1>(s, ks, A, W, C, ref);
        run<1, 1, 32, 1, 1, 1, 1>(s, ks, A, W, C, ref);
        run<1, 1, 32, 1, 1, 1, 2>(s, ks, A, W, C, ref);
        run<1, 1, 32, 1, 1, 1, 3>(s, ks, A, W, C, ref);
        run<1, 1, 32, 1, 1, 1, 7>(s, ks, A, W, C, ref);
        run<1, 2, 32, 1, 1, 1, 7>(s, ks, A, W, C, ref);
        continue;
      }
      run<1, 1, 32, 1, 0, 1>(s, ks, A, W, C, ref);
      run<1, 1, 32, 1, 1, 1>(s, ks, A, W, C, ref);
      run<1, 1, 32, 2, 1, 1>(s, ks, A, W, C, ref);
      run<1, 1, 64, 1, 1, 1>(s, ks, A, W, C, ref);
      run<1, 1, 64, 2, 1, 1>(s, ks, A, W, C, ref);
      run<2, 1, 32, 2, 1, 1>(s, ks, A, W, C, ref);
      run<1, 2, 32, 2, 1, 1>(s, ks, A, W, C, ref);
      run<2, 2, 32, 2, 1, 1>(s, ks, A, W, C, ref);
      run<2, 2, 16, 2, 1, 1>(s, ks, A, W, C, ref);
      run<2, 1, 64, 2, 1, 1>(s, ks, A, W, C, ref);
    }
    for (auto &w : W) CHECK(hipFree(w));
    CHECK(hipFree(A)); CHECK(hipFree(C));
  }
  return 0;
}
