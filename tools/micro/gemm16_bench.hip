// tools/micro/gemm16_bench.hip -- design-space probe + correctness check for csrc/gemm16_core.h (not product code).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o gemm16_bench gemm16_bench.hip -lrocblas && ./gemm16_bench
// The Q-Former products of the bench step (B = 8: 256 query rows + 160 text rows = 416 live rows; 2048 scene
// tokens), forward (weights k-contiguous) and input-gradient (weights n-contiguous) forms, every tiling of the
// list below, weights rotating over enough copies to stay HBM-cold (a step touches every weight once per pass),
// rocBLAS's default pick beside it.  Checks sampled outputs against a double-precision host product.
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../situation3d_amd/csrc/gemm16_core.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Shape {
  const char *name;
  int bmode, batch, M, N, K, act;
  bool bias, addend;
  double lib_us;   // the tuned library's time in the step (profiles/r03_a_step.md), for the table
};

static const Shape shapes[] = {
    {"fwd QKV       416x2304x768 +b", 0, 1, 416, 2304, 768, 0, true, false, 20.2},
    {"fwd out-proj  416x768x768", 0, 1, 416, 768, 768, 0, false, false, 10.2},
    {"fwd FFN-up  2x256x3072x768 +b+gelu", 0, 2, 256, 3072, 768, 1, true, false, 23.3 + 5.7},
    {"fwd FFN-dn  2x256x768x3072", 0, 2, 256, 768, 3072, 0, false, false, 31.1},
    {"fwd cross-Q   256x768x768 +b", 0, 1, 256, 768, 768, 0, true, false, 10.7},
    {"fwd KV-proj  2048x9216x256 +b", 0, 1, 2048, 9216, 256, 0, true, false, 91.0},
    {"dX  datt      416x768x768", 1, 1, 416, 768, 768, 0, false, false, 7.8},
    {"dX  dres      416x768x2304 +C", 1, 1, 416, 768, 2304, 0, false, true, 28.9},
    {"dX  gact    2x256x3072x768 *gelu'", 1, 2, 256, 3072, 768, 2, false, false, 21.9 + 5.4},
    {"dX  gx      2x256x768x3072 +C", 1, 2, 256, 768, 3072, 0, false, true, 28.9},
    {"dX  g_enc    2048x256x9216", 1, 1, 2048, 256, 9216, 0, false, false, 81.3},
};

struct Config { const char *name; int id; };
// id -> template instance
#define CONFIGS(X)            \
  X(0, 2, 2, 2, 2, 4, 2) \
  X(1, 2, 2, 2, 2, 6, 2) \
  X(2, 1, 2, 2, 2, 4, 3) \
  X(3, 1, 2, 2, 4, 4, 2) \
  X(4, 1, 2, 4, 2, 4, 2) \
  X(5, 2, 2, 2, 4, 4, 1) \
  X(6, 2, 2, 4, 2, 4, 1) \
  X(7, 1, 3, 2, 4, 4, 1) \
  X(8, 1, 2, 4, 4, 4, 1) \
  X(9, 2, 3, 2, 2, 4, 2) \
  X(10, 1, 3, 2, 2, 4, 2) \
  X(11, 1, 4, 2, 2, 4, 2) \
  X(12, 1, 4, 2, 4, 4, 1) \
  X(13, 2, 2, 2, 4, 6, 1)

static hipError_t run_config(int id, const gemm16::Problem &p, int bmode, hipStream_t s) {
  switch (id) {
#define X(ID, AB, BB, WGM, WGN, PF, OCC) case ID: return gemm16::launch<AB, BB, WGM, WGN, PF, OCC>(p, bmode, s);
    CONFIGS(X)
#undef X
  }
  return hipErrorInvalidValue;
}
static void config_dims(int id, int *tm, int *tn, int *kw) {
  *kw = 1;
  if (id == 22) { *tm = 128; *tn = 128; return; }
  if (id == 23) { *tm = 64; *tn = 256; return; }
  if (id >= 20) { *tm = 64; *tn = 128; return; }
  switch (id) {
#define X(ID, AB, BB, WGM, WGN, PF, OCC) case ID: *tm = 16 * AB * WGM; *tn = 16 * BB * WGN; return;
    CONFIGS(X)
#undef X
  }
}
static std::string config_name(int id) {
  char buf[64];
  if (id == 20) return "x6 w32x32 g2x4";
  if (id == 21) return "x6 w32x64 g2x2";
  if (id == 22) return "x6 w32x64 g4x2";
  if (id == 23) return "x6 w32x64 g2x4";
  switch (id) {
#define X(ID, AB, BB, WGM, WGN, PF, OCC) case ID: snprintf(buf, 64, "w%dx%d g%dx%d p%d o%d", 16 * AB, 16 * BB, WGM, WGN, PF, OCC); return buf;
    CONFIGS(X)
#undef X
  }
  return "?";
}
constexpr int NCONFIG = 14;

static float frand() { return (float)((rand() & 0xffff) / 32768.0 - 1.0); }

int main(int argc, char **argv) {
  const int only_shape = argc > 1 ? atoi(argv[1]) : -1;
  const bool quick = argc > 2 && !strcmp(argv[2], "quick");
  // ./gemm16_bench <shape> one <config> <splits>: only that combination (for rocprofv3 --pmc passes)
  const bool one = argc > 4 && !strcmp(argv[2], "one");
  const int one_cfg = one ? atoi(argv[3]) : -1, one_splits = one ? atoi(argv[4]) : -1;
  hipStream_t stream;
  CHECK(hipStreamCreate(&stream));
  rocblas_handle h;
  rocblas_create_handle(&h);
  rocblas_set_stream(h, stream);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int REPS = 48;

  int si = 0;
  for (const Shape &sh : shapes) {
    if (only_shape >= 0 && si++ != only_shape) continue;
    const size_t a_el = (size_t)sh.batch * sh.M * sh.K, b_el = (size_t)sh.batch * sh.N * sh.K, c_el = (size_t)sh.batch * sh.M * sh.N;
    // enough weight copies for ~400 MB
    int nbuf = (int)(400e6 / (b_el * 4)) + 1;
    if (nbuf > 48) nbuf = 48;
    if (nbuf < 4) nbuf = 4;
    std::vector<float> hA(a_el), hB(b_el), hbias((size_t)sh.batch * sh.N), hadd(c_el), haux(c_el);
    srand(1234);
    const float wscale = 1.f / sqrtf((float)sh.K);
    for (auto &v : hA) v = frand();
    for (auto &v : hB) v = frand() * wscale * 2.f;
    for (auto &v : hbias) v = frand() * 0.1f;
    for (auto &v : hadd) v = frand();
    for (auto &v : haux) v = frand() * 2.f;
    float *dA, *dB, *dC, *dbias, *dadd, *daux;
    const int max_splits = 8;
    CHECK(hipMalloc(&dA, a_el * 4));
    CHECK(hipMalloc(&dB, b_el * 4 * nbuf));
    CHECK(hipMalloc(&dC, c_el * 4 * max_splits));
    CHECK(hipMalloc(&dbias, hbias.size() * 4));
    CHECK(hipMalloc(&dadd, c_el * 4));
    CHECK(hipMalloc(&daux, c_el * 4));
    CHECK(hipMemcpy(dA, hA.data(), a_el * 4, hipMemcpyHostToDevice));
    for (int i = 0; i < nbuf; ++i) CHECK(hipMemcpy(dB + (size_t)i * b_el, hB.data(), b_el * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dbias, hbias.data(), hbias.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dadd, hadd.data(), c_el * 4, hipMemcpyHostToDevice));

    gemm16::Problem p;
    memset(&p, 0, sizeof(p));
#ifdef GEMM16_TIMING
    static unsigned long long *dbg_always = nullptr;
    if (!dbg_always) CHECK(hipMalloc(&dbg_always, 64 * 8));
    p.dbg = dbg_always;
#endif
    p.A = dA; p.B = dB; p.C = dC; p.Cs = dC + c_el;
    p.bias = sh.bias ? dbias : nullptr;
    p.addend = sh.addend ? dadd : nullptr;
    p.aux = sh.act ? daux : nullptr;
    p.M = sh.M; p.N = sh.N; p.K = sh.K;
    p.lda = sh.K; p.ldb = sh.bmode == 0 ? sh.K : sh.N; p.ldc = sh.N;
    p.sA = (long)sh.M * sh.K; p.sB = (long)sh.N * sh.K; p.sC = (long)sh.M * sh.N; p.sBias = sh.N;
    p.slab = (long)c_el;
    p.batch = sh.batch; p.act = sh.act;
    const double gflop = 2.0 * sh.batch * sh.M * (double)sh.N * sh.K * 1e-9;

    // ---- rocBLAS default pick (column-major view: C^T = op(W) * X^T)
    double rb_us = 0;
    {
      const float one = 1.f, zero = 0.f;
      auto call = [&](int i) {
        const float *Bw = dB + (size_t)(i % nbuf) * b_el;
        if (sh.bmode == 0)
          rocblas_sgemm_strided_batched(h, rocblas_operation_transpose, rocblas_operation_none, sh.N, sh.M, sh.K, &one, Bw,
                                        sh.K, (rocblas_stride)sh.N * sh.K, dA, sh.K, (rocblas_stride)sh.M * sh.K, &zero, dC,
                                        sh.N, (rocblas_stride)sh.M * sh.N, sh.batch);
        else
          rocblas_sgemm_strided_batched(h, rocblas_operation_none, rocblas_operation_none, sh.N, sh.M, sh.K, &one, Bw, sh.N,
                                        (rocblas_stride)sh.N * sh.K, dA, sh.K, (rocblas_stride)sh.M * sh.K, &zero, dC, sh.N,
                                        (rocblas_stride)sh.M * sh.N, sh.batch);
      };
      for (int i = 0; i < 4; ++i) call(i);
      CHECK(hipStreamSynchronize(stream));
      CHECK(hipEventRecord(e0, stream));
      for (int i = 0; i < REPS; ++i) call(i);
      CHECK(hipEventRecord(e1, stream));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      rb_us = ms * 1e3 / REPS;
    }
    printf("\n== %s  (%.2f GFLOP; tuned library in the step %.1f us; rocBLAS default here %.1f us = %.0f TF)\n", sh.name, gflop,
           sh.lib_us, rb_us, gflop / rb_us * 1e-3 * 1e6 * 1e-3);

    for (int idx = 0; idx < NCONFIG + 4; ++idx) {
      const int id = idx < NCONFIG ? idx : 20 + idx - NCONFIG;     // 20, 21: the bf16 x 6 core
      if (one && id != one_cfg) continue;
      if (quick && idx < NCONFIG && getenv("X6_ONLY")) continue;
      int tm, tn, kw;
      config_dims(id, &tm, &tn, &kw);
      const int tiles = ((sh.M + tm - 1) / tm) * ((sh.N + tn - 1) / tn) * sh.batch;
      const int chunks = sh.K / 32;
      for (int splits = 1; splits <= max_splits; ++splits) {
        if (one && splits != one_splits) continue;
        if (sh.act && splits > 1) break;
        const int wgs = tiles * splits;
        if (wgs > 1280 && splits > 1) break;
        if (splits > 1 && wgs < 120) continue;
        if (chunks / splits < 2 * kw) break;
        if (quick && !(wgs >= 120 && wgs <= 1100)) continue;
        p.splits = splits;
        // correctness (first use of this config on this shape: sampled)
        CHECK(hipMemsetAsync(dC, 0xff, c_el * 4 * splits, stream));
        if (sh.act) CHECK(hipMemcpyAsync(daux, haux.data(), c_el * 4, hipMemcpyHostToDevice, stream));
        p.B = dB;
        hipError_t e = run_config(id, p, sh.bmode, stream);
        if (e != hipSuccess) { printf("   %-16s splits %d: launch error %s\n", config_name(id).c_str(), splits, hipGetErrorString(e)); continue; }
        CHECK(hipStreamSynchronize(stream));
        std::vector<float> hC(c_el * splits), hAuxOut;
        CHECK(hipMemcpy(hC.data(), dC, c_el * 4 * splits, hipMemcpyDeviceToHost));
        if (sh.act == 1) { hAuxOut.resize(c_el); CHECK(hipMemcpy(hAuxOut.data(), daux, c_el * 4, hipMemcpyDeviceToHost)); }
        double worst = 0;
        for (int t = 0; t < 400; ++t) {
          const int b = rand() % sh.batch;
          int m = rand() % sh.M, n = rand() % sh.N;
          if (t < 8) { m = sh.M - 1 - (t & 1); n = sh.N - 1 - (t >> 1); }
          double ref = 0;
          for (int k = 0; k < sh.K; ++k) {
            const double av = hA[((size_t)b * sh.M + m) * sh.K + k];
            const double bv = sh.bmode == 0 ? hB[((size_t)b * sh.N + n) * sh.K + k] : hB[(size_t)b * sh.N * sh.K + (size_t)k * sh.N + n];
            ref += av * bv;
          }
          if (sh.bias) ref += hbias[(size_t)b * sh.N + n];
          const size_t ci = ((size_t)b * sh.M + m) * sh.N + n;
          if (sh.act == 1) {
            const double pre = ref;
            if (fabs(hAuxOut[ci] - pre) > 1e-4 * (1 + fabs(pre))) worst = fmax(worst, fabs(hAuxOut[ci] - pre));
            ref = 0.5 * pre * (1 + erf(pre * 0.70710678118654752440));
          } else if (sh.act == 2) {
            const double u = haux[ci];
            ref *= 0.5 * (1 + erf(u * 0.70710678118654752440)) + u * 0.39894228040143267794 * exp(-0.5 * u * u);
          }
          if (sh.addend) ref += hadd[ci];
          double got = 0;
          for (int z = 0; z < splits; ++z) got += hC[(size_t)z * c_el + ci];
          worst = fmax(worst, fabs(got - ref) / (1 + fabs(ref)));
        }
        // timing
        for (int i = 0; i < 3; ++i) { p.B = dB + (size_t)(i % nbuf) * b_el; run_config(id, p, sh.bmode, stream); }
        CHECK(hipStreamSynchronize(stream));
        CHECK(hipEventRecord(e0, stream));
        for (int i = 0; i < REPS; ++i) { p.B = dB + (size_t)((i + 3) % nbuf) * b_el; run_config(id, p, sh.bmode, stream); }
        CHECK(hipEventRecord(e1, stream));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / REPS;
#ifdef GEMM16_TIMING
        {
          unsigned long long *dbg, h[60];
          CHECK(hipMalloc(&dbg, 60 * 8));
          CHECK(hipMemset(dbg, 0, 60 * 8));
          p.dbg = dbg; p.B = dB;
          run_config(id, p, sh.bmode, stream);
          CHECK(hipStreamSynchronize(stream));
          CHECK(hipMemcpy(h, dbg, 60 * 8, hipMemcpyDeviceToHost));
          printf("      stamps (cycles since kernel start of wg 0 / wave 0):");
          for (int i = 1; i < 60 && h[i]; ++i) printf(" %llu", h[i] - h[i - 1]);
          printf("\n");
          CHECK(hipFree(dbg));
          p.dbg = dbg_always;
        }
#endif
        printf("   %-16s splits %d  wgs %4d  %7.2f us  %6.1f TF  err %.1e%s\n", config_name(id).c_str(), splits, wgs, us,
               gflop / us * 1e3, worst, worst > 2e-5 ? "  <-- WRONG" : "");
        fflush(stdout);
      }
    }
    CHECK(hipFree(dA)); CHECK(hipFree(dB)); CHECK(hipFree(dC)); CHECK(hipFree(dbias)); CHECK(hipFree(dadd)); CHECK(hipFree(daux));
  }
  return 0;
}
