// tools/micro/gemmp_bench.hip -- design-space probe + correctness check for csrc/gemmp_core.h (not product code).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o gemmp_bench gemmp_bench.hip -lrocblas && ./gemmp_bench [shape] [config]
// The Q-Former products of the bench step (B = 8: 416 live rows, 2048 scene tokens) on PRE-SPLIT operands (three bf16
// planes, made on the host here): forward (both operands reduction-contiguous), input gradient (weight planes read
// through the transposing LDS read), layer-batched weight gradients (both operands through it).  Weights rotate over
// enough copies to stay HBM-cold; rocBLAS's default f32 pick beside it; sampled outputs against a double product.
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../situation3d_amd/csrc/gemmp_core.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Shape {
  const char *name;
  int modes, batch, M, N, K, act;
  bool bias, addend, planes_out;
  double lib_us;   // the tuned library's time in the step (profiles/r04_a_step.md / r04_gemmx6.md)
};

static const Shape shapes[] = {
    {"fwd QKV       416x2304x768 +b", 0, 1, 416, 2304, 768, 0, true, false, false, 17.3},
    {"fwd out-proj  416x768x768", 0, 1, 416, 768, 768, 0, false, false, false, 8.3},
    {"fwd FFN-up  2x256x3072x768 +b+gelu ->planes", 0, 2, 256, 3072, 768, 1, true, false, true, 35.0},
    {"fwd FFN-dn  2x256x768x3072", 0, 2, 256, 768, 3072, 0, false, false, false, 25.8},
    {"fwd cross-Q   256x768x768 +b", 0, 1, 256, 768, 768, 0, true, false, false, 8.0},
    {"fwd KV-proj  2048x9216x256 +b", 0, 1, 2048, 9216, 256, 0, true, false, false, 87.2},
    {"dX  datt      416x768x768", 1, 1, 416, 768, 768, 0, false, false, false, 8.3},
    {"dX  dres      416x768x2304 +C", 1, 1, 416, 768, 2304, 0, false, true, false, 21.7},
    {"dX  gact    2x256x3072x768 *gelu' ->planes", 1, 2, 256, 3072, 768, 2, false, false, true, 20.5},
    {"dX  gx      2x256x768x3072 +C", 1, 2, 256, 768, 3072, 0, false, true, false, 25.9},
    {"dX  g_enc    2048x256x9216", 1, 1, 2048, 256, 9216, 0, false, false, false, 84.6},
    {"dW  ffn2   24x 768x3072 r256", 2, 24, 768, 3072, 256, 0, false, false, false, 0},
    {"dW  ffn1   24x 3072x768 r256", 2, 24, 3072, 768, 256, 0, false, false, false, 0},
    {"dW  wo     12x 768x768 r416", 2, 12, 768, 768, 416, 0, false, false, false, 0},
    {"dW  wqkv   12x 2304x768 r416", 2, 12, 2304, 768, 416, 0, false, false, false, 0},
    {"dW  wkv     9216x256 r2048", 2, 1, 9216, 256, 2048, 0, false, false, false, 0},
    {"dW  tail   3x 768x768 r208 (B = 4: ragged reduction)", 2, 3, 768, 768, 208, 0, false, false, false, 0},
    // config 5 (3D-LLM shape, d_enc 1408): key / value projection of one cross layer over 40 000 of the 320 000 point tokens
    {"fwd cfg5 K/V  40000x1536x1408 +b", 0, 1, 40000, 1536, 1408, 0, true, false, false, 0},
    {"dX  cfg5 g_enc 40000x1408x1536", 1, 1, 40000, 1408, 1536, 0, false, false, false, 0},
    {"dW  cfg5 wkv  1536x1408 r40000", 2, 1, 1536, 1408, 40000, 0, false, false, false, 0},
};

#ifndef CONFIGS
#define CONFIGS(X)          \
  X(0, 1, 1, 1, 4, 4, 1)    \
  X(1, 1, 1, 2, 4, 4, 1)    \
  X(2, 1, 1, 2, 2, 4, 1)    \
  X(3, 1, 2, 2, 2, 4, 1)    \
  X(4, 2, 2, 2, 2, 3, 1)    \
  X(5, 2, 1, 2, 4, 4, 1)
#endif

static hipError_t run_config(int id, const gemmp::Problem &p, int modes, hipStream_t s) {
  switch (id) {
#define X(ID, MB, NB, WGM, WGN, PF, OCC) case ID: return gemmp::launch<MB, NB, WGM, WGN, PF, OCC>(p, modes, s);
    CONFIGS(X)
#undef X
  }
  return hipErrorInvalidValue;
}
static bool config_dims(int id, int *tm, int *tn) {
  switch (id) {
#define X(ID, MB, NB, WGM, WGN, PF, OCC) case ID: *tm = 32 * MB * WGM; *tn = 32 * NB * WGN; return true;
    CONFIGS(X)
#undef X
  }
  return false;
}
static std::string config_name(int id) {
  char buf[64];
  switch (id) {
#define X(ID, MB, NB, WGM, WGN, PF, OCC) case ID: snprintf(buf, 64, "%d: w%dx%d g%dx%d p%d", ID, 32 * MB, 32 * NB, WGM, WGN, PF); return buf;
    CONFIGS(X)
#undef X
  }
  return "?";
}

static float frand() { return (float)((rand() & 0xffff) / 32768.0 - 1.0); }

static unsigned short bf16_rne(float x) {
  unsigned u;
  memcpy(&u, &x, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
static float bf16_f(unsigned short h) {
  unsigned u = (unsigned)h << 16;
  float x;
  memcpy(&x, &u, 4);
  return x;
}
// chunked planes ([C / 32][R][3][32] bf16 per batch element) of X (R, C): X(r, c) = a(r, c), or a(c, r) when transposed
// (a is then stored (C, R) row-major)
static void make_planes(const std::vector<float> &a, int batch, int R, int C, bool transposed, std::vector<unsigned short> &out) {
  const size_t per = (size_t)R * C;
  out.assign(3 * batch * per, 0);
  for (int b = 0; b < batch; ++b)
    for (int r = 0; r < R; ++r)
      for (int c = 0; c < C; ++c) {
        float x = a[(size_t)b * per + (transposed ? (size_t)c * R + r : (size_t)r * C + c)];
        const size_t o = (size_t)b * per * 3 + ((size_t)(c / 32) * R + r) * 96 + (c % 32);
        for (int pl = 0; pl < 3; ++pl) {
          const unsigned short h = bf16_rne(x);
          out[o + 32 * pl] = h;
          x -= bf16_f(h);
        }
      }
}

int main(int argc, char **argv) {
  const int only_shape = argc > 1 ? atoi(argv[1]) : -1;
  const int only_cfg = argc > 2 ? atoi(argv[2]) : -1;
  hipStream_t stream;
  CHECK(hipStreamCreate(&stream));
  rocblas_handle h;
  rocblas_create_handle(&h);
  rocblas_set_stream(h, stream);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int REPS = 48;

  int si = -1;
  for (const Shape &sh : shapes) {
    ++si;
    if (only_shape >= 0 && si != only_shape) continue;
    const size_t a_el = (size_t)sh.batch * sh.M * sh.K, b_el = (size_t)sh.batch * sh.N * sh.K, c_el = (size_t)sh.batch * sh.M * sh.N;
    // enough weight copies for ~400 MB of planes (forward / input gradient: B is the weight)
    int nbuf = sh.modes == 2 ? 2 : (int)(400e6 / (b_el * 6)) + 1;
    if (nbuf > 48) nbuf = 48;
    if (nbuf < 2) nbuf = 2;
    std::vector<float> hA(a_el), hB(b_el), hbias((size_t)sh.batch * sh.N), hadd(c_el), haux(c_el);
    srand(1234 + si);
    const float wscale = 1.f / sqrtf((float)sh.K);
    for (auto &v : hA) v = frand();
    for (auto &v : hB) v = frand() * wscale * 2.f;
    for (auto &v : hbias) v = frand() * 0.1f;
    for (auto &v : hadd) v = frand();
    for (auto &v : haux) v = frand() * 2.f;
    std::vector<unsigned short> pA, pB;
    if (sh.modes == 2) make_planes(hA, sh.batch, sh.K, sh.M, true, pA); else make_planes(hA, sh.batch, sh.M, sh.K, false, pA);
    if (sh.modes != 0) make_planes(hB, sh.batch, sh.K, sh.N, true, pB); else make_planes(hB, sh.batch, sh.N, sh.K, false, pB);
    unsigned short *dA, *dB, *dCp;
    float *dAf, *dBf, *dC, *dbias, *dadd, *daux;
    CHECK(hipMalloc(&dA, pA.size() * 2));
    CHECK(hipMalloc(&dB, pB.size() * 2 * nbuf));
    CHECK(hipMalloc(&dAf, a_el * 4));
    CHECK(hipMalloc(&dBf, b_el * 4));
    CHECK(hipMalloc(&dC, c_el * 4));
    CHECK(hipMalloc(&dCp, c_el * 6));
    CHECK(hipMalloc(&dbias, hbias.size() * 4));
    CHECK(hipMalloc(&dadd, c_el * 4));
    CHECK(hipMalloc(&daux, c_el * 4));
    CHECK(hipMemcpy(dA, pA.data(), pA.size() * 2, hipMemcpyHostToDevice));
    for (int i = 0; i < nbuf; ++i) CHECK(hipMemcpy(dB + (size_t)i * pB.size(), pB.data(), pB.size() * 2, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dAf, hA.data(), a_el * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dBf, hB.data(), b_el * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dbias, hbias.data(), hbias.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dadd, hadd.data(), c_el * 4, hipMemcpyHostToDevice));

    gemmp::Problem p;
    memset(&p, 0, sizeof(p));
#ifdef GEMMP_TIMING
    static unsigned long long *dbg_always = nullptr;
    if (!dbg_always) CHECK(hipMalloc(&dbg_always, 64 * 8));
    p.dbg = dbg_always;
#endif
    p.A = dA; p.B = dB; p.C = dC;
    p.csA = 96L * (sh.modes == 2 ? sh.K : sh.M); p.csB = 96L * (sh.modes == 0 ? sh.N : sh.K);
    p.extA = (long)sh.M * sh.K * 6; p.extB = (long)sh.N * sh.K * 6;
    p.Cp = sh.planes_out ? dCp : nullptr; p.csC = 96L * sh.M; p.sCp = 3L * sh.M * sh.N;
    static float *d_ws = nullptr;
    static unsigned *d_cnt = nullptr;
    if (!d_ws) { CHECK(hipMalloc(&d_ws, 256u << 20)); CHECK(hipMalloc(&d_cnt, 1 << 20)); CHECK(hipMemset(d_cnt, 0, 1 << 20)); }
    p.ws = d_ws; p.cnt = d_cnt;
    p.bias = sh.bias ? dbias : nullptr;
    p.addend = sh.addend ? dadd : nullptr;
    p.aux = sh.act ? daux : nullptr;
    p.M = sh.M; p.N = sh.N; p.K = sh.K;
    p.ldc = sh.N;
    p.sA = 3L * sh.M * sh.K; p.sB = 3L * sh.N * sh.K; p.sC = (long)sh.M * sh.N; p.sBias = sh.N;
    p.batch = sh.batch; p.act = sh.act;
    const double gflop = 2.0 * sh.batch * sh.M * (double)sh.N * sh.K * 1e-9;

    // ---- rocBLAS default f32 pick on the unsplit operands in the layouts the step has them (row-major x (M, K),
    // W (N, K); weight gradient: dY (K, M), X (K, N)); column-major view: C^T (N, M) = op(B) op(A)
    double rb_us = 0;
    {
      const float one = 1.f, zero = 0.f;
      std::vector<float> tA, tB;
      if (sh.modes == 2) {   // both stored reduction-major
        tA.resize(a_el); tB.resize(b_el);
        for (int b = 0; b < sh.batch; ++b)
          for (int k = 0; k < sh.K; ++k) {
            for (int m = 0; m < sh.M; ++m) tA[(size_t)b * sh.M * sh.K + (size_t)k * sh.M + m] = hA[(size_t)b * sh.M * sh.K + (size_t)m * sh.K + k];
            for (int n = 0; n < sh.N; ++n) tB[(size_t)b * sh.N * sh.K + (size_t)k * sh.N + n] = hB[(size_t)b * sh.N * sh.K + (size_t)n * sh.K + k];
          }
        CHECK(hipMemcpy(dAf, tA.data(), a_el * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(dBf, tB.data(), b_el * 4, hipMemcpyHostToDevice));
      } else if (sh.modes == 1) {
        tB.resize(b_el);
        for (int b = 0; b < sh.batch; ++b)
          for (int k = 0; k < sh.K; ++k)
            for (int n = 0; n < sh.N; ++n) tB[(size_t)b * sh.N * sh.K + (size_t)k * sh.N + n] = hB[(size_t)b * sh.N * sh.K + (size_t)n * sh.K + k];
        CHECK(hipMemcpy(dBf, tB.data(), b_el * 4, hipMemcpyHostToDevice));
      }
      auto call = [&]() {
        if (sh.modes == 0)
          rocblas_sgemm_strided_batched(h, rocblas_operation_transpose, rocblas_operation_none, sh.N, sh.M, sh.K, &one, dBf,
                                        sh.K, (rocblas_stride)sh.N * sh.K, dAf, sh.K, (rocblas_stride)sh.M * sh.K, &zero, dC,
                                        sh.N, (rocblas_stride)sh.M * sh.N, sh.batch);
        else if (sh.modes == 1)
          rocblas_sgemm_strided_batched(h, rocblas_operation_none, rocblas_operation_none, sh.N, sh.M, sh.K, &one, dBf, sh.N,
                                        (rocblas_stride)sh.N * sh.K, dAf, sh.K, (rocblas_stride)sh.M * sh.K, &zero, dC, sh.N,
                                        (rocblas_stride)sh.M * sh.N, sh.batch);
        else
          rocblas_sgemm_strided_batched(h, rocblas_operation_none, rocblas_operation_transpose, sh.N, sh.M, sh.K, &one, dBf, sh.N,
                                        (rocblas_stride)sh.N * sh.K, dAf, sh.M, (rocblas_stride)sh.M * sh.K, &zero, dC, sh.N,
                                        (rocblas_stride)sh.M * sh.N, sh.batch);
      };
      for (int i = 0; i < 4; ++i) call();
      CHECK(hipStreamSynchronize(stream));
      CHECK(hipEventRecord(e0, stream));
      for (int i = 0; i < REPS; ++i) call();
      CHECK(hipEventRecord(e1, stream));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      rb_us = ms * 1e3 / REPS;
    }
    printf("\n== [%d] %s  (%.2f GFLOP; tuned library in the step %.1f us; rocBLAS default here, hot operands, %.1f us = %.0f TF)\n", si,
           sh.name, gflop, sh.lib_us, rb_us, gflop / rb_us * 1e3);

    for (int id = 0; id < 32; ++id) {
      int tm, tn;
      if (!config_dims(id, &tm, &tn)) continue;
      if (only_cfg >= 0 && id != only_cfg) continue;
      const int tiles = ((sh.M + tm - 1) / tm) * ((sh.N + tn - 1) / tn) * sh.batch;
      for (int splits = 1; splits <= 8; ++splits) {
      const int wgs = tiles * splits;
      if (splits > 1 && (wgs > 1100 || sh.K / 32 / splits < 3)) break;
      if (splits > 1 && wgs < 100) continue;
      if ((size_t)wgs * tm * tn * 4 > (256u << 20)) break;
      p.splits = splits;
      CHECK(hipMemsetAsync(dC, 0xff, c_el * 4, stream));
      CHECK(hipMemsetAsync(dCp, 0xff, c_el * 6, stream));
      if (sh.act) CHECK(hipMemcpyAsync(daux, haux.data(), c_el * 4, hipMemcpyHostToDevice, stream));
      p.B = dB;
      (void)hipGetLastError();   // a failed attribute call of the configuration before leaves its error behind
      hipError_t e = run_config(id, p, sh.modes, stream);
      if (e != hipSuccess) { printf("   %-20s launch error %s\n", config_name(id).c_str(), hipGetErrorString(e)); continue; }
      CHECK(hipStreamSynchronize(stream));
      std::vector<float> hC(c_el), hAuxOut;
      std::vector<unsigned short> hCp;
      CHECK(hipMemcpy(hC.data(), dC, c_el * 4, hipMemcpyDeviceToHost));
      if (sh.act == 1) { hAuxOut.resize(c_el); CHECK(hipMemcpy(hAuxOut.data(), daux, c_el * 4, hipMemcpyDeviceToHost)); }
      if (sh.planes_out) { hCp.resize(3 * c_el); CHECK(hipMemcpy(hCp.data(), dCp, c_el * 6, hipMemcpyDeviceToHost)); }
      double worst = 0, worst_planes = 0;
      for (int t = 0; t < 600; ++t) {
        const int b = rand() % sh.batch;
        int m = rand() % sh.M, n = rand() % sh.N;
        if (t < 8) { m = sh.M - 1 - (t & 1); n = sh.N - 1 - (t >> 1); }
        if (t >= 8 && t < 16) { m = (t & 1) * 31; n = ((t >> 1) & 3) * 17; }
        double ref = 0;
        for (int k = 0; k < sh.K; ++k)
          ref += (double)hA[((size_t)b * sh.M + m) * sh.K + k] * (double)hB[((size_t)b * sh.N + n) * sh.K + k];
        if (sh.bias) ref += hbias[(size_t)b * sh.N + n];
        const size_t ci = ((size_t)b * sh.M + m) * sh.N + n;
        if (sh.act == 1) {
          const double pre = ref;
          worst = fmax(worst, fabs(hAuxOut[ci] - pre) / (1 + fabs(pre)));
          ref = 0.5 * pre * (1 + erf(pre * 0.70710678118654752440));
        } else if (sh.act == 2) {
          const double u = haux[ci];
          ref *= 0.5 * (1 + erf(u * 0.70710678118654752440)) + u * 0.39894228040143267794 * exp(-0.5 * u * u);
        }
        if (sh.addend) ref += hadd[ci];
        worst = fmax(worst, fabs(hC[ci] - ref) / (1 + fabs(ref)));
        if (sh.planes_out) {   // the planes must add up to the f32 result exactly
          const size_t pi = (size_t)b * 3 * sh.M * sh.N + ((size_t)(n / 32) * sh.M + m) * 96 + (n % 32);
          const float s = (bf16_f(hCp[pi]) + bf16_f(hCp[pi + 32])) + bf16_f(hCp[pi + 64]);
          worst_planes = fmax(worst_planes, fabs((double)s - (double)hC[ci]));
        }
      }
      for (int i = 0; i < 3; ++i) { p.B = dB + (size_t)(i % nbuf) * pB.size(); run_config(id, p, sh.modes, stream); }
      CHECK(hipStreamSynchronize(stream));
      CHECK(hipEventRecord(e0, stream));
      for (int i = 0; i < REPS; ++i) { p.B = dB + (size_t)((i + 3) % nbuf) * pB.size(); run_config(id, p, sh.modes, stream); }
      CHECK(hipEventRecord(e1, stream));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      const double us = ms * 1e3 / REPS;
#ifdef GEMMP_TIMING
      {
        unsigned long long *dbg, hh[60];
        CHECK(hipMalloc(&dbg, 60 * 8));
        CHECK(hipMemset(dbg, 0, 60 * 8));
        p.dbg = dbg; p.B = dB;
        run_config(id, p, sh.modes, stream);
        CHECK(hipStreamSynchronize(stream));
        CHECK(hipMemcpy(hh, dbg, 60 * 8, hipMemcpyDeviceToHost));
        printf("      stamps (cycles, wg 0 / wave 0):");
        for (int i = 1; i < 60 && hh[i]; ++i) printf(" %llu", hh[i] - hh[i - 1]);
        printf("\n");
        CHECK(hipFree(dbg));
        p.dbg = dbg_always;
      }
#endif
      printf("   %-20s x%d wgs %4d  %7.2f us  %6.1f TF  err %.1e%s", config_name(id).c_str(), splits, wgs, us, gflop / us * 1e3, worst,
             worst > 2e-5 ? "  <-- WRONG" : "");
      if (sh.planes_out) printf("  planes-sum err %.1e%s", worst_planes, worst_planes > 0 ? "  <-- WRONG" : "");
      printf("\n");
      fflush(stdout);
      }
    }
    CHECK(hipFree(dA)); CHECK(hipFree(dB)); CHECK(hipFree(dAf)); CHECK(hipFree(dBf)); CHECK(hipFree(dC)); CHECK(hipFree(dCp));
    CHECK(hipFree(dbias)); CHECK(hipFree(dadd)); CHECK(hipFree(daux));
  }
  return 0;
}
