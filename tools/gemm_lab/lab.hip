// lab.hip — standalone correctness + timing harness for the LDS-DMA GEMM family (gemm_dma.h) on MI355X.
//   make -C tools/gemm_lab && gpurun -- tools/gemm_lab/lab.bin [name filter]      (LAB_DIAG=1: in-kernel clock stamps)
// Checks every configuration against an fp64 host reference on ragged shapes, then times it on the config-1
// shapes with operands rotated through > 256 MiB (no cache-resident replays), next to the library's r01 kernels.
#include "../../esc-gnn_amd/csrc/gemm_dma.h"
#include <vector>
#include <string>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <functional>

namespace esc {
bool prof_slot(int, hipEvent_t*, hipEvent_t*) { return false; }
}
using namespace esc;
using namespace esc::dma;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static uint32_t rng_state = 12345u;
static float frand() { rng_state = rng_state * 1664525u + 1013904223u; return ((rng_state >> 8) & 0xFFFF) / 32768.0f - 1.0f; }

struct Dev {
  float* p = nullptr; size_t n = 0;
  explicit Dev(size_t n_) : n(n_) { CK(hipMalloc(&p, (n ? n : 1) * 4)); }
  ~Dev() { (void)hipFree(p); }
  void upload(const std::vector<float>& h) { CK(hipMemcpy(p, h.data(), h.size() * 4, hipMemcpyHostToDevice)); }
  std::vector<float> download(size_t cnt) const { std::vector<float> h(cnt); CK(hipMemcpy(h.data(), p, cnt * 4, hipMemcpyDeviceToHost)); return h; }
};

enum Kind { NT = 0, NN = 1, TN = 2 };
typedef std::function<void(const GArgs&, hipStream_t)> Launch;
struct Variant { std::string name; Kind kind; Launch fn; int pro; bool stats; int bk; };

template <int BM, int BN, int BK, int WM, int WN, int ST, int LW, int PRO, bool STATS>
static Variant nt(const char* name) {
  return Variant{name, NT, [](const GArgs& g, hipStream_t s) { CK((launch_gemm<BM, BN, BK, WM, WN, ST, LW, false, false, PRO, STATS, false>(g, 0, s))); }, PRO, STATS, BK};
}
template <int BM, int BN, int BK, int WM, int WN, int ST, int LW>
static Variant nn(const char* name) {
  return Variant{name, NN, [](const GArgs& g, hipStream_t s) { CK((launch_gemm<BM, BN, BK, WM, WN, ST, LW, false, true, 0, false, false>(g, 0, s))); }, 0, false, BK};
}
template <int BM, int BN, int BK, int WM, int WN, int ST, int LW, int PRO>
static Variant tn(const char* name) {
  return Variant{name, TN, [](const GArgs& g, hipStream_t s) { CK((launch_gemm<BM, BN, BK, WM, WN, ST, LW, true, true, PRO, false, true>(g, 0, s))); }, PRO, false, BK};
}

// generic host reference: C[m][n] = sum_r a(m,r) b(n,r)
static bool check(const Variant& v, int M, int N, int R, int lda, int ldb, int ldc, bool accumulate, int per_split) {
  const bool a_rm = v.kind == TN, b_rm = v.kind != NT;
  const size_t a_n = a_rm ? (size_t)R * lda : (size_t)M * lda, b_n = b_rm ? (size_t)R * ldb : (size_t)N * ldb;
  const int splits = per_split > 0 ? (R + per_split - 1) / per_split : 1;
  std::vector<float> hA(a_n), hB(b_n), hb(N), hC((size_t)splits * M * ldc), hs(1280), hh(1280);
  for (auto& x : hA) x = frand();
  for (auto& x : hB) x = frand();
  for (auto& x : hb) x = frand();
  for (auto& x : hC) x = frand();
  for (auto& x : hs) x = 0.5f + 0.5f * frand();
  for (auto& x : hh) x = 0.3f * frand();
  Dev dA(a_n), dB(b_n), db(N), dC(hC.size() + 64), ds(1280), dh(1280), dst((size_t)(M / 32 + 1) * N * 2), ddb((size_t)splits * M);
  dA.upload(hA); dB.upload(hB); db.upload(hb); dC.upload(hC); ds.upload(hs); dh.upload(hh);
  CK(hipMemset(dC.p + hC.size(), 0x7f, 64 * 4));
  GArgs g{};
  g.A = dA.p; g.lda = lda; g.B = dB.p; g.ldb = ldb; g.C = dC.p; g.ldc = ldc; g.bias = v.kind == NT ? db.p : nullptr;
  g.pro_scale = v.pro ? ds.p : nullptr; g.pro_shift = v.pro ? dh.p : nullptr;
  g.col_stats = v.stats ? reinterpret_cast<float2*>(dst.p) : nullptr;
  g.db_part = ddb.p;
  g.M = M; g.N = N; g.R = R; g.red_per_split = per_split; g.accumulate = accumulate;
  v.fn(g, nullptr);
  CK(hipDeviceSynchronize());
  std::vector<float> out = dC.download(hC.size() + 64);
  auto a_at = [&](int m, int r) -> double {
    double a = a_rm ? hA[(size_t)r * lda + m] : hA[(size_t)m * lda + r];
    if (v.pro == 1) { a = a * hs[r] + hh[r]; if (a < 0) a = 0; }
    return a;
  };
  auto b_at = [&](int n, int r) -> double {
    double b = b_rm ? hB[(size_t)r * ldb + n] : hB[(size_t)n * ldb + r];
    if (v.pro == 2) { b = b * hs[n] + hh[n]; if (b < 0) b = 0; }
    return b;
  };
  double maxerr = 0, maxref = 0;
  std::vector<double> ref((size_t)M * N);
  bool ok = true;
  for (int sp = 0; sp < splits; ++sp) {
    const int r0 = per_split > 0 ? sp * per_split : 0, r1 = per_split > 0 ? std::min(R, r0 + per_split) : R;
    for (int m = 0; m < M; ++m)
      for (int n = 0; n < N; ++n) {
        double s = v.kind == NT ? hb[n] : 0.0;
        for (int r = r0; r < r1; ++r) s += a_at(m, r) * b_at(n, r);
        if (accumulate) s += hC[((size_t)sp * M + m) * ldc + n];
        ref[(size_t)m * N + n] = s;
        maxerr = std::fmax(maxerr, std::fabs(s - out[((size_t)sp * M + m) * ldc + n]));
        maxref = std::fmax(maxref, std::fabs(s));
      }
    for (int m = 0; m < M && ok; ++m)
      for (int n = N; n < ldc; ++n)
        if (out[((size_t)sp * M + m) * ldc + n] != hC[((size_t)sp * M + m) * ldc + n]) { ok = false; printf("   pad column touched at (%d,%d)\n", m, n); break; }
  }
  ok = ok && maxerr <= 2e-5 * std::fmax(1.0, maxref);
  for (int i = 0; i < 64; ++i) { uint32_t u; memcpy(&u, &out[hC.size() + i], 4); if (u != 0x7f7f7f7fu) { ok = false; printf("   guard word %d overwritten\n", i); break; } }
  double serr = 0;
  if (v.stats && !accumulate) {
    std::vector<float> st = dst.download((size_t)(M / 32 + 1) * N * 2);
    for (int blk = 0; blk * 32 < M; ++blk)
      for (int n = 0; n < N; ++n) {
        const int r0 = blk * 32, r1 = std::min(M, r0 + 32);
        double mu = 0, m2 = 0;
        for (int m = r0; m < r1; ++m) mu += ref[(size_t)m * N + n];
        mu /= (r1 - r0);
        for (int m = r0; m < r1; ++m) { double d = ref[(size_t)m * N + n] - mu; m2 += d * d; }
        serr = std::fmax(serr, std::fabs(mu - st[((size_t)blk * N + n) * 2]));
        serr = std::fmax(serr, std::fabs(m2 - st[((size_t)blk * N + n) * 2 + 1]) / std::fmax(1.0, m2));
      }
    if (serr > 1e-4) ok = false;
  }
  double dberr = 0;
  if (v.kind == TN) {     // bias gradient partials: per split sums of A's columns
    std::vector<float> dbp = ddb.download((size_t)splits * M);
    for (int sp = 0; sp < splits; ++sp) {
      const int r0 = per_split > 0 ? sp * per_split : 0, r1 = per_split > 0 ? std::min(R, r0 + per_split) : R;
      for (int m = 0; m < M; ++m) {
        double s = 0;
        for (int r = r0; r < r1; ++r) s += hA[(size_t)r * lda + m];
        dberr = std::fmax(dberr, std::fabs(s - dbp[(size_t)sp * M + m]));
      }
    }
    if (dberr > 1e-4) ok = false;
  }
  printf("  check %-34s M=%d N=%d R=%d lda=%d ldb=%d ldc=%d acc=%d per=%d : maxerr %.3g (ref %.3g) stats %.3g db %.3g %s\n", v.name.c_str(), M, N, R,
         lda, ldb, ldc, (int)accumulate, per_split, maxerr, maxref, serr, dberr, ok ? "OK" : "FAIL");
  return ok;
}

extern "C" int esc_linear_fwd(const float* X, int64_t ld_x, const float* W, int64_t ld_w, const float* bias,
                              const float* in_scale, const float* in_shift, int64_t M, int64_t N, int64_t K,
                              float* Y, int64_t ld_y, float* col_stats, void* stream);
extern "C" int esc_linear_bwd_input(const float* dY, int64_t ld_dy, const float* W, int64_t ld_w, int64_t M,
                                    int64_t N, int64_t K, float* dX, int64_t ld_dx, int accumulate, void* stream);
extern "C" int esc_linear_bwd_weight(const float* dY, int64_t ld_dy, const float* X, int64_t ld_x,
                                     const float* in_scale, const float* in_shift, int64_t M, int64_t N,
                                     int64_t K, float* dW, int64_t ld_dw, float* db, float* slabs, void* stream);
extern "C" int64_t esc_linear_bwd_weight_scratch(int64_t M, int64_t N, int64_t K);

static double time_us(const std::function<void(int)>& fn, int reps = 40) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 5; ++i) fn(i);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a, nullptr));
  for (int i = 0; i < reps; ++i) fn(i);
  CK(hipEventRecord(b, nullptr));
  CK(hipEventSynchronize(b));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
  return ms * 1e3 / reps;
}

static void diag(const Variant& v, GArgs a) {
  const int maxwg = 1 << 16;
  unsigned long long* st;
  CK(hipMalloc(&st, (size_t)maxwg * 12 * 8));
  CK(hipMemset(st, 0, (size_t)maxwg * 12 * 8));
  a.stamps = st;
  for (int rep = 0; rep < 3; ++rep) v.fn(a, nullptr);
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> h((size_t)maxwg * 12);
  CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
  double seg[3] = {0, 0, 0}, clk = 0; int n = 0;
  unsigned long long rmin = ~0ull, rmax = 0;
  for (int w = 0; w < maxwg; ++w) {
    const unsigned long long* q = &h[(size_t)w * 12];
    if (q[0] == 0) continue;
    for (int i = 0; i < 3; ++i) seg[i] += (double)(q[i + 1] - q[i]);
    const double dr = (double)(q[9] - q[6]);
    if (dr > 0) clk += (double)(q[3] - q[0]) / dr * 100.0;      // MHz
    rmin = std::min(rmin, q[6]); rmax = std::max(rmax, q[9]);
    ++n;
  }
  if (n) printf("   | %d WGs: pro %.0f loop %.0f epi %.0f cyc, %.0f MHz, span %.2f us", n, seg[0] / n, seg[1] / n, seg[2] / n, clk / n,
                (double)(rmax - rmin) / 100.0);
  CK(hipFree(st));
}

__global__ void null_kernel(float* p) { if (p && threadIdx.x == 9999) p[0] = 1.f; }

int main(int argc, char** argv) {
  const char* filter = argc > 1 ? argv[1] : "";
  const bool want_diag = getenv("LAB_DIAG") != nullptr;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("device %s, %d CUs\n", prop.gcnArchName, prop.multiProcessorCount);

  std::vector<Variant> vs;
  vs.push_back(nt<128, 128, 32, 2, 2, 3, 4, 0, false>("NT 128x128x32 s3 L4"));
  vs.push_back(nt<128, 128, 32, 2, 2, 3, 4, 1, true>("NT 128x128x32 s3 L4 PRO+STATS"));
  vs.push_back(nt<128, 64, 32, 2, 2, 3, 2, 0, false>("NT 128x64x32 s3 L2"));
  vs.push_back(nt<64, 64, 32, 2, 2, 3, 2, 0, false>("NT 64x64x32 s3 L2"));
  vs.push_back(nt<64, 64, 32, 2, 2, 3, 2, 1, true>("NT 64x64x32 s3 L2 PRO+STATS"));
  vs.push_back(nt<64, 64, 64, 2, 2, 3, 2, 0, false>("NT 64x64x64 s3 L2"));
  vs.push_back(nt<64, 64, 64, 2, 2, 3, 4, 0, false>("NT 64x64x64 s3 L4"));
  vs.push_back(nt<64, 32, 64, 2, 1, 3, 2, 0, false>("NT 64x32x64 w2x1 s3 L2"));
  vs.push_back(nt<32, 64, 64, 1, 2, 3, 2, 0, false>("NT 32x64x64 w1x2 s3 L2"));
  vs.push_back(nn<128, 128, 32, 2, 2, 3, 4>("NN 128x128x32 s3 L4"));
  vs.push_back(nn<128, 64, 32, 2, 2, 3, 2>("NN 128x64x32 s3 L2"));
  vs.push_back(nn<64, 64, 32, 2, 2, 3, 2>("NN 64x64x32 s3 L2"));
  vs.push_back(nn<64, 64, 64, 2, 2, 3, 2>("NN 64x64x64 s3 L2"));
  vs.push_back(tn<128, 128, 32, 2, 2, 3, 4, 0>("TN 128x128x32 s3 L4"));
  vs.push_back(tn<128, 128, 32, 2, 2, 3, 4, 2>("TN 128x128x32 s3 L4 PRO"));
  vs.push_back(tn<64, 64, 32, 2, 2, 3, 2, 0>("TN 64x64x32 s3 L2"));
  vs.push_back(tn<64, 64, 64, 2, 2, 3, 2, 0>("TN 64x64x64 s3 L2"));
  vs.push_back(tn<64, 64, 32, 2, 2, 3, 2, 2>("TN 64x64x32 s3 L2 PRO"));

  bool all_ok = true;
  printf("== correctness (fp64 host reference)\n");
  for (auto& v : vs) {
    if (strstr(v.name.c_str(), filter) == nullptr) continue;
    if (v.kind == NT) {
      all_ok &= check(v, 333, 256, 256, 256, 256, 256, false, 0);
      all_ok &= check(v, 200, 138, 128, 160, 128, 150, true, 0);     // ragged M and N, lda > K, ldc > N, accumulate
      all_ok &= check(v, 97, 64, 320, 320, 320, 64, false, 0);
    } else if (v.kind == NN) {
      all_ok &= check(v, 333, 256, 256, 256, 256, 256, false, 0);
      all_ok &= check(v, 200, 136, 128, 160, 140, 152, true, 0);
      all_ok &= check(v, 97, 320, 64, 64, 320, 320, false, 0);
    } else {
      all_ok &= check(v, 256, 256, 1000, 256, 256, 256, false, 256);    // split over the reduction, ragged last split
      all_ok &= check(v, 136, 200, 333, 140, 204, 200, false, 0);
      all_ok &= check(v, 64, 320, 500, 64, 320, 320, false, 192);
    }
  }
  printf("== all checks %s\n", all_ok ? "PASSED" : "FAILED");

  {
    double t = time_us([&](int) { hipLaunchKernelGGL(null_kernel, dim3(256), dim3(256), 0, nullptr, (float*)nullptr); }, 200);
    printf("null kernel back-to-back: %.2f us/launch\n", t);
  }
  struct Shape { int M, N, K; const char* what; };
  const Shape shapes[] = {{15200, 256, 256, "edge rows"}, {2400, 256, 256, "node rows"}, {2400, 256, 1280, "lin1"}};
  for (const Shape& sh : shapes) {
    const int M = sh.M, N = sh.N, K = sh.K;            // Linear(K -> N) on M rows
    const size_t x_n = (size_t)M * K, y_n = (size_t)M * N;
    int nbuf = (int)((400u << 20) / ((x_n + y_n) * 4)) + 1;
    if (nbuf > 64) nbuf = 64;
    if (nbuf < 2) nbuf = 2;
    Dev dX(x_n * nbuf), dY(y_n * nbuf), dW((size_t)N * K), db(N), ds(1280), dh(1280), dst((size_t)(M / 32 + 1) * N * 2);
    Dev dSl((size_t)esc_linear_bwd_weight_scratch(M, N, K) + 256 * (size_t)N * K / 64 + (size_t)N * 256), dDw((size_t)N * K), dDb(N);
    {
      std::vector<float> h(x_n);
      for (auto& x : h) x = frand();
      for (int i = 0; i < nbuf; ++i) CK(hipMemcpy(dX.p + x_n * i, h.data(), x_n * 4, hipMemcpyHostToDevice));
      std::vector<float> hy(y_n);
      for (auto& x : hy) x = frand();
      for (int i = 0; i < nbuf; ++i) CK(hipMemcpy(dY.p + y_n * i, hy.data(), y_n * 4, hipMemcpyHostToDevice));
      std::vector<float> w((size_t)N * K);
      for (auto& x : w) x = frand();
      dW.upload(w);
      std::vector<float> b(N, 0.1f), s1(1280, 0.9f), s2(1280, 0.05f);
      db.upload(b); ds.upload(s1); dh.upload(s2);
    }
    const double flop = 2.0 * M * N * K;
    printf("== %s  Linear(%d -> %d) on %d rows  (%.2f GFLOP per GEMM, %d rotating buffers)\n", sh.what, K, N, M, flop * 1e-9, nbuf);
    {
      double t = time_us([&](int i) { const int q = i % nbuf; esc_linear_fwd(dX.p + x_n * q, K, dW.p, K, db.p, nullptr, nullptr, M, N, K, dY.p + y_n * q, N, nullptr, nullptr); });
      printf("  %-36s %8.2f us  %6.1f TFLOP/s\n", "r01 esc_linear_fwd", t, flop / t * 1e-6);
      t = time_us([&](int i) { const int q = i % nbuf; esc_linear_bwd_input(dY.p + y_n * q, N, dW.p, K, M, N, K, dX.p + x_n * q, K, 0, nullptr); });
      printf("  %-36s %8.2f us  %6.1f TFLOP/s\n", "r01 esc_linear_bwd_input", t, flop / t * 1e-6);
      t = time_us([&](int i) { const int q = i % nbuf; esc_linear_bwd_weight(dY.p + y_n * q, N, dX.p + x_n * q, K, nullptr, nullptr, M, N, K, dDw.p, K, dDb.p, dSl.p, nullptr); });
      printf("  %-36s %8.2f us  %6.1f TFLOP/s (incl. slab reduce)\n", "r01 esc_linear_bwd_weight", t, flop / t * 1e-6);
    }
    for (auto& v : vs) {
      if (strstr(v.name.c_str(), filter) == nullptr) continue;
      GArgs g{};
      g.bias = nullptr; g.accumulate = 0;
      g.pro_scale = v.pro ? ds.p : nullptr; g.pro_shift = v.pro ? dh.p : nullptr;
      g.col_stats = v.stats ? reinterpret_cast<float2*>(dst.p) : nullptr;
      std::function<void(GArgs&, int)> bind;
      char extra[64] = "";
      if (v.kind == NT) {
        g.lda = K; g.B = dW.p; g.ldb = K; g.ldc = N; g.bias = db.p; g.M = M; g.N = N; g.R = K;
        bind = [&](GArgs& a, int q) { a.A = dX.p + x_n * q; a.C = dY.p + y_n * q; };
      } else if (v.kind == NN) {      // dX[M,K] = dY[M,N] W[N,K]
        g.lda = N; g.B = dW.p; g.ldb = K; g.ldc = K; g.M = M; g.N = K; g.R = N;
        bind = [&](GArgs& a, int q) { a.A = dY.p + y_n * q; a.C = dX.p + x_n * q; };
      } else {                        // dW[N,K] = dY^T X, split over M into ~256 workgroups
        g.lda = N; g.ldb = K; g.ldc = K; g.M = N; g.N = K; g.R = M; g.C = dSl.p;
        const int bm = strstr(v.name.c_str(), "128x128") ? 128 : 64;
        const int tiles = ((N + bm - 1) / bm) * ((K + bm - 1) / bm);
        int splits = std::max(1, 256 / tiles);
        int per = ((M + splits - 1) / splits + v.bk - 1) / v.bk * v.bk;
        if (per < 128) per = 128;
        splits = (M + per - 1) / per;
        g.red_per_split = per;
        g.db_part = dSl.p + (size_t)splits * N * K;
        snprintf(extra, sizeof extra, " [%d splits of %d rows]", splits, per);
        bind = [&](GArgs& a, int q) { a.A = dY.p + y_n * q; a.B = dX.p + x_n * q; };
      }
      double t = time_us([&](int i) { GArgs a = g; bind(a, i % nbuf); v.fn(a, nullptr); });
      printf("  %-36s %8.2f us  %6.1f TFLOP/s%s", v.name.c_str(), t, flop / t * 1e-6, extra);
      if (want_diag) { GArgs a = g; bind(a, 0); diag(v, a); }
      printf("\n");
    }
  }
  return all_ok ? 0 : 1;
}
