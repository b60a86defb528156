// Stand-alone check + timing harness for the planes pipeline kernels (tuning aid):
//   pack(A fp32) -> GEMM(planes -> planes, W1, LeakyReLU) -> GEMM(planes -> fp32, W2), compared with a double-precision
//   host computation on sampled rows; then each GEMM variant timed at the given shape.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off [-DUSF_STAMP] tools/exp_planes.hip -o tools/exp_planes_x
//   tools/exp_planes_x [M] [k-blocks] [output blocks]
#include "../usflows_amd/csrc/usf_planes.hip"
#include <stdarg.h>
#include <string.h>
#include <algorithm>
#include <cmath>
#include <vector>
namespace usf { void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fprintf(stderr, "\n"); } }
static unsigned short bf16_rn(float x) { unsigned u; memcpy(&u, &x, 4); u += 0x7fff + ((u >> 16) & 1); return (unsigned short)(u >> 16); }
static float bf16_f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }
static int NPL = 3;     // 3: bf16x3 planes, 2: fp16x2 planes (argv[4])
static unsigned short f16_bits(_Float16 h) { unsigned short u; memcpy(&u, &h, 2); return u; }
// planes image [NPL][rows][32*nk] of logical W [rows][32*nk] with the slot permutation on K
static std::vector<unsigned short> planes_of(const std::vector<float>& W, int64_t rows, int64_t nk) {
  const int64_t K = 32 * nk;
  std::vector<unsigned short> P(NPL * rows * K, 0);
  for (int64_t n = 0; n < rows; ++n) for (int64_t kb = 0; kb < nk; ++kb) for (int s = 0; s < 32; ++s) {
    float x = W[n * K + 32 * kb + usf::plane_feature_of_slot(s)];
    if (NPL == 3) {
      unsigned short h = bf16_rn(x); float r = x - bf16_f(h); unsigned short m = bf16_rn(r); float r2 = r - bf16_f(m);
      P[(0 * rows + n) * K + 32 * kb + s] = h; P[(1 * rows + n) * K + 32 * kb + s] = m; P[(2 * rows + n) * K + 32 * kb + s] = bf16_rn(r2);
    } else {
      _Float16 h = (_Float16)x; float r = x - (float)h; _Float16 m = (_Float16)r;
      P[(0 * rows + n) * K + 32 * kb + s] = f16_bits(h); P[(1 * rows + n) * K + 32 * kb + s] = f16_bits(m);
    }
  }
  return P;
}
int main(int argc, char** argv) {
  const int64_t M = argc > 1 ? atoll(argv[1]) : 65536, nkb = argc > 2 ? atoll(argv[2]) : 25, nout = argc > 3 ? atoll(argv[3]) : 25;
  if (argc > 4) NPL = atoi(argv[4]);
  const int fmt = NPL == 2 ? USF_PLANES_F16X2 : USF_PLANES_BF16X3;
  const int64_t K = 32 * nkb, N = 32 * nout, np = (M + 15) / 16;
  std::vector<float> hA(M * K), hW1(N * K), hW2(N * N), hb(N);
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
  for (auto& v : hA) v = rnd() * 4.f;
  for (auto& v : hW1) v = rnd() * 0.1f;
  for (auto& v : hW2) v = rnd() * 0.1f;
  for (auto& v : hb) v = rnd();
  auto P1 = planes_of(hW1, N, nkb), P2 = planes_of(hW2, N, nout);
  float *A, *bias, *C; void *pA, *pB, *W1, *W2; int32_t* idx;
  hipMalloc(&A, M * K * 4); hipMalloc(&bias, N * 4); hipMalloc(&C, M * N * 4);
  hipMalloc(&pA, np * nkb * NPL * 1024); hipMalloc(&pB, np * nout * NPL * 1024); hipMalloc(&W1, P1.size() * 2); hipMalloc(&W2, P2.size() * 2);
  hipMalloc(&idx, K * 4);
  std::vector<int32_t> hidx(K); for (int64_t i = 0; i < K; ++i) hidx[i] = (int32_t)i;
  hipMemcpy(A, hA.data(), M * K * 4, hipMemcpyHostToDevice); hipMemcpy(bias, hb.data(), N * 4, hipMemcpyHostToDevice);
  hipMemcpy(W1, P1.data(), P1.size() * 2, hipMemcpyHostToDevice); hipMemcpy(W2, P2.data(), P2.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(idx, hidx.data(), K * 4, hipMemcpyHostToDevice);
  usf_pack_planes_desc pd = {}; pd.src = A; pd.ld = K; pd.M = M; pd.nkb = nkb; pd.idx = idx; pd.planes = pA; pd.format = fmt;
  usf_gemm_planes_desc g1 = {}; g1.A = pA; g1.a_nkb = nkb; g1.nk = nkb; g1.W_planes = W1; g1.ldw = K; g1.w_plane_stride = N * K; g1.w_rows = N;
  g1.bias = bias; g1.C_planes = pB; g1.c_nkb = nout; g1.c_kbn = nout; g1.M = M; g1.res_sign = 1.f; g1.act = USF_ACT_LEAKY_RELU; g1.slope = 0.01f; g1.format = fmt;
  usf_gemm_planes_desc g2 = {}; g2.A = pB; g2.a_nkb = nout; g2.nk = nout; g2.W_planes = W2; g2.ldw = N; g2.w_plane_stride = N * N; g2.w_rows = N;
  g2.C_f32 = C; g2.ldc = N; g2.N = N; g2.M = M; g2.res_sign = 1.f; g2.format = fmt;
#ifdef USF_STAMP
  unsigned long long* dbg; hipMalloc(&dbg, 2 * 16384 * 8 * 8); hipMemset(dbg, 0, 2 * 16384 * 8 * 8); usf::g_pdbg = dbg;
#endif
  if (usf::pack_planes(&pd, 0) || usf::gemm_planes(&g1, 0) || usf::gemm_planes(&g2, 0)) return 1;
  if (hipDeviceSynchronize() != hipSuccess) { printf("kernel fault\n"); return 2; }
  // ---- check sampled rows ----
  std::vector<float> hC(M * N); hipMemcpy(hC.data(), C, M * N * 4, hipMemcpyDeviceToHost);
  double maxerr = 0, maxref = 0;
  const int64_t rows[] = {0, 1, 15, 16, 17, 255, 256, M / 2 - 1, M / 2, M - 257, M - 17, M - 1};
  for (int64_t m : rows) {
    if (m < 0 || m >= M) continue;
    std::vector<double> h(N);
    for (int64_t n = 0; n < N; ++n) { double a = hb[n]; for (int64_t k = 0; k < K; ++k) a += (double)hA[m * K + k] * hW1[n * K + k]; h[n] = a > 0 ? a : a * 0.01; }
    for (int64_t n = 0; n < N; ++n) { double a = 0; for (int64_t k = 0; k < N; ++k) a += h[k] * hW2[n * N + k];
      maxerr = fmax(maxerr, fabs(a - hC[m * N + n])); maxref = fmax(maxref, fabs(a)); }
  }
  printf("[%s] check: max abs err %.3e vs max |ref| %.3e -> rel %.2e %s\n", NPL == 2 ? "f16x2" : "bf16x3", maxerr, maxref, maxerr / maxref, maxerr / maxref < 5e-6 ? "OK" : "FAIL");
  // ---- timing ----
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto timeit = [&](const char* name, auto fn, double flops) {
    for (int i = 0; i < 3; ++i) fn();
    hipDeviceSynchronize();
    const int iters = 20;
    hipEventRecord(e0, 0); for (int i = 0; i < iters; ++i) fn(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= iters;
    printf("%-28s M=%lld N=%lld K=%lld: %.3f ms  %.1f TF/s (fp32-equivalent)\n", name, (long long)M, (long long)N, (long long)K, ms, flops / ms / 1e9);
  };
  timeit("pack", [&]() { usf::pack_planes(&pd, 0); }, 0.0);
  timeit("gemm planes->planes (act)", [&]() { usf::gemm_planes(&g1, 0); }, 2.0 * M * N * K);
  timeit("gemm planes->fp32", [&]() { usf::gemm_planes(&g2, 0); }, 2.0 * M * N * N);
  usf_gemm_planes_desc g3 = g1; g3.residual = pB; g3.act = USF_ACT_NONE;      // in-place residual (MaskedCoupling's last layer)
  timeit("gemm planes->planes (+res)", [&]() { usf::gemm_planes(&g3, 0); }, 2.0 * M * N * K);
#ifdef USF_STAMP
  unsigned long long* span; hipMalloc(&span, 64 * 16);
  { std::vector<unsigned long long> init(128); for (int i = 0; i < 64; ++i) { init[2 * i] = ~0ull; init[2 * i + 1] = 0; }
    hipMemcpy(span, init.data(), 64 * 16, hipMemcpyHostToDevice); }
  usf::g_pspan = span;
  for (int i = 0; i < 40; ++i) usf::gemm_planes(&g1, 0);      // stamps of the LAST of 40 back-to-back launches (sustained clocks)
  hipDeviceSynchronize();
  usf::g_pspan = nullptr;
  { std::vector<unsigned long long> sp(128); hipMemcpy(sp.data(), span, 64 * 16, hipMemcpyDeviceToHost);
    double din = 0, gap = 0; for (int i = 30; i < 39; ++i) { din += (sp[2 * i + 1] - sp[2 * i]) * 0.01; gap += (sp[2 * i + 2] - sp[2 * i + 1]) * 0.01; }
    printf("  launches 30..38: first wave start -> last wave end %.1f us; last end -> next launch's first start %.1f us\n", din / 9, gap / 9); }
  std::vector<unsigned long long> hd(2 * 16384 * 8); hipMemcpy(hd.data(), dbg, 2 * 16384 * 8 * 8, hipMemcpyDeviceToHost);
  double sm[6] = {0, 0, 0, 0, 0, 0}; int nw = 0;
  for (int w = 0; w < 16384; ++w) if (hd[w * 8 + 4]) { for (int j = 0; j < 6; ++j) sm[j] += hd[w * 8 + j]; ++nw; }
  { unsigned long long mn = ~0ull, mx = 0; double busy = 0;
    for (int w = 0; w < 16384; ++w) if (hd[w * 8 + 4]) { mn = std::min(mn, hd[w * 8 + 6]); mx = std::max(mx, hd[w * 8 + 7]); busy += hd[w * 8 + 7] - hd[w * 8 + 6]; }
    // (the stamp table keeps one entry per (virtual block % 1024, wave): with 1280 virtual blocks the first 256 are overwritten)
    printf("  realtime: first tile start -> last tile end %.1f us; mean tile span %.1f us (x5 rounds = %.1f us)\n", (mx - mn) * 0.01, busy / nw * 0.01, busy / nw * 0.05); }
  { // persistent grid of 256 blocks: physical block b ran virtual blocks b + 256 i (round i)
    unsigned long long k0 = ~0ull;
    for (int w = 0; w < 16384; ++w) if (hd[w * 8 + 4]) k0 = std::min(k0, hd[w * 8 + 6]);
    for (int r = 0; r < 6; ++r) {
      double st = 0, en = 0, mxe = 0; int nb = 0;
      for (int b = 0; b < 256; ++b) { const size_t e = (size_t)(b + 256 * r) * 8 * 8;
        if (b + 256 * r >= 2048 || !hd[e + 4]) continue;
        st += (hd[e + 6] - k0) * 0.01; en += (hd[e + 7] - k0) * 0.01; mxe = std::max(mxe, (hd[e + 7] - k0) * 0.01); ++nb; }
      if (nb) printf("    round %d: mean start %.1f us, mean end %.1f us (span %.1f), last end %.1f us  [%d blocks]\n", r, st / nb, en / nb, (en - st) / nb, mxe, nb);
    } }
#if USF_STAMP >= 2
  { double ph[4] = {0, 0, 0, 0}; int n2 = 0;
    for (int w = 0; w < 16384; ++w) if (hd[w * 8 + 4]) { for (int j = 0; j < 4; ++j) ph[j] += hd[(16384 + w) * 8 + j]; ++n2; }
    printf("  per slab and wave (cycles): pairs before barrier %.0f, stage stores %.0f, barrier %.0f, pairs after %.0f\n",
           ph[0] / n2 / nkb, ph[1] / n2 / nkb, ph[2] / n2 / nkb, ph[3] / n2 / nkb); }
#endif
  printf("  waves %d: cycles per wave: prologue %.0f loop %.0f (%.0f per slab) epilogue %.0f total %.0f; in-kernel clock %.0f MHz\n", nw, sm[0] / nw, sm[1] / nw, sm[1] / nw / nkb, sm[2] / nw, sm[3] / nw, sm[3] / sm[5] * 100.0);
#endif
  return 0;
}
