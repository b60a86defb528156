// Harness for the weight gradient from pre-split planes (tuning aid; correctness: tests/test_train_kernels_gpu.py)
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off tools/exp_wgradp.hip -o tools/exp_wgradp
//   (-DUSF_STAMP: per-slab work / barrier-wait cycles of the planes kernel's MFMA and loader waves -> tools/exp_wgradp_s)
//   tools/exp_wgradp [M] [N] [K]      -- times usf_wgrad_f32 (loader waves) and usf_wgrad_planes_f32 and compares their bits
#include "../usflows_amd/csrc/usf_train.hip"
#include "../usflows_amd/csrc/usf_wgrad_planes.hip"
#include <stdarg.h>
#include <vector>
#include <string.h>
namespace usf { void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fprintf(stderr, "\n"); } }
int main(int argc, char** argv) {
  const int64_t M = argc > 1 ? atoll(argv[1]) : 65536, N = argc > 2 ? atoll(argv[2]) : 784, K = argc > 3 ? atoll(argv[3]) : 784;
  std::vector<float> hy(M * N), ha(M * K);
  unsigned s = 777;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
  for (auto& v : hy) v = rnd();
  for (auto& v : ha) v = rnd() * 3.f;
  float *Y, *A, *G, *G2, *ws; int64_t wsf = 0;
  usf::wgrad_workspace_floats(M, N, K, &wsf);
  { const int64_t w2 = usf::wgrad_planes_workspace_floats(M, N, K); if (w2 > wsf) wsf = w2; }
  const int64_t Mp = (M + 31) / 32 * 32, ldy = (N + 31) / 32 * 32, lda = (K + 31) / 32 * 32;
  void *Yp, *Ap;
  float* CS = nullptr; if (getenv("EXP_COLSUM")) hipMalloc(&CS, N * 4);
  hipMalloc(&Y, M * N * 4); hipMalloc(&A, M * K * 4); hipMalloc(&G, N * K * 4); hipMalloc(&G2, N * K * 4); hipMalloc(&ws, wsf * 4);
  hipMalloc(&Yp, 3 * Mp * ldy * 2); hipMalloc(&Ap, 3 * Mp * lda * 2);
  hipMemcpy(Y, hy.data(), M * N * 4, hipMemcpyHostToDevice); hipMemcpy(A, ha.data(), M * K * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int it = 20;
  float ms;
  for (int i = 0; i < 3; ++i) if (usf::wgrad(Y, N, A, K, M, N, K, G, K, 1.f, 0.f, 1, ws, wsf, 0, nullptr, 0.f, 0.f)) return 1;
  hipEventRecord(e0);
  for (int i = 0; i < it; ++i) usf::wgrad(Y, N, A, K, M, N, K, G, K, 1.f, 0.f, 1, ws, wsf, 0, nullptr, 0.f, 0.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  hipEventElapsedTime(&ms, e0, e1); ms /= it;
  printf("wgrad (fp32 operands)  M=%lld N=%lld K=%lld: %.3f ms  %.1f TF/s fp32-equivalent\n", (long long)M, (long long)N, (long long)K, ms, 2.0 * M * N * K / ms / 1e9);
  if (usf::split_planes(Y, N, M, N, Yp, ldy, Mp * ldy, 0) || usf::split_planes(A, K, M, K, Ap, lda, Mp * lda, 0)) return 1;
  hipEventRecord(e0);
  for (int i = 0; i < it; ++i) usf::split_planes(Y, N, M, N, Yp, ldy, Mp * ldy, 0);
  hipEventRecord(e1); hipEventSynchronize(e1);
  hipEventElapsedTime(&ms, e0, e1); ms /= it;
  printf("split_planes %lld x %lld: %.3f ms  (%.2f TB/s)\n", (long long)M, (long long)N, ms, 10.0 * M * N / ms / 1e9);
#ifdef USF_STAMP
  unsigned long long* dbg; hipMalloc(&dbg, 1024 * 12 * 4 * 8); hipMemset(dbg, 0, 1024 * 12 * 4 * 8); usf::g_wpdbg = dbg;
#endif
  for (int i = 0; i < 3; ++i) if (usf::wgrad_planes(Yp, ldy, Mp * ldy, 0, Ap, lda, Mp * lda, 0, M, N, K, G2, K, 1.f, 0.f, CS, 1.f, 0.f, ws, wsf, 0)) return 1;
  if (hipDeviceSynchronize() != hipSuccess) { printf("kernel fault\n"); return 2; }
  hipEventRecord(e0);
  for (int i = 0; i < it; ++i) usf::wgrad_planes(Yp, ldy, Mp * ldy, 0, Ap, lda, Mp * lda, 0, M, N, K, G2, K, 1.f, 0.f, CS, 1.f, 0.f, ws, wsf, 0);
  hipEventRecord(e1); hipEventSynchronize(e1);
  hipEventElapsedTime(&ms, e0, e1); ms /= it;
  printf("wgrad (planes)         M=%lld N=%lld K=%lld: %.3f ms  %.1f TF/s fp32-equivalent\n", (long long)M, (long long)N, (long long)K, ms, 2.0 * M * N * K / ms / 1e9);
#ifdef USF_STAMP
  {
    std::vector<unsigned long long> h(1024 * 12 * 4); hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost);
    double w[2] = {0, 0}, b[2] = {0, 0}, sl[2] = {0, 0}; int n[2] = {0, 0};
    for (int i = 0; i < 1024 * 8; ++i) if (h[4 * i + 3]) { const int r = (i % 8) >= 4; w[r] += h[4 * i]; b[r] += h[4 * i + 1]; sl[r] += h[4 * i + 2]; ++n[r]; }
    for (int r = 0; r < 2; ++r) if (n[r]) printf("  %s waves (%d): per slab: work %.0f cycles, barrier wait %.0f cycles\n", r ? "loader" : "MFMA  ", n[r], w[r] / sl[r], b[r] / sl[r]);
  }
#endif
  std::vector<float> g(N * K), g2(N * K);
  hipMemcpy(g.data(), G, N * K * 4, hipMemcpyDeviceToHost); hipMemcpy(g2.data(), G2, N * K * 4, hipMemcpyDeviceToHost);
  size_t diff = 0; double mx = 0;
  for (size_t i = 0; i < g.size(); ++i) { if (memcmp(&g[i], &g2[i], 4)) ++diff; const double d = fabs((double)g[i] - g2[i]); if (d > mx) mx = d; }
  double gm = 0; for (float v : g) gm = fmax(gm, fabs((double)v));
  printf("bits differ in %zu of %zu entries, max |diff| %.3g (max |g| %.3g)\n", diff, g.size(), mx, gm);
  { char buf[1024]; usf::wgrad_planes_describe(M, N, K, buf, sizeof(buf)); printf("  schedule: %s\n", buf); }
  return mx > 1e-4 * gm ? 3 : 0;
}
