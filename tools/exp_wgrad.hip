// Stamp harness for the loader-wave weight-gradient kernel (tuning aid; correctness: tests/test_train_kernels_gpu.py)
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DUSF_STAMP tools/exp_wgrad.hip -o tools/exp_wgrad_s
//   tools/exp_wgrad_s [M] [N] [K]
#include "../usflows_amd/csrc/usf_train.hip"
#include <stdarg.h>
#include <vector>
namespace usf { void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fprintf(stderr, "\n"); } }
int main(int argc, char** argv) {
  const int64_t M = argc > 1 ? atoll(argv[1]) : 65536, N = argc > 2 ? atoll(argv[2]) : 784, K = argc > 3 ? atoll(argv[3]) : 784;
  std::vector<float> hy(M * N), ha(M * K);
  unsigned s = 777;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
  for (auto& v : hy) v = rnd();
  for (auto& v : ha) v = rnd() * 3.f;
  float *Y, *A, *G, *ws; int64_t wsf = 0;
  usf::wgrad_workspace_floats(M, N, K, &wsf);
  hipMalloc(&Y, M * N * 4); hipMalloc(&A, M * K * 4); hipMalloc(&G, N * K * 4); hipMalloc(&ws, wsf * 4);
  hipMemcpy(Y, hy.data(), M * N * 4, hipMemcpyHostToDevice); hipMemcpy(A, ha.data(), M * K * 4, hipMemcpyHostToDevice);
#ifdef USF_STAMP
  unsigned long long* dbg; hipMalloc(&dbg, 1024 * 12 * 4 * 8); hipMemset(dbg, 0, 1024 * 12 * 4 * 8); usf::g_wdbg = dbg;
#endif
  for (int i = 0; i < 3; ++i) if (usf::wgrad(Y, N, A, K, M, N, K, G, K, 1.f, 0.f, 1, ws, wsf, 0, nullptr, 0.f, 0.f)) return 1;
  if (hipDeviceSynchronize() != hipSuccess) { printf("kernel fault\n"); return 2; }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  const int it = 20;
  for (int i = 0; i < it; ++i) usf::wgrad(Y, N, A, K, M, N, K, G, K, 1.f, 0.f, 1, ws, wsf, 0, nullptr, 0.f, 0.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= it;
  printf("wgrad M=%lld N=%lld K=%lld: %.3f ms  %.1f TF/s fp32-equivalent\n", (long long)M, (long long)N, (long long)K, ms, 2.0 * M * N * K / ms / 1e9);
#ifdef USF_STAMP
  std::vector<unsigned long long> h(1024 * 12 * 4); hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost);
  double w[2] = {0, 0}, b[2] = {0, 0}, sl[2] = {0, 0}; int n[2] = {0, 0};
  for (int i = 0; i < 1024 * 12; ++i) if (h[4 * i + 3]) { const int r = (i % 12) >= 4; w[r] += h[4 * i]; b[r] += h[4 * i + 1]; sl[r] += h[4 * i + 2]; ++n[r]; }
  for (int r = 0; r < 2; ++r) if (n[r]) printf("  %s waves (%d): per slab: work %.0f cycles, barrier wait %.0f cycles\n", r ? "loader" : "MFMA  ", n[r], w[r] / sl[r], b[r] / sl[r]);
#endif
  return 0;
}
