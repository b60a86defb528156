// Stand-alone timing harness for usf_linear_f32 (tuning aid; not part of the library).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off [-DUSF_ABLATE=n] tools/exp_gemm.hip -o tools/exp_gemm_n
#include "../usflows_amd/csrc/usf_linear.hip"
#include <stdarg.h>
#ifndef USF_ABLATE
#define USF_ABLATE 0
#endif
#include <vector>
namespace usf { void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fprintf(stderr, "\n"); } }

int main(int argc, char** argv) {
  const int64_t M = argc > 1 ? atoll(argv[1]) : 65536, N = argc > 2 ? atoll(argv[2]) : 784, K = argc > 3 ? atoll(argv[3]) : 784;
  const int iters = 20;
  float *A, *W, *C, *bias;
  hipMalloc(&A, M * K * 4); hipMalloc(&W, N * K * 4); hipMalloc(&C, M * N * 4); hipMalloc(&bias, N * 4);
  std::vector<float> h(M * K);
  unsigned s = 12345;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.0f - 0.5f; }
  hipMemcpy(A, h.data(), M * K * 4, hipMemcpyHostToDevice);
  hipMemcpy(W, h.data(), N * K * 4, hipMemcpyHostToDevice);
  hipMemcpy(bias, h.data(), N * 4, hipMemcpyHostToDevice);
  usf_linear_desc d = {};
  d.A = A; d.lda = K; d.W = W; d.ldw = K; d.bias = bias; d.C = C; d.ldc = N; d.M = M; d.N = N; d.K = K; d.res_sign = 1.f;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
#ifdef USF_STAMP
  unsigned long long* dbg; hipMalloc(&dbg, 4096 * 8 * 8); hipMemset(dbg, 0, 4096 * 8 * 8);
  usf::g_dbg = dbg;
#endif
  for (int i = 0; i < 3; ++i) usf::linear_dispatch(&d, 0);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) usf::linear_dispatch(&d, 0);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= iters;
  printf("ABLATE=%d tile=%s M=%lld N=%lld K=%lld: %.3f ms  %.1f TF/s\n", USF_ABLATE, getenv("USF_LINEAR_TILE") ? getenv("USF_LINEAR_TILE") : "auto",
         (long long)M, (long long)N, (long long)K, ms, 2.0 * M * N * K / ms / 1e9);
#ifdef USF_STAMP
  std::vector<unsigned long long> hd(4096 * 8);
  hipMemcpy(hd.data(), dbg, 4096 * 8 * 8, hipMemcpyDeviceToHost);
  double sm[6] = {0, 0, 0, 0, 0, 0}; int nw = 0;
  for (int w = 0; w < 4096; ++w) if (hd[w * 8 + 5]) { for (int j = 0; j < 6; ++j) sm[j] += hd[w * 8 + j]; ++nw; }
  printf("  waves %d: per wave cycles: loop %.0f last %.0f epi %.0f sync %.0f total %.0f tiles %.1f\n", nw, sm[0] / nw, sm[1] / nw, sm[2] / nw, sm[3] / nw, sm[4] / nw, sm[5] / nw);
#endif
  return 0;
}
