// Stand-alone timing harness for the bf16x3 linear kernel (tuning aid).
#include "../usflows_amd/csrc/usf_linear_bf16x3.hip"
#include <stdarg.h>
#include <string.h>
#include <vector>
namespace usf { void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fprintf(stderr, "\n"); } }
static unsigned short bf16_rn(float x) { unsigned u; memcpy(&u, &x, 4); u += 0x7fff + ((u >> 16) & 1); return (unsigned short)(u >> 16); }
static float bf16_f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }
int main(int argc, char** argv) {
  const int64_t M = argc > 1 ? atoll(argv[1]) : 65536, N = argc > 2 ? atoll(argv[2]) : 784, K = argc > 3 ? atoll(argv[3]) : 784;
  const int64_t Kp = (K + 31) / 32 * 32;
  std::vector<float> hA(M * K), hW(N * K);
  unsigned s = 12345;
  for (auto& v : hA) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.0f - 0.5f; }
  for (auto& v : hW) { s = s * 1664525u + 1013904223u; v = (((s >> 8) & 0xffff) / 65536.0f - 0.5f) * 0.1f; }
  std::vector<unsigned short> hP(3 * N * Kp, 0);
  for (int64_t n = 0; n < N; ++n) for (int64_t k = 0; k < K; ++k) {
    float x = hW[n * K + k]; unsigned short h = bf16_rn(x); float r = x - bf16_f(h); unsigned short m = bf16_rn(r); float r2 = r - bf16_f(m);
    hP[(0 * N + n) * Kp + k] = h; hP[(1 * N + n) * Kp + k] = m; hP[(2 * N + n) * Kp + k] = bf16_rn(r2);
  }
  float *A, *W, *C, *bias; void* P;
  hipMalloc(&A, M * K * 4); hipMalloc(&W, N * K * 4); hipMalloc(&C, M * N * 4); hipMalloc(&bias, N * 4); hipMalloc(&P, hP.size() * 2);
  hipMemcpy(A, hA.data(), M * K * 4, hipMemcpyHostToDevice); hipMemcpy(W, hW.data(), N * K * 4, hipMemcpyHostToDevice);
  hipMemcpy(bias, hA.data(), N * 4, hipMemcpyHostToDevice); hipMemcpy(P, hP.data(), hP.size() * 2, hipMemcpyHostToDevice);
  usf_linear_desc d = {};
  d.A = A; d.lda = K; d.W = W; d.ldw = K; d.bias = bias; d.C = C; d.ldc = N; d.M = M; d.N = N; d.K = K; d.res_sign = 1.f;
  d.W_split = P; d.ldw_split = Kp; d.split_plane_stride = N * Kp;
#ifdef USF_STAMP
  unsigned long long* dbg; hipMalloc(&dbg, 2 * 8192 * 8 * 8); hipMemset(dbg, 0, 2 * 8192 * 8 * 8); usf::g_bdbg = dbg;
#endif
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) usf::linear_bf16x3_dispatch(&d, 0);
  hipDeviceSynchronize();
  const int iters = 20;
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) usf::linear_bf16x3_dispatch(&d, 0);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= iters;
  printf("bf16x3 M=%lld N=%lld K=%lld: %.3f ms  %.1f TF/s (fp32-equivalent)\n", (long long)M, (long long)N, (long long)K, ms, 2.0 * M * N * K / ms / 1e9);
#ifdef USF_STAMP
  std::vector<unsigned long long> hd(2 * 8192 * 8);
  hipMemcpy(hd.data(), dbg, 2 * 8192 * 8 * 8, hipMemcpyDeviceToHost);
  double sm[6] = {0, 0, 0, 0, 0, 0}; int nw = 0;
  for (int w = 0; w < 8192; ++w) if (hd[w * 8 + 5]) { for (int j = 0; j < 6; ++j) sm[j] += hd[w * 8 + j]; ++nw; }
  { double pq[5] = {0, 0, 0, 0, 0};
    for (int w = 0; w < 8192; ++w) if (hd[w * 8 + 5]) for (int j = 0; j < 5; ++j) pq[j] += hd[8192 * 8 + w * 8 + j];
    printf("  loop phases per wave: issue %.0f step0 %.0f step1+split %.0f store %.0f barrier %.0f\n", pq[0] / nw, pq[1] / nw, pq[2] / nw, pq[3] / nw, pq[4] / nw); }
  printf("  waves %d: cycles per wave: prologue %.0f loop %.0f last %.0f epilogue %.0f total %.0f\n", nw, sm[0] / nw, sm[1] / nw, sm[2] / nw, sm[3] / nw, sm[4] / nw);
#endif
  return 0;
}
