// Stand-alone check + timing harness for the planes pipeline kernels (tuning aid):
//   pack(A fp32) -> GEMM(planes -> planes, W1, LeakyReLU) -> GEMM(planes -> fp32, W2), compared with a double-precision
//   host computation on sampled rows; then each GEMM variant timed at the given shape.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off [-DUSF_STAMP] tools/exp_planes.hip -o tools/exp_planes_x
//   tools/exp_planes_x [M] [k-blocks] [output blocks]
#include "../usflows_amd/csrc/usf_planes.hip"
#include <stdarg.h>
#include <string.h>
#include <cmath>
#include <vector>
namespace usf { void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fprintf(stderr, "\n"); } }
static unsigned short bf16_rn(float x) { unsigned u; memcpy(&u, &x, 4); u += 0x7fff + ((u >> 16) & 1); return (unsigned short)(u >> 16); }
static float bf16_f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }
// planes image [3][rows][32*nk] of logical W [rows][32*nk] with the slot permutation on K
static std::vector<unsigned short> planes_of(const std::vector<float>& W, int64_t rows, int64_t nk) {
  const int64_t K = 32 * nk;
  std::vector<unsigned short> P(3 * rows * K, 0);
  for (int64_t n = 0; n < rows; ++n) for (int64_t kb = 0; kb < nk; ++kb) for (int s = 0; s < 32; ++s) {
    float x = W[n * K + 32 * kb + usf::plane_feature_of_slot(s)];
    unsigned short h = bf16_rn(x); float r = x - bf16_f(h); unsigned short m = bf16_rn(r); float r2 = r - bf16_f(m);
    P[(0 * rows + n) * K + 32 * kb + s] = h; P[(1 * rows + n) * K + 32 * kb + s] = m; P[(2 * rows + n) * K + 32 * kb + s] = bf16_rn(r2);
  }
  return P;
}
int main(int argc, char** argv) {
  const int64_t M = argc > 1 ? atoll(argv[1]) : 65536, nkb = argc > 2 ? atoll(argv[2]) : 25, nout = argc > 3 ? atoll(argv[3]) : 25;
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
  hipMalloc(&pA, np * nkb * 3072); hipMalloc(&pB, np * nout * 3072); hipMalloc(&W1, P1.size() * 2); hipMalloc(&W2, P2.size() * 2);
  hipMalloc(&idx, K * 4);
  std::vector<int32_t> hidx(K); for (int64_t i = 0; i < K; ++i) hidx[i] = (int32_t)i;
  hipMemcpy(A, hA.data(), M * K * 4, hipMemcpyHostToDevice); hipMemcpy(bias, hb.data(), N * 4, hipMemcpyHostToDevice);
  hipMemcpy(W1, P1.data(), P1.size() * 2, hipMemcpyHostToDevice); hipMemcpy(W2, P2.data(), P2.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(idx, hidx.data(), K * 4, hipMemcpyHostToDevice);
  usf_pack_planes_desc pd = {}; pd.src = A; pd.ld = K; pd.M = M; pd.nkb = nkb; pd.idx = idx; pd.planes = pA;
  usf_gemm_planes_desc g1 = {}; g1.A = pA; g1.a_nkb = nkb; g1.nk = nkb; g1.W_planes = W1; g1.ldw = K; g1.w_plane_stride = N * K; g1.w_rows = N;
  g1.bias = bias; g1.C_planes = pB; g1.c_nkb = nout; g1.c_kbn = nout; g1.M = M; g1.res_sign = 1.f; g1.act = USF_ACT_LEAKY_RELU; g1.slope = 0.01f;
  usf_gemm_planes_desc g2 = {}; g2.A = pB; g2.a_nkb = nout; g2.nk = nout; g2.W_planes = W2; g2.ldw = N; g2.w_plane_stride = N * N; g2.w_rows = N;
  g2.C_f32 = C; g2.ldc = N; g2.N = N; g2.M = M; g2.res_sign = 1.f;
#ifdef USF_STAMP
  unsigned long long* dbg; hipMalloc(&dbg, 8192 * 8 * 8); hipMemset(dbg, 0, 8192 * 8 * 8); usf::g_pdbg = dbg;
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
  printf("check: max abs err %.3e vs max |ref| %.3e -> rel %.2e %s\n", maxerr, maxref, maxerr / maxref, maxerr / maxref < 5e-6 ? "OK" : "FAIL");
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
  usf::gemm_planes(&g1, 0); hipDeviceSynchronize();
  std::vector<unsigned long long> hd(8192 * 8); hipMemcpy(hd.data(), dbg, 8192 * 8 * 8, hipMemcpyDeviceToHost);
  double sm[4] = {0, 0, 0, 0}; int nw = 0;
  for (int w = 0; w < 8192; ++w) if (hd[w * 8 + 4]) { for (int j = 0; j < 4; ++j) sm[j] += hd[w * 8 + j]; ++nw; }
  printf("  waves %d: cycles per wave: prologue %.0f loop %.0f (%.0f per slab) epilogue %.0f total %.0f\n", nw, sm[0] / nw, sm[1] / nw, sm[1] / nw / nkb, sm[2] / nw, sm[3] / nw);
#endif
  return 0;
}
