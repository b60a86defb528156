// Stand-alone timing harness for the bf16x3 fused coupling kernel at the cfg2 shape (tuning aid).
#include "../usflows_amd/csrc/usf_coupling_bf16x3.hip"
#include <stdarg.h>
#include <string.h>
#include <vector>
namespace usf { void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fprintf(stderr, "\n"); } }
static unsigned rs = 777;
static float rnd(float scale) { rs = rs * 1664525u + 1013904223u; return (((rs >> 8) & 0xffff) / 65536.0f - 0.5f) * scale; }
static float* dev_rand(size_t n, float scale) {
  std::vector<float> h(n); for (auto& v : h) v = rnd(scale);
  float* d; hipMalloc(&d, n * 4); hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice); return d;
}
static unsigned short bf16_rn(float x) { unsigned u; memcpy(&u, &x, 4); u += 0x7fff + ((u >> 16) & 1); return (unsigned short)(u >> 16); }
static float bf16_f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }
// three planes [3][rows][ld] of a random matrix
static void* dev_planes(size_t rows, size_t ld, float scale) {
  std::vector<unsigned short> h(3 * rows * ld);
  for (size_t i = 0; i < rows * ld; ++i) {
    float x = rnd(scale); unsigned short a = bf16_rn(x); float r = x - bf16_f(a); unsigned short b = bf16_rn(r); float r2 = r - bf16_f(b);
    h[i] = a; h[rows * ld + i] = b; h[2 * rows * ld + i] = bf16_rn(r2);
  }
  void* d; hipMalloc(&d, h.size() * 2); hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice); return d;
}
int main(int argc, char** argv) {
  const int64_t M = argc > 1 ? atoll(argv[1]) : 65536;
  const int D = 784, H = 256, NP = 392, NTR = 392, KP = 416, NPAD = 416;
  float* z = dev_rand((size_t)M * D, 1.0f);
  usf_coupling_desc d = {};
  d.z = z; d.out = z; d.ldz = D; d.ldo = D; d.M = M; d.off_pass = 392; d.n_pass = NP; d.off_trans = 0; d.n_trans = NTR;
  d.n_hidden = 2; d.hidden[0] = H; d.hidden[1] = H;
  d.b_in = dev_rand(H, 0.05f); d.b_hid[0] = dev_rand(H, 0.05f); d.b_out = dev_rand(NPAD, 0.05f);
  d.split_in = dev_planes(H, KP, 0.05f); d.split_in_ld = KP; d.split_in_plane = (int64_t)H * KP;
  d.split_hid[0] = dev_planes(H, H, 0.05f); d.split_hid_ld = H; d.split_hid_plane = (int64_t)H * H;
  d.split_out = dev_planes(NPAD, H, 0.05f); d.split_out_ld = H; d.split_out_plane = (int64_t)NPAD * H;
  d.sign = -1.f; d.slope = 0.01f; d.act = USF_ACT_LEAKY_RELU;
#ifdef USF_STAMP
  unsigned long long* dbg; hipMalloc(&dbg, 4096 * 8 * 8); hipMemset(dbg, 0, 4096 * 8 * 8); usf::g_c3dbg = dbg;
#endif
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) usf::coupling_bf16x3_dispatch(&d, 0);
  hipDeviceSynchronize();
  const int iters = 20;
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) usf::coupling_bf16x3_dispatch(&d, 0);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= iters;
  const double fl = 2.0 * M * (392.0 * 256 + 256.0 * 256 + 256.0 * 392);
  printf("coupling bf16x3 M=%lld: %.3f ms  %.1f TF/s (fp32-equivalent)\n", (long long)M, ms, fl / ms / 1e9);
#ifdef USF_STAMP
  std::vector<unsigned long long> hd(4096 * 8);
  hipMemcpy(hd.data(), dbg, 4096 * 8 * 8, hipMemcpyDeviceToHost);
  double sm[6] = {0, 0, 0, 0, 0, 0}; int nw = 0;
  for (int w = 0; w < 4096; ++w) if (hd[w * 8 + 5]) { for (int j = 0; j < 6; ++j) sm[j] += hd[w * 8 + j]; ++nw; }
  printf("  waves %d: cycles per wave: phase1 %.0f phase2 %.0f phase3 %.0f total %.0f (MFMA-bound per SIMD: 2 waves x 3264 x 16 = 104448)\n", nw, sm[0] / nw, sm[1] / nw, sm[2] / nw, sm[3] / nw);
#endif
  return 0;
}
