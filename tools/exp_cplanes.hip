// Timing harness for the fused coupling on planes (tuning aid; correctness lives in tests/test_planes_gpu.py):
//   cfg2 shape by default: 65 536 rows, z of 25 blocks, conditioning blocks 0..12, transformed blocks 12..24, hidden 256.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off [-DUSF_STAMP] tools/exp_cplanes.hip -o tools/exp_cplanes_x
//   tools/exp_cplanes_x [M] [NPL] [n_hidden]
#include "../usflows_amd/csrc/usf_coupling_planes.hip"
#include <stdarg.h>
#include <string.h>
#include <vector>
namespace usf {
void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fprintf(stderr, "\n"); }
}
static int NPL = 3;
static unsigned short bf16_rn(float x) { unsigned u; memcpy(&u, &x, 4); u += 0x7fff + ((u >> 16) & 1); return (unsigned short)(u >> 16); }
static float bf16_f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }
static unsigned short f16_bits(_Float16 h) { unsigned short u; memcpy(&u, &h, 2); return u; }
static unsigned rs = 4242;
static float rnd() { rs = rs * 1664525u + 1013904223u; return ((rs >> 8) & 0xffff) / 65536.0f - 0.5f; }
// n values -> NPL planes laid out [plane][n]
static void split_into(const std::vector<float>& x, std::vector<unsigned short>& P) {
  const size_t n = x.size(); P.assign(NPL * n, 0);
  for (size_t i = 0; i < n; ++i) {
    if (NPL == 3) { unsigned short h = bf16_rn(x[i]); float r = x[i] - bf16_f(h); unsigned short m = bf16_rn(r); P[i] = h; P[n + i] = m; P[2 * n + i] = bf16_rn(r - bf16_f(m)); }
    else { _Float16 h = (_Float16)x[i]; P[i] = f16_bits(h); P[n + i] = f16_bits((_Float16)(x[i] - (float)h)); }
  }
}
int main(int argc, char** argv) {
  const int64_t M = argc > 1 ? atoll(argv[1]) : 65536;
  if (argc > 2) NPL = atoi(argv[2]);
  const int NH = argc > 3 ? atoi(argv[3]) : 2;
  const int64_t nkb = 25, nkp = 13, nkt = 13, np = (M + 15) / 16;
  // z planes: a 64-panel random pattern replicated (chunk = NPL KiB: planes of a chunk are consecutive)
  const int64_t pat_panels = 64, chunk_elems = 512;
  std::vector<unsigned short> zpat(pat_panels * nkb * NPL * chunk_elems);
  for (int64_t c = 0; c < pat_panels * nkb; ++c) {
    std::vector<float> x(chunk_elems); for (auto& v : x) v = rnd() * 4.f;
    std::vector<unsigned short> P; split_into(x, P);
    memcpy(&zpat[c * NPL * chunk_elems], P.data(), NPL * chunk_elems * 2);
  }
  char* z; hipMalloc(&z, np * nkb * NPL * 1024);
  const size_t patb = zpat.size() * 2;
  for (int64_t p0 = 0; p0 < np; p0 += pat_panels) hipMemcpy(z + p0 * nkb * NPL * 1024, zpat.data(), std::min<int64_t>(pat_panels, np - p0) * nkb * NPL * 1024, hipMemcpyHostToDevice);
  (void)patb;
  auto wplanes = [&](int64_t rows, int64_t K, float scale) {
    std::vector<float> w(rows * K); for (auto& v : w) v = rnd() * scale;
    std::vector<unsigned short> P; split_into(w, P);
    void* d; hipMalloc(&d, P.size() * 2); hipMemcpy(d, P.data(), P.size() * 2, hipMemcpyHostToDevice); return d;
  };
  auto bias = [&](int64_t n) { std::vector<float> b(n); for (auto& v : b) v = rnd() * 0.1f; float* d; hipMalloc(&d, n * 4); hipMemcpy(d, b.data(), n * 4, hipMemcpyHostToDevice); return d; };
  usf_coupling_planes_desc d = {};
  d.z = z; d.z_nkb = nkb; d.M = M; d.kb_p0 = 0; d.nk_p = nkp; d.kb_t0 = 12; d.nk_t = nkt; d.n_hidden = NH; d.hidden_padded = 256;
  d.W_in = wplanes(256, 32 * nkp, 0.08f); d.ldw_in = 32 * nkp; d.w_in_plane = 256 * 32 * nkp; d.b_in = bias(256);
  for (int i = 0; i + 1 < NH; ++i) { d.W_hid[i] = wplanes(256, 256, 0.1f); d.b_hid[i] = bias(256); }
  d.ldw_hid = 256; d.w_hid_plane = 256 * 256;
  d.W_out = wplanes(32 * nkt, 256, 0.0005f); d.ldw_out = 256; d.w_out_plane = 32 * nkt * 256; d.b_out = bias(32 * nkt);
  d.sign = 1.f; d.slope = 0.01f; d.act = USF_ACT_LEAKY_RELU; d.format = NPL == 2 ? USF_PLANES_F16X2 : USF_PLANES_BF16X3;
  int32_t* flag; hipMalloc(&flag, 4); hipMemset(flag, 0, 4); d.range_flag = flag;
#ifdef USF_STAMP
  unsigned long long* dbg; hipMalloc(&dbg, 2048 * 8 * 12 * 8); hipMemset(dbg, 0, 2048 * 8 * 12 * 8); usf::g_cdbg = dbg;
#endif
  // ---- the two kernels on the same input must agree bit for bit (same summation order per accumulator) ----
  {
    const size_t zb = (size_t)np * nkb * NPL * 1024;
    char* z2; hipMalloc(&z2, zb); hipMemcpy(z2, z, zb, hipMemcpyDeviceToDevice);
    char* z0; hipMalloc(&z0, zb); hipMemcpy(z0, z, zb, hipMemcpyDeviceToDevice);
    usf::g_cp_w32 = 0; d.z = z; if (usf::coupling_planes(&d, 0)) return 1;
    usf::g_cp_w32 = 1; d.z = z2; if (usf::coupling_planes(&d, 0)) return 1;
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel fault\n"); return 2; }
    std::vector<unsigned short> a(zb / 2), b(zb / 2), c(zb / 2);
    hipMemcpy(a.data(), z, zb, hipMemcpyDeviceToHost); hipMemcpy(b.data(), z2, zb, hipMemcpyDeviceToHost); hipMemcpy(c.data(), z0, zb, hipMemcpyDeviceToHost);
    size_t diff = 0, changed = 0, first = (size_t)-1;
    size_t lastchunk = (size_t)-1; int shown = 0;
    for (size_t i = 0; i < a.size(); ++i) {
      if (a[i] != b[i]) {
        ++diff; if (first == (size_t)-1) first = i;
        const size_t ch = i / (NPL * 512);
        if (ch != lastchunk && shown < 12) { printf("   differs: panel %zu block %zu plane %zu lane %zu slot %zu: 16-row %04x, 32-row %04x, before %04x\n", ch / nkb, ch % nkb, (i % (NPL * 512)) / 512, (i % 512) / 8, i % 8, a[i], b[i], c[i]); ++shown; }
        lastchunk = ch;
      }
      if (a[i] != c[i]) ++changed;
    }
    printf("16-row kernel vs 32-row kernel: %zu of %zu plane elements differ (first at %zd: panel %zd block %zd); the layer changed %zu elements -> %s\n",
           diff, a.size(), (ssize_t)first, first == (size_t)-1 ? -1 : (ssize_t)(first / (nkb * NPL * 512)), first == (size_t)-1 ? -1 : (ssize_t)((first / (NPL * 512)) % nkb), changed, diff == 0 && changed > 0 ? "OK" : "FAIL");
    hipMemcpy(z, z0, zb, hipMemcpyDeviceToDevice); d.z = z; hipFree(z2); hipFree(z0);
  }
  for (int variant = 0; variant < 2; ++variant) {
  usf::g_cp_w32 = variant;
  for (int i = 0; i < 5; ++i) if (usf::coupling_planes(&d, 0)) return 1;
  if (hipDeviceSynchronize() != hipSuccess) { printf("kernel fault\n"); return 2; }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 40;
  hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) usf::coupling_planes(&d, 0);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= iters;
  const double flops = 2.0 * M * (392.0 * 256 + (NH - 1) * 256.0 * 256 + 256.0 * 392);
  int32_t hf; hipMemcpy(&hf, flag, 4, hipMemcpyDeviceToHost);
  printf("[%s NH=%d %s] coupling on planes M=%lld: %.4f ms  %.1f TF/s fp32-equivalent (algorithmic)  range flag %d\n", NPL == 2 ? "f16x2" : "bf16x3", NH, variant ? "32-row waves" : "16-row waves", (long long)M, ms, flops / ms / 1e9, hf);
#ifdef USF_STAMP
  std::vector<unsigned long long> h(2048 * 8 * 12); hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost);
  double sm[3] = {0, 0, 0}; int nw = 0;
  for (size_t i = 0; i < 2048 * 8; ++i) if (h[4 * i + 3]) { for (int j = 0; j < 3; ++j) sm[j] += (double)h[4 * i + j]; ++nw; }
  const double tot = (sm[0] + sm[1] + sm[2]) / nw;
  const int npr = NPL == 3 ? 6 : 3;
  printf("  waves %d: cycles per wave: phase 1 %.0f (MFMA issue of a SIMD's two waves %d), phase 2 %.0f (%d), phase 3 %.0f (%d), total %.0f -> %.1f %% matrix-pipe issue\n",
         nw, sm[0] / nw, 2 * 13 * 16 * npr * 16, sm[1] / nw, 2 * (NH - 1) * 8 * 16 * npr * 16, sm[2] / nw, 2 * 13 * 16 * npr * 16,
         tot, 100.0 * (2.0 * (13 * 16 * 2 + (NH - 1) * 8 * 16) * npr * 16) / tot);
#if USF_STAMP >= 2
  if (variant == 1) { double bw[3] = {0, 0, 0}, mn = 0, mx = 0; int n2 = 0;
    for (size_t i = 0; i < 2048 * 8; ++i) if (h[4 * i + 3]) { const unsigned long long* o2 = &h[2048 * 8 * 4 + 8 * i]; for (int j = 0; j < 3; ++j) bw[j] += (double)o2[j]; mn += (double)o2[3]; mx += (double)o2[4]; ++n2; }
    printf("  32-row waves: barrier wait per wave and phase %.0f / %.0f / %.0f cycles; barrier-to-barrier interval min %.0f max %.0f (ideal 3072)\n", bw[0] / n2, bw[1] / n2, bw[2] / n2, mn / n2, mx / n2); }
#endif
  hipMemset(dbg, 0, 2048 * 8 * 12 * 8);
#endif
  }
  return 0;
}
