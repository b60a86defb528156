// Split-precision variant of the fused dense layer:  C = A @ W^T + bias  with fp32-equivalent accuracy on
// the bf16 matrix cores (v_mfma_f32_16x16x32_bf16, 16x the f32 MFMA rate; on random data the chip holds a ~17 %
// higher clock on this shape than on 32x32x16 at equal cycles per flop -- tools/exp_mfma_peak.hip).
//
// Every fp32 operand is written as the exact sum of three bf16 numbers (round-to-nearest residual split:
// x = x1 + x2 + x3, 8 + 8 + 8 significant bits) and the product is expanded, keeping the six terms whose
// weight is >= 2^-16 of the leading one:
//      a.w ~= a1 w1 + (a1 w2 + a2 w1) + (a1 w3 + a2 w2 + a3 w1)            (dropped: 2^-24 relative)
// Products of bf16 numbers are exact in fp32 and the MFMA accumulates in fp32, so the result carries the
// same error class as an fp32 FMA chain (measured through the whole 65-layer cfg2 flow: 6.9e-7 vs 6.7e-7 max
// relative error against the fp64 reference) at 16/6 = 2.7x the f32-MFMA throughput.
//   * W is split once at parameter-pack time (three bf16 planes, K padded to 32 with zeros).
//   * A stays fp32 in HBM (the coupling kernels read/write it); each wave splits its own operand
//     fragments in registers (v_cvt_pk_bf16_f32 + subtracts) -- VALU work that runs beside the MFMAs.
//   * Same tiling idioms as usf_linear.hip: waves tile M only, activation fragments global -> registers,
//     weight planes through LDS (k-chunk-major image, register-staged, double-buffered), XCD-aware block
//     map, bias in the accumulator init, LDS-transposed whole-cache-line stores.
#include <stdlib.h>

#include "usf_common.h"

namespace usf {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct Lin3Args {
  const float* A; const __bf16* Wp; const float* bias; const float* post_mul; float* C;
  const float* pre_div; const float* pre_sub;
  const float* residual; const float* addend; int64_t ldr, ldadd; float res_sign;
  int64_t lda, ldc, ldwp, plane_stride;
  int M, N, K, nbm, nbn;
  float slope; int act;
  int bias_vec;
  unsigned long long* dbg;              // tuning builds only (USF_STAMP)
  __bf16* Apl; unsigned ldpl, plstride, plbytes;   // side output: planes of A (usf_linear_desc::A_planes_out), or NULL
};

__device__ __forceinline__ void split3(const f32x4 x0, const f32x4 x1, bf16x8& p1, bf16x8& p2, bf16x8& p3) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float x = (j < 4) ? x0[j & 3] : x1[j & 3];
    const __bf16 h = (__bf16)x;
    const float r = x - (float)h;              // exact
    const __bf16 m = (__bf16)r;
    const float r2 = r - (float)m;             // exact
    p1[j] = h; p2[j] = m; p3[j] = (__bf16)r2;
  }
}

template <int TN, int WM, bool PRO, int NB, bool SIDE = false>
__global__ __launch_bounds__(WM * 64, 2) void linear_bf16x3_kernel(const Lin3Args p) {
  constexpr int NT = WM * 64;
  constexpr int BM = WM * 32;
  constexpr int BN = TN * 32;
  constexpr int BK = 32;                       // two 16-k MFMA steps per slab
  // Weight stage = 3 planes x 4 k-chunks x BN rows of 16-byte slots; LDS image slot(plane, chunk c, row r) =
  // (plane * 4 + c) * BN + (r ^ 2c).  The chunk stride is a multiple of 256 B, which is what the lane groups
  // of ds_read_b128 ({0-3,12-15,20-27}, ...) want from the MFMA fragment pattern (lane (j, g) reads row
  // 16 ft + j of chunk g: conflict free).  The slots are dealt to the threads row-major (4 consecutive lanes =
  // the 4 chunks = 64 contiguous bytes of one row; a 16-lane group = 4 cache lines for the address unit), and
  // the XOR keeps those stores conflict free as well: the 8 lanes of a ds_write_b128 group (2 rows x 4 chunks)
  // land on 8 different 16-B bank groups.
  // The last round of the deal is partial: its surplus threads repeat the first slots of the deal (same
  // source, same destination, same bytes as the owner writes), so the K loop stays ONE basic block (a
  // branch around the store lets the compiler sink the operand split behind it, out of the MFMA shadow).
  constexpr int CS = BN;
  constexpr int NSLOT = 3 * 4 * BN;
  constexpr int NWV = (NSLOT + NT - 1) / NT;   // float4 staged per thread per slab
  constexpr int IMG = 12 * CS;
  static_assert(NWV * NT - NSLOT <= NSLOT, "surplus threads wrap once");
  static_assert(BN % 32 == 0, "chunk stride arithmetic assumes 16-row multiples");
  constexpr int STG = IMG * 4;                 // floats per staging buffer
  // Weight slabs live in a ring of NB buffers: slab s + NB/2 is staged while slab s multiplies, and the block
  // meets at a barrier once per NB/2 slabs (NB = 4: every second slab -- with one 8-wave block per CU the pipe
  // idles ~900 cycles around each barrier).  The scratch is a separate LDS object so the compiler can tell its
  // traffic from the ring's.  Activation slabs are fetched in whole 128-B lines (8 lanes per row) and turned
  // into MFMA operand fragments through the scratch; the epilogue sends the output tiles the other way.
  static_assert(NB == 2 || NB == 4, "ring of 2 or 4 weight buffers");
  constexpr int D = NB / 2;
  __shared__ __attribute__((aligned(16))) float wring[NB * STG];
  __shared__ __attribute__((aligned(16))) float ascr[WM * 1024];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int lj = lane & 15;                    // MFMA 16x16x32: lane = (row-in-tile j, k-group g)
  const int lg = lane >> 4;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  const int bid = blockIdx.x;
  const int xcd = bid & 7;
  const int seq = bid >> 3;
  const int panel = (seq / p.nbn) * 8 + xcd;
  const int bn = seq % p.nbn;
  if (panel >= p.nbm) return;
  const int row0 = panel * BM + wave * 32;
  const int n0 = bn * BN;

  // ---- activations: lane = (row ar + 8 i, 16-B chunk ac) of the wave's 32 x 32 slab, i = 0..3 ----
  const int ar = lane >> 3, ac = lane & 7;
  unsigned arow[4];                            // element offsets from p.A (the dispatcher checks they fit 32 bits)
#pragma unroll
  for (int i = 0; i < 4; ++i) arow[i] = (unsigned)min(row0 + ar + 8 * i, p.M - 1) * (unsigned)p.lda;
  // PRO: A' = A / pre_div - pre_sub before the split (ScaleTransform.backward + bias of the tail affine
  // layer, transforms.py:116-125, 960); one k-chunk per lane serves its four rows
  const float* pdiv = p.pre_div ? p.pre_div : p.pre_sub;
  const float* psub = p.pre_sub ? p.pre_sub : p.pre_div;
  const bool has_div = p.pre_div != nullptr, has_sub = p.pre_sub != nullptr;
  const f32x4 one4 = {1.f, 1.f, 1.f, 1.f};
  f32x4 dvr = one4, svr = zero4;
  auto issue_a = [&](int k0, f32x4 (&dst)[4]) {
    const int kc = min(k0 + 4 * ac, p.K - 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) dst[i] = *reinterpret_cast<const f32x4*>(p.A + (arow[i] + (unsigned)kc));
    if (PRO) {
      dvr = *reinterpret_cast<const f32x4*>(pdiv + kc);
      svr = *reinterpret_cast<const f32x4*>(psub + kc);
    }
  };
  // scratch image: row r = 32 floats, its 16-B chunk c stored at position c ^ swz(r), swz a bit shuffle of
  // h = (r >> 1) & 7: (h2 ^ h1, h0, h1).  With the lane groups ds_read_b128 / ds_write_b128 are served in, the
  // line-shaped accesses (8 lanes per row) and the fragment-shaped ones (one row per lane, k-chunk by lane >> 4)
  // are conflict free in both directions (activations in, output tiles out).
  auto swz = [](int r) { const int h = r >> 1; return (((h >> 2) ^ (h >> 1)) & 1) | ((h & 1) << 1) | (((h >> 1) & 1) << 2); };
  float* const scr = ascr + wave * 1024;
  auto transpose_a = [&](int k0, f32x4 (&src)[4], f32x4 (&frag)[4]) {
    // (no zero fill past K: those lanes hold a clamped, finite re-read and the weight planes are zero there --
    // the header's padding contract)
    (void)k0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f32x4 v = src[i];
      if (PRO) v = v / (has_div ? dvr : one4) - (has_sub ? svr : zero4);
      const int r = ar + 8 * i;
      *reinterpret_cast<f32x4*>(scr + r * 32 + 4 * (ac ^ swz(r))) = v;
    }
    // operand fragment of batch tile b (16 rows): lane (j, g) holds row 16 b + j, k = 8 g + (0..7) = chunks 2 g + u
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int r = 16 * b + lj;
        frag[2 * b + u] = *reinterpret_cast<const f32x4*>(scr + r * 32 + 4 * ((2 * lg + u) ^ swz(r)));
      }
  };

  // ---- weights ----
  unsigned wsrc[NWV];                          // element offsets from p.Wp
  int wdst[NWV];
#pragma unroll
  for (int i = 0; i < NWV; ++i) {
    const int idx = tid + NT * i;
    const int idc = (idx < NSLOT) ? idx : idx - NSLOT;
    const int pl = idc / (4 * BN), rem = idc % (4 * BN);
    const int r = rem >> 2, ch = rem & 3;
    wsrc[i] = (unsigned)(pl * p.plane_stride + (int64_t)min(n0 + r, p.N - 1) * p.ldwp + 8 * ch);
    wdst[i] = 4 * ((pl * 4 + ch) * CS + (r ^ (2 * ch)));
  }
  auto issue_w = [&](int k0, f32x4 (&dst)[NWV]) {
#pragma unroll
    for (int i = 0; i < NWV; ++i) dst[i] = *reinterpret_cast<const f32x4*>(p.Wp + (wsrc[i] + (unsigned)k0));
  };
  auto store_w = [&](float* wb, const f32x4 (&src)[NWV]) {
#pragma unroll
    for (int i = 0; i < NWV; ++i) *reinterpret_cast<f32x4*>(wb + wdst[i]) = src[i];
  };

  // accumulators (C^T: batch row on the lane, 4 consecutive output features per register group) start at the bias
  const bool has_bias = p.bias != nullptr, has_pm = p.post_mul != nullptr;
  constexpr int FT = BN / 16;                  // 16-feature tiles per wave; each against 2 batch tiles of 16 rows
  f32x4 acc[FT][2];
#pragma unroll
  for (int ft = 0; ft < FT; ++ft) {
    const int col = n0 + ft * 16 + 4 * lg;
    f32x4 bv = zero4;
    if (has_bias) {
      if (p.bias_vec) {
        bv = *reinterpret_cast<const f32x4*>(p.bias + min(col, p.N - 4));
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) bv[j] = p.bias[min(col + j, p.N - 1)];
      }
    }
    acc[ft][0] = bv;
    acc[ft][1] = bv;
  }

#ifdef USF_STAMP
#define BSTAMP(v) unsigned long long v = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0)
#else
#define BSTAMP(v)
#endif
  BSTAMP(b0);
  // Pipeline: weights of slab s+1 are fetched during step 0 of slab s and stored behind its step 1; the
  // activation lines of slab s+2 are fetched during step 1 of slab s, right after the lines of slab s+1 went
  // through the scratch -- a full slab of latency cover; slab s+1's operand planes are split (VALU) in the
  // shadow of step 1's MFMAs.
  // SIDE: the blocks also write the operand planes they split -- every column block of a row panel sees the whole operand,
  // so block bn writes the slabs s with s % nbn == bn (a wave-uniform branch around six 16-byte stores per owned slab).
  // Rows >= M are not written (an offset beyond the buffer: dropped): the caller's buffer holds zeros there.
  // Measured on the 784 x 784 GEMM of the training step (65 536 rows, 404 us without the side output): all stores through
  // column block 0 + 53 us; dealt out, every block issuing every store (foreign slabs dropped by the bounds check) + 42 us;
  // this form: see r04_tuning_experiments.md section 7.
  typedef unsigned side_u32x4 __attribute__((ext_vector_type(4)));
  __amdgpu_buffer_rsrc_t side_rs = __builtin_amdgcn_make_buffer_rsrc(SIDE ? p.Apl : nullptr, 0, SIDE ? (int)p.plbytes : 0, 0x00020000);
  unsigned side_vo[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int row = row0 + 16 * b + lj;
    side_vo[b] = (SIDE && row < p.M) ? ((unsigned)row * p.ldpl + 8u * (unsigned)lg) * 2u : 0x80000000u;
  }
  auto side_store = [&](int k0, const bf16x8 (&pl)[2][3]) {
    if (!SIDE) return;
    if ((k0 >> 5) % p.nbn != bn) return;                                     // scalar
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        // (the whole offset in the vector register, soffset 0: with a scalar-register soffset the compiler's hazard model
        // sees no "store data overwritten right behind a 16-byte store" hazard and gfx950 has it -- measured: the last
        // dword of a store replaced by the next store's data in some lanes)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(side_u32x4, pl[b][q]), side_rs,
                                               (int)(side_vo[b] + ((unsigned)k0 + (unsigned)q * p.plstride) * 2u), 0, 0);
      }
  };
  bf16x8 pc[2][3], pn[2][3];
  f32x4 a_nxt[4], af[4];
  f32x4 wst[NWV];
  const int nslab = (p.K + BK - 1) / BK;
#pragma unroll
  for (int d = 0; d < D; ++d) {
    issue_w(min(d, nslab - 1) * BK, wst);
    store_w(wring + d * STG, wst);
  }
  issue_a(0, a_nxt);
  transpose_a(0, a_nxt, af);
  __builtin_amdgcn_sched_barrier(0);
  issue_a(BK, a_nxt);
  split3(af[0], af[1], pc[0][0], pc[0][1], pc[0][2]);
  split3(af[2], af[3], pc[1][0], pc[1][1], pc[1][2]);
  side_store(0, pc);
  __syncthreads();

  // half h of the slab's feature tiles: per tile 3 weight-plane fragments (lane (j, g): row 16 ft + j, k-chunk g)
  // against both batch tiles, smallest terms first; the two batch tiles alternate, so consecutive MFMAs are
  // independent
  auto compute_half = [&](const float* rb, int h) {
    const float* wl = rb + 4 * (lg * CS + (lj ^ (2 * lg)));
    constexpr int H0 = FT / 2;
    // feature tiles in PAIRS where the half has an even start: the six products of two tiles interleaved give four
    // independent accumulators in a row (acc[ft][0], acc[ft][1], acc[ft+1][0], acc[ft+1][1]) instead of two
#define USF_MM(FT_, W, P)                                                                           \
      acc[FT_][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W, pc[0][P], acc[FT_][0], 0, 0, 0);     \
      acc[FT_][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W, pc[1][P], acc[FT_][1], 0, 0, 0)
    int ft = h * H0;
#pragma unroll
    for (int pr = 0; pr < H0 / 2; ++pr, ft += 2) {
      const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(wl + 4 * (0 * 4 * CS + ft * 16));
      const bf16x8 a2 = *reinterpret_cast<const bf16x8*>(wl + 4 * (1 * 4 * CS + ft * 16));
      const bf16x8 a3 = *reinterpret_cast<const bf16x8*>(wl + 4 * (2 * 4 * CS + ft * 16));
      const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(wl + 4 * (0 * 4 * CS + (ft + 1) * 16));
      const bf16x8 b2 = *reinterpret_cast<const bf16x8*>(wl + 4 * (1 * 4 * CS + (ft + 1) * 16));
      const bf16x8 b3 = *reinterpret_cast<const bf16x8*>(wl + 4 * (2 * 4 * CS + (ft + 1) * 16));
      USF_MM(ft, a3, 0); USF_MM(ft + 1, b3, 0); USF_MM(ft, a2, 1); USF_MM(ft + 1, b2, 1);
      USF_MM(ft, a1, 2); USF_MM(ft + 1, b1, 2); USF_MM(ft, a2, 0); USF_MM(ft + 1, b2, 0);
      USF_MM(ft, a1, 1); USF_MM(ft + 1, b1, 1); USF_MM(ft, a1, 0); USF_MM(ft + 1, b1, 0);
    }
    if (H0 & 1) {
      const bf16x8 w1 = *reinterpret_cast<const bf16x8*>(wl + 4 * (0 * 4 * CS + ft * 16));
      const bf16x8 w2 = *reinterpret_cast<const bf16x8*>(wl + 4 * (1 * 4 * CS + ft * 16));
      const bf16x8 w3 = *reinterpret_cast<const bf16x8*>(wl + 4 * (2 * 4 * CS + ft * 16));
      USF_MM(ft, w3, 0); USF_MM(ft, w2, 1); USF_MM(ft, w1, 2); USF_MM(ft, w2, 0); USF_MM(ft, w1, 1); USF_MM(ft, w1, 0);
    }
#undef USF_MM
  };
  // Issue order pins (masks: 0x008 MFMA, 0x002 VALU, 0x020 VMEM read, 0x100 DS read, 0x200 DS write).
  // Weight fragments are read one tile ahead of the MFMAs that use them.
  //  * step 0 carries the next slab's weight loads, dealt over its tiles (issued as one burst behind the
  //    barrier the block's loads queue in the address unit for ~2000 cycles while no wave reaches an MFMA),
  //    the scratch round trip of the next activation slab and, once that has freed the registers, the
  //    activation loads of the slab after it (a full slab of latency cover);
  //  * step 1 carries the operand split of the next slab (VALU in the MFMA shadow).
  constexpr int HT = FT / 2;                                     // tiles per half
  constexpr int LPT0 = (NWV + HT - 1) / HT;
  constexpr int T_SW = (HT >= 4) ? 1 : 0;                        // tile that carries the scratch writes
  constexpr int T_SR = T_SW + 1;                                 // ... the scratch fragment reads
  constexpr int T_LA = (T_SR + 1 < HT) ? T_SR + 1 : T_SR;        // ... the activation loads
  constexpr int NLA = PRO ? 6 : 4;
  constexpr int VPT = (100 + HT - 1) / HT;
#define USF_PIN_HALF0()                                                                           \
  do {                                                                                            \
    __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);                                            \
    _Pragma("unroll") for (int f_ = 0; f_ < HT; ++f_) {                                           \
      __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);                                         \
      if (f_ + 2 < HT) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);                         \
      __builtin_amdgcn_sched_group_barrier(0x020, LPT0, 0);                                       \
      if (f_ == T_SW) __builtin_amdgcn_sched_group_barrier(0x200, 4, 0);                          \
      if (f_ == T_SR) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);                          \
      if (f_ == T_LA) __builtin_amdgcn_sched_group_barrier(0x020, NLA, 0);                        \
    }                                                                                             \
  } while (0)
#define USF_PIN_HALF1()                                                                           \
  do {                                                                                            \
    __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);                                            \
    _Pragma("unroll") for (int f_ = 0; f_ < HT; ++f_) {                                           \
      __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);                                         \
      if (f_ + 2 < HT) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);                         \
      __builtin_amdgcn_sched_group_barrier(0x002, VPT, 0);                                        \
    }                                                                                             \
  } while (0)

  BSTAMP(b1);
#ifdef USF_STAMP
  unsigned long long ph[5] = {0, 0, 0, 0, 0};
#endif
#if defined(USF_STAMP) && USF_STAMP >= 2     // in-loop stamps drain the LDS queue five times per slab: a separate build
#define LSTAMP(v) BSTAMP(v)
#define LACC(i, a, b) ph[i] += (b) - (a)
#else
#define LSTAMP(v)
#define LACC(i, a, b)
#endif
  for (int s = 0; s + 1 < nslab; ++s) {
    const float* rb = wring + (s % NB) * STG;
    float* wb = wring + ((s + D) % NB) * STG;
    const int kw = min(s + D, nslab - 1) * BK;
    const int k1 = (s + 1) * BK;
    LSTAMP(l0);
    issue_w(kw, wst);
    compute_half(rb, 0);
    transpose_a(k1, a_nxt, af);
    issue_a(k1 + BK, a_nxt);
    USF_PIN_HALF0();
    __builtin_amdgcn_sched_barrier(0);
    LSTAMP(l2);
    split3(af[0], af[1], pn[0][0], pn[0][1], pn[0][2]);
    split3(af[2], af[3], pn[1][0], pn[1][1], pn[1][2]);
    compute_half(rb, 1);
    USF_PIN_HALF1();
    __builtin_amdgcn_sched_barrier(0);
    LSTAMP(l3);
    store_w(wb, wst);
    side_store(k1, pn);
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) pc[q][pl] = pn[q][pl];
    __builtin_amdgcn_sched_barrier(0);
    LSTAMP(l4);
    if (D == 1 || ((s + 1) % D) == 0) __syncthreads();
    LSTAMP(l5);
    LACC(1, l0, l2); LACC(2, l2, l3); LACC(3, l3, l4); LACC(4, l4, l5);
  }
  BSTAMP(b2);
  // (K tail: the planes are zero-padded to 32, the fragments zero-selected)
  {
    const float* rb = wring + ((nslab - 1) % NB) * STG;
    compute_half(rb, 0);
    compute_half(rb, 1);
  }

  BSTAMP(b3);
  // ---- epilogue: transpose each 32x32 tile through the wave's LDS scratch -> whole-cache-line stores ----
  float* tw = scr;                   // same swizzled 32 x 32 image as the activation path, the other way round
  const int rr = lane >> 3, cc = 4 * (lane & 7);
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    // lane (j, g) of feature tile ft, batch tile b holds features 16 ft + 4 g + (0..3) of batch row 16 b + j
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int r = 16 * b + lj;
        *reinterpret_cast<f32x4*>(tw + r * 32 + 4 * ((4 * q + lg) ^ swz(r))) = acc[2 * tn + q][b];
      }
    // everything element-wise happens after the transpose, where a lane owns 4 consecutive features of a
    // row and the wave touches whole cache lines (addend / residual are read the same way)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = rr + 8 * i;
      f32x4 v = *reinterpret_cast<const f32x4*>(tw + r * 32 + 4 * ((lane & 7) ^ swz(r)));
      const int row = row0 + r;
      const int col = n0 + tn * 32 + cc;
      const int rowc = min(row, p.M - 1), colc = min(col, p.N - 4);
      if (p.act == USF_ACT_GATE) {
        const f32x4 h = *reinterpret_cast<const f32x4*>(p.addend + (int64_t)rowc * p.ldadd + colc);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = gate_apply(v[j], h[j], p.slope);
      } else {
        if (p.addend) v = v + *reinterpret_cast<const f32x4*>(p.addend + (int64_t)rowc * p.ldadd + colc);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = act_apply(v[j], p.act, p.slope);
      }
      if (p.residual) v = *reinterpret_cast<const f32x4*>(p.residual + (int64_t)rowc * p.ldr + colc) + p.res_sign * v;
      if (has_pm) v = v * *reinterpret_cast<const f32x4*>(p.post_mul + colc);
      float* dst = p.C + (int64_t)row * p.ldc + col;
      if (row < p.M) {
        if (col + 3 < p.N) {
          *reinterpret_cast<f32x4*>(dst) = v;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (col + j < p.N) dst[j] = v[j];
        }
      }
    }
  }
#ifdef USF_STAMP
  BSTAMP(b4);
  if (p.dbg && lane == 0) {
    unsigned long long* o = p.dbg + (size_t)((blockIdx.x % 1024) * WM + wave) * 8;
    o[0] = b1 - b0; o[1] = b2 - b1; o[2] = b3 - b2; o[3] = b4 - b3; o[4] = b4 - b0; o[5] = 1;
    unsigned long long* o2 = p.dbg + 8192 * 8 + (size_t)((blockIdx.x % 1024) * WM + wave) * 8;
    for (int i = 0; i < 5; ++i) o2[i] = ph[i];
  }
#endif
}

#ifdef USF_STAMP
unsigned long long* g_bdbg = nullptr;
#endif

template <int TN, int WM, int NB>
static int launch3(Lin3Args a, hipStream_t stream) {
  const bool pro = a.pre_div != nullptr || a.pre_sub != nullptr;
  constexpr int BM = WM * 32, BN = TN * 32;
  a.nbm = (a.M + BM - 1) / BM;
  a.nbn = (a.N + BN - 1) / BN;
  const int64_t grid = (((int64_t)a.nbm + 7) / 8) * 8 * a.nbn;
  if (grid > 0x7fffffffLL) { set_error("usf_linear_f32(bf16x3): grid too large"); return -3; }
  if (a.Apl) hipLaunchKernelGGL((linear_bf16x3_kernel<TN, WM, false, NB, true>), dim3((unsigned)grid), dim3(WM * 64), 0, stream, a);
  else if (pro) hipLaunchKernelGGL((linear_bf16x3_kernel<TN, WM, true, NB>), dim3((unsigned)grid), dim3(WM * 64), 0, stream, a);
  else hipLaunchKernelGGL((linear_bf16x3_kernel<TN, WM, false, NB>), dim3((unsigned)grid), dim3(WM * 64), 0, stream, a);
  return check_launch("usf_linear_f32(bf16x3)");
}

// true when this descriptor can take the split-precision kernel
bool linear_bf16x3_eligible(const usf_linear_desc* d) {
  return d->W_split != nullptr && (d->K & 7) == 0 && (d->N & 3) == 0 &&
         (!d->residual || (aligned16(d->residual) && (d->ldr & 3) == 0)) &&
         (!d->addend || (aligned16(d->addend) && (d->ldadd & 3) == 0)) && !(d->residual && d->addend) &&
         (!d->post_mul || aligned16(d->post_mul)) &&
         (!d->pre_div || aligned16(d->pre_div)) && (!d->pre_sub || aligned16(d->pre_sub)) &&
         (d->ldc & 3) == 0 && aligned16(d->C) && aligned16(d->W_split) && (d->ldw_split & 7) == 0 &&
         d->ldw_split >= ((d->K + 31) / 32) * 32 && d->M > 64 && d->N > 64 &&
         d->M * d->lda < (1LL << 31) && 3 * d->split_plane_stride < (1LL << 31);   // 32-bit element offsets
}

// which instantiation serves an [M, N] output: 3000 + 100 TN + 10 WM + NB (reported by usf_linear_variant)
int linear_bf16x3_variant(int M, int N) {
  const int pad160 = ((N + 159) / 160) * 160 - N;
  const int pad128 = ((N + 127) / 128) * 128 - N;
  // 8-wave blocks (256 rows) stage each weight slab once per 256 rows: 2 % faster in the flow than 4-wave
  // blocks at M = 65536; USFLOWS_AMD_TUNE=bf16x3_wm=4 / 8 forces the 4- / 8-wave tile (tuning aid)
  const long long wm_ = tuning("bf16x3_wm", 0);
  const int wm4 = wm_ == 4 ? 1 : (wm_ == 8 ? 2 : 0);
  // small batches are latency-bound by one block's serial K loop: narrow column blocks (64 wide) shorten the
  // per-slab MFMA chain 2.5x and put 2.5x more blocks on the chip
  if ((int64_t)((M + 127) / 128) * ((N + 159) / 160) < 256) return 3244;
  if (pad160 < pad128) {
    // Wave quantisation: 256-row blocks run one per CU, so M / 256 x 5 column blocks fill ceil(. / 256) rounds and a
    // mostly empty last round costs a full one (M = 32768: 2.5 rounds -> 3).  128-row blocks (two per CU) cut the
    // tail in half; they stage every weight slab twice per CU, so they only win when the last round is < ~70 % full
    // (measured, 33 affine launches: M = 16384 4.68 -> 3.97 ms, 32768 8.69 -> 6.71 ms; 24576 / 65536 stay on 256).
    bool big = M >= 2048 && wm4 != 1;
    if (big && wm4 == 0) {
      const double rounds = (double)(((int64_t)M + 255) / 256) * ((N + 159) / 160) / 256.0;
      const double frac = rounds - (double)(int64_t)rounds;
      if (rounds > 1.0 && frac > 0.0 && frac < 0.7) big = false;
    }
    return big ? 3584 : 3542;
  }
  return 3442;
}

int linear_bf16x3_dispatch(const usf_linear_desc* d, hipStream_t stream) {
  Lin3Args a;
  a.A = d->A; a.Wp = reinterpret_cast<const __bf16*>(d->W_split); a.bias = d->bias; a.post_mul = d->post_mul; a.C = d->C;
  a.pre_div = d->pre_div; a.pre_sub = d->pre_sub;
  a.residual = d->residual; a.addend = d->addend; a.ldr = d->ldr; a.ldadd = d->ldadd; a.res_sign = d->res_sign;
  a.lda = d->lda; a.ldc = d->ldc; a.ldwp = d->ldw_split; a.plane_stride = d->split_plane_stride;
  a.M = (int)d->M; a.N = (int)d->N; a.K = (int)d->K; a.nbm = a.nbn = 0;
  a.slope = d->slope; a.act = d->act;
  a.bias_vec = (d->bias && aligned16(d->bias) && (d->N & 3) == 0) ? 1 : 0;
  a.dbg = nullptr;
#ifdef USF_STAMP
  a.dbg = g_bdbg;
#endif
  a.Apl = nullptr; a.ldpl = a.plstride = a.plbytes = 0;
  if (d->A_planes_out) {                 // (linear_dispatch checked the geometry and that there is no prologue)
    a.Apl = reinterpret_cast<__bf16*>(d->A_planes_out);
    a.ldpl = (unsigned)d->ldp_out; a.plstride = (unsigned)d->planes_out_stride;
    a.plbytes = (unsigned)(3 * d->planes_out_stride * 2);
  }
  switch (linear_bf16x3_variant(a.M, a.N)) {
    case 3244: return launch3<2, 4, 4>(a, stream);
    case 3584: return launch3<5, 8, 4>(a, stream);
    case 3542: return launch3<5, 4, 2>(a, stream);
    default:   return launch3<4, 4, 2>(a, stream);
  }
}

}  // namespace usf
