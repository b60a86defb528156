// Split-precision (bf16x3) variant of the fused additive-coupling kernel: same structure as
// usf_coupling.hip -- hidden activations live in MFMA accumulators, only weights travel through LDS -- but
// every product runs on the bf16 matrix cores (v_mfma_f32_16x16x32_bf16) with both operands split three
// ways (x = x1 + x2 + x3 exactly, six MFMAs per fp32-equivalent product; see usf_linear_bf16x3.hip).
//
//  * 512-thread block = 8 waves x 16 batch rows; one block per CU (2 waves per SIMD).  A weight stage is
//    [256 x 32 k] (k-slab of W_in / W_h) or [32 n x 256 k] (n-tile of W_out), three bf16 planes = 48 KB,
//    double-buffered; the 8 waves share it, so a stage is staged once per 128 batch rows.
//  * accumulator -> operand: the 16x16 f32 accumulator has hidden unit 4*(lane>>4)+r of batch row lane&15
//    in register r.  Two consecutive tiles (hidden 0-15, 16-31 of a 32-k step) give each lane 8 values
//    whose k indices are {4g..4g+3, 16+4g..16+4g+3}; the hidden-layer weights are stored with that
//    k-permutation (pack time), so the split accumulators ARE the B-operand fragments of the next layer.
//  * output orientation as in the f32 kernel: a lane ends with 4 consecutive output features of its row
//    -> 16-byte residual load / store.
#include "usf_common.h"

namespace usf {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int C3_ROWS = 128;               // batch rows per block (8 waves x 16)
constexpr int C3_NT = 512;
constexpr int C3_HMAX = 256;
#ifndef C3_AHEAD
#define C3_AHEAD 3                         // weight fragments are read this many tiles ahead of their MFMAs
#endif

struct Cpl3Args {
  const float* z; float* out; int64_t ldz;
  int M, off_pass, n_pass, off_trans, n_trans, n_trans4;
  const __bf16* Win; int64_t ld_in, pl_in; const float* b_in;
  const __bf16* Whid[2]; int64_t ld_hid, pl_hid; const float* b_hid[2];
  const __bf16* Wout; int64_t ld_out, pl_out; const float* b_out;
  const float* ctx; const float* W_ctx; const float* b_ctx;
  float sign, slope; int act;
  unsigned long long* dbg;              // tuning builds only (USF_STAMP)
  float* hsave[3]; int64_t ld_hs;       // usf_coupling_desc::hidden_out, or NULLs
  const float* gate[3]; int64_t ld_gate; // act == USF_ACT_GATE: usf_coupling_desc::gate
};

__device__ __forceinline__ void c3_split3(const f32x4 x0, const f32x4 x1, bf16x8& p1, bf16x8& p2, bf16x8& p3) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float x = (j < 4) ? x0[j & 3] : x1[j & 3];
    const __bf16 h = (__bf16)x;
    const float r = x - (float)h;
    const __bf16 m = (__bf16)r;
    const float r2 = r - (float)m;
    p1[j] = h; p2[j] = m; p3[j] = (__bf16)r2;
  }
}

// six-term product, smallest terms first
#define C3_MFMA6(ACC, W1, W2, W3, A1, A2, A3)                                   \
  do {                                                                          \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W3, A1, ACC, 0, 0, 0);        \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W2, A2, ACC, 0, 0, 0);        \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W1, A3, ACC, 0, 0, 0);        \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W2, A1, ACC, 0, 0, 0);        \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W1, A2, ACC, 0, 0, 0);        \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W1, A1, ACC, 0, 0, 0);        \
  } while (0)

template <int NH, int T>
__global__ __launch_bounds__(C3_NT, 2) void coupling_bf16x3_kernel(const Cpl3Args p) {
  constexpr int HP = 16 * T;               // padded hidden width
  constexpr int KS = T / 2;                // 32-k steps over a hidden layer
  constexpr int SLOTS = 3 * 4 * HP;        // 16-B slots per stage (k-slab: 3 planes x 4 chunks x HP rows;
                                           //                       n-tile: 3 planes x (HP/8) chunks x 32 rows)
  constexpr int NST = SLOTS / C3_NT;       // float4 staged per thread per stage
  constexpr int NPP = NST / 3;             // ... per plane
  static_assert(SLOTS % (3 * C3_NT) == 0, "stage shape");
  constexpr int NC = HP / 8;               // 16-B chunks per n-tile row
  static_assert(NC % 8 == 0, "n-tile rows are dealt in whole 128-B lines");
  // two weight stages + per wave a 16 x 32 fp32 scratch (activation slabs in, output tiles out: global memory
  // is touched in whole 128-B lines, 8 lanes per row, and the scratch turns lines into MFMA fragments and back)
  __shared__ __attribute__((aligned(16))) float lds[2][SLOTS * 4];
  __shared__ __attribute__((aligned(16))) float cscr[8 * 512];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int lj = lane & 15;                // MFMA 16x16x32: lane = (row-in-tile j, k-group g)
  const int lg = lane >> 4;
  const int lr = lane >> 3, lc = lane & 7; // line-shaped access: lane = (row lr (+8), 16-B chunk lc)
  const int wrow0 = blockIdx.x * C3_ROWS + wave * 16;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  // scratch swizzle: chunk c of row r at position c ^ swz(r) (see usf_linear_bf16x3.hip)
  auto swz = [](int r) { const int h = r >> 1; return (((h >> 2) ^ (h >> 1)) & 1) | ((h & 1) << 1) | (((h >> 1) & 1) << 2); };
  float* const scr = cscr + wave * 512;

  // ---- weight stages -----------------------------------------------------------------------------------------
  // k-slab image: slot(plane q, chunk c, row r) = (q * 4 + c) * HP + (r ^ 2c); dealt row-major (4 consecutive
  // lanes = the 4 chunks = 64 contiguous bytes of a row, 16 lanes = 4 cache lines); the XOR keeps the stores
  // (8 lanes = 2 rows x 4 chunks per ds_write_b128 group) on 8 different 16-B bank groups while the fragment
  // reads (lane (j, g): row 16 ht + j of chunk g) stay on a chunk stride that is a multiple of 256 B.
  unsigned ksrc[NPP];                      // element offset of (row, chunk) inside a plane
  int kdst[NPP];                           // float offset of slot(0, c, r)
#pragma unroll
  for (int i = 0; i < NPP; ++i) {
    const int idx = tid + C3_NT * i;
    const int r = idx >> 2, c = idx & 3;
    kdst[i] = 4 * (c * HP + (r ^ (2 * c)));
    ksrc[i] = (unsigned)r;                 // multiplied by the layer's leading dimension at issue time
  }
  auto issue_k = [&](const __bf16* W, int64_t ld, int64_t pl, int k0, f32x4 (&st)[NST]) {
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int i = 0; i < NPP; ++i)
        st[q * NPP + i] = *reinterpret_cast<const f32x4*>(W + q * pl + (int64_t)ksrc[i] * ld + k0 + 8 * ((tid + C3_NT * i) & 3));
  };
  auto store_k = [&](int buf, const f32x4 (&st)[NST]) {
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int i = 0; i < NPP; ++i) *reinterpret_cast<f32x4*>(&lds[buf][q * 16 * HP + kdst[i]]) = st[q * NPP + i];
  };
  // n-tile image: slot(plane q, chunk c, row r) = (q * NC + c) * 32 + (r ^ 2 (c & 7)); dealt row-major (8
  // consecutive lanes = 8 chunks = one 128-B line of a row)
  auto n_rc = [&](int i, int& r, int& c) { const int idx = tid + C3_NT * i; r = idx / NC; c = idx % NC; };
  auto issue_n = [&](const __bf16* W, int64_t ld, int64_t pl, int n0, f32x4 (&st)[NST]) {
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int i = 0; i < NPP; ++i) {
        int r, c; n_rc(i, r, c);
        st[q * NPP + i] = *reinterpret_cast<const f32x4*>(W + q * pl + (int64_t)(n0 + r) * ld + 8 * c);
      }
  };
  auto store_n = [&](int buf, const f32x4 (&st)[NST]) {
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int i = 0; i < NPP; ++i) {
        int r, c; n_rc(i, r, c);
        *reinterpret_cast<f32x4*>(&lds[buf][4 * ((q * NC + c) * 32 + (r ^ (2 * (c & 7))))]) = st[q * NPP + i];
      }
  };

  const int nS1 = (p.n_pass + 31) / 32;
  const int nS3 = (p.n_trans + 31) / 32;

  // accumulators start at the layer bias
  f32x4 X1[T], X2[T];
#pragma unroll
  for (int t = 0; t < T; ++t) X1[t] = *reinterpret_cast<const f32x4*>(p.b_in + t * 16 + 4 * lg);

  // ---- conditioning half: lines of the wave's 16 rows, 2 loads per lane per slab ----
  f32x4 st[NST];
  f32x4 zl[2];
  unsigned zoff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) zoff[i] = (unsigned)min(wrow0 + lr + 8 * i, p.M - 1) * (unsigned)p.ldz + (unsigned)p.off_pass;
  auto issue_z = [&](int k0, f32x4 (&dst)[2]) {
    const unsigned kc = (unsigned)min(k0 + 4 * lc, p.n_pass - 4);
#pragma unroll
    for (int i = 0; i < 2; ++i) dst[i] = *reinterpret_cast<const f32x4*>(p.z + (zoff[i] + kc));
  };
  // lines -> scratch -> operand fragment (lane (j, g): row j, k = 8 g + (0..7)) -> three bf16 planes
  auto z_planes = [&](int k0, const f32x4 (&src)[2], bf16x8 (&pl)[3]) {
    const bool live = k0 + 4 * lc < p.n_pass;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = lr + 8 * i;
      *reinterpret_cast<f32x4*>(scr + r * 32 + 4 * (lc ^ swz(r))) = live ? src[i] : zero4;
    }
    const f32x4 f0 = *reinterpret_cast<const f32x4*>(scr + lj * 32 + 4 * ((2 * lg) ^ swz(lj)));
    const f32x4 f1 = *reinterpret_cast<const f32x4*>(scr + lj * 32 + 4 * ((2 * lg + 1) ^ swz(lj)));
    c3_split3(f0, f1, pl[0], pl[1], pl[2]);
  };

  // one 32-k step over all hidden tiles: X[ht] += W(ht) . B, B given as 3 planes; W fragments from a k-slab
  auto mfma_slab = [&](int buf, f32x4 (&X)[T], const bf16x8 (&b)[3]) {
    const float* wl = &lds[buf][4 * (lg * HP + (lj ^ (2 * lg)))];
#pragma unroll
    for (int ht = 0; ht < T; ++ht) {
      const bf16x8 w1 = *reinterpret_cast<const bf16x8*>(wl + 4 * ((0 * 4) * HP + ht * 16));
      const bf16x8 w2 = *reinterpret_cast<const bf16x8*>(wl + 4 * ((1 * 4) * HP + ht * 16));
      const bf16x8 w3 = *reinterpret_cast<const bf16x8*>(wl + 4 * ((2 * 4) * HP + ht * 16));
      // issue priority alternates tile by tile: on a SIMD the older wave otherwise wins the matrix pipe whenever
      // both are ready, reaches the barrier ~1100 cycles early and leaves the younger one to run alone; with the
      // toggle whichever wave is a tile behind outranks the other, so the two advance in step (phases 1 and 2:
      // -6.5 % on the layer in the flow; no gain in the output phase or in the linear kernel, not used there)
      if (ht & 1) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(1);
      C3_MFMA6(X[ht], w1, w2, w3, b[0], b[1], b[2]);
    }
  };
  // Issue order pins (0x008 MFMA, 0x002 VALU, 0x020 VMEM read, 0x100 DS read, 0x200 DS write): weight fragments
  // C3_AHEAD tiles ahead of their MFMAs (on each SIMD the older wave wins the matrix pipe and reaches the barrier
  // ~1100 cycles early; the younger one then runs alone and has to cover the LDS latency by itself); the next stage's global loads dealt over the first tiles (one burst behind
  // the barrier would queue in the address unit while no wave reaches an MFMA); NSW scratch writes + NSR scratch
  // reads early, then VPT VALU per tile (the next operand's split, in the MFMA shadow)
#define C3_PIN_SLAB(NLD, NSW, NSR, VPT)                                                           \
  do {                                                                                            \
    __builtin_amdgcn_sched_group_barrier(0x100, 3 * C3_AHEAD, 0);                                 \
    _Pragma("unroll") for (int ht_ = 0; ht_ < T; ++ht_) {                                         \
      __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);                                          \
      if (ht_ + C3_AHEAD < T) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);                  \
      if (ht_ < (NLD)) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                         \
      if (ht_ == 1 && (NSW) > 0) __builtin_amdgcn_sched_group_barrier(0x200, (NSW), 0);           \
      if (ht_ == 2 && (NSR) > 0) __builtin_amdgcn_sched_group_barrier(0x100, (NSR), 0);           \
      if (ht_ >= 3 && (VPT) > 0) __builtin_amdgcn_sched_group_barrier(0x002, (VPT), 0);           \
    }                                                                                             \
  } while (0)

  auto issue_after_input = [&](f32x4 (&s_)[NST]) {
    if (NH >= 2) issue_k(p.Whid[0], p.ld_hid, p.pl_hid, 0, s_); else issue_n(p.Wout, p.ld_out, p.pl_out, 0, s_);
  };
  auto store_after_input = [&](int buf, const f32x4 (&s_)[NST]) {
    if (NH >= 2) store_k(buf, s_); else store_n(buf, s_);
  };

#ifdef USF_STAMP
#define C3STAMP(v) unsigned long long v = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0)
#else
#define C3STAMP(v)
#endif
  C3STAMP(t0);
  int g = 0;
  bf16x8 zp[3], zn[3];
  issue_k(p.Win, p.ld_in, p.pl_in, 0, st);
  issue_z(0, zl);
  z_planes(0, zl, zp);
  __builtin_amdgcn_sched_barrier(0);
  issue_z(32, zl);                         // (clamped when the conditioning half is a single slab)
  store_k(0, st);
  __syncthreads();

  // ================= phase 1: X1[h][row] += W_in[h][k] * z[row][k] ============================
  // slab s: multiplies with planes zp; the lines of slab s+1 (fetched a slab ago) go through the scratch and are
  // split in the MFMA shadow; the lines of slab s+2 and the weights of slab s+1 are fetched
  for (int s = 0; s + 1 < nS1; ++s, ++g) {
    const int buf = g & 1;
    issue_k(p.Win, p.ld_in, p.pl_in, (s + 1) * 32, st);
    mfma_slab(buf, X1, zp);
    z_planes((s + 1) * 32, zl, zn);
    issue_z((s + 2) * 32, zl);
    C3_PIN_SLAB(NST + 2, 2, 2, 6);
    __builtin_amdgcn_sched_barrier(0);
    store_k(buf ^ 1, st);
#pragma unroll
    for (int q = 0; q < 3; ++q) zp[q] = zn[q];
    __syncthreads();
  }
  {
    const int buf = g & 1;
    issue_after_input(st);
    mfma_slab(buf, X1, zp);
    C3_PIN_SLAB(NST, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    store_after_input(buf ^ 1, st);
    __syncthreads();
    ++g;
  }

  C3STAMP(t1);
  const int rowc = min(wrow0 + lj, p.M - 1);
  auto ctx_act = [&](f32x4 (&X)[T], bool with_ctx, int layer) {
    if (p.act == USF_ACT_GATE) {
      // the conditioner's backward pass: the (Leaky)ReLU backward from the forward's saved output of the layer this
      // gradient belongs to (lane (j, g): units 16 ht + 4 g .. + 3 of row j -- the layout save_hidden writes)
      const float* G = p.gate[layer] + (int64_t)rowc * p.ld_gate + 4 * lg;
#pragma unroll
      for (int ht = 0; ht < T; ++ht) {
        const f32x4 hv = *reinterpret_cast<const f32x4*>(G + 16 * ht);
#pragma unroll
        for (int t = 0; t < 4; ++t) X[ht][t] = gate_apply(X[ht][t], hv[t], p.slope);
      }
      return;
    }
    const float cv = with_ctx ? p.ctx[rowc] : 0.f;
#pragma unroll
    for (int ht = 0; ht < T; ++ht) {
      f32x4 wc = zero4, bc = zero4;
      if (with_ctx) {
        wc = *reinterpret_cast<const f32x4*>(p.W_ctx + ht * 16 + 4 * lg);
        bc = *reinterpret_cast<const f32x4*>(p.b_ctx + ht * 16 + 4 * lg);
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float v = X[ht][t];
        if (with_ctx) v = v + (cv * wc[t] + bc[t]);
        X[ht][t] = act_apply(v, p.act, p.slope);
      }
    }
  };
  ctx_act(X1, p.ctx != nullptr, 0);
  // training: the hidden activations go to HBM for the backward pass (lane (j, g) holds units 16 ht + 4 g .. + 3 of row j)
  auto save_hidden = [&](const f32x4 (&X)[T], float* H) {
    if (H == nullptr || wrow0 + lj >= p.M) return;
    float* d = H + (int64_t)(wrow0 + lj) * p.ld_hs + 4 * lg;
#pragma unroll
    for (int ht = 0; ht < T; ++ht) *reinterpret_cast<f32x4*>(d + 16 * ht) = X[ht];
  };
  save_hidden(X1, p.hsave[0]);

  // ================= phase 2: Xout[h2][row] += W_h[h2][h1'] * Xin[h1'][row] ====================
  auto hidden_layer = [&](f32x4 (&Xin)[T], f32x4 (&Xout)[T], int l) {
    bf16x8 xc[3], xn[3];
    c3_split3(Xin[0], Xin[1], xc[0], xc[1], xc[2]);            // k order {4g.., 16+4g..}: the weights' pack-time order
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int buf = g & 1;
      const bool last = (ks + 1 == KS);
      const bool next_is_hidden = (l + 2 < NH);
      if (!last) issue_k(p.Whid[l], p.ld_hid, p.pl_hid, (ks + 1) * 32, st);
      else if (next_is_hidden) issue_k(p.Whid[l + 1], p.ld_hid, p.pl_hid, 0, st);
      else issue_n(p.Wout, p.ld_out, p.pl_out, 0, st);
      mfma_slab(buf, Xout, xc);
      if (!last) c3_split3(Xin[2 * ks + 2], Xin[2 * ks + 3], xn[0], xn[1], xn[2]);
      C3_PIN_SLAB(NST, 0, 0, 5);
      __builtin_amdgcn_sched_barrier(0);
      if (!last || next_is_hidden) store_k(buf ^ 1, st); else store_n(buf ^ 1, st);
      if (!last) {
#pragma unroll
        for (int q = 0; q < 3; ++q) xc[q] = xn[q];
      }
      __syncthreads();
      ++g;
    }
    ctx_act(Xout, false, l + 1);
  };
  if (NH >= 2) {
#pragma unroll
    for (int t = 0; t < T; ++t) X2[t] = *reinterpret_cast<const f32x4*>(p.b_hid[0] + t * 16 + 4 * lg);
    hidden_layer(X1, X2, 0);
    save_hidden(X2, p.hsave[1]);
  }
  if (NH >= 3) {
#pragma unroll
    for (int t = 0; t < T; ++t) X1[t] = *reinterpret_cast<const f32x4*>(p.b_hid[1] + t * 16 + 4 * lg);
    hidden_layer(X2, X1, 1);
    save_hidden(X1, p.hsave[2]);
  }

  C3STAMP(t2);
  // ================= phase 3: out[row][n] = z[row][n] + sign * (b_out[n] + sum_h X[h][row] W_out[n][h']) =====
  auto output_layer = [&](f32x4 (&X)[T]) {
    // split the final hidden activations once: KS steps x 3 planes
    bf16x8 xp[KS][3];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) c3_split3(X[2 * ks], X[2 * ks + 1], xp[ks][0], xp[ks][1], xp[ks][2]);
    // the transformed half is read and written in whole lines: lane (lr (+8), lc) <-> 4 features of one row
    unsigned ooff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) ooff[i] = (unsigned)min(wrow0 + lr + 8 * i, p.M - 1) * (unsigned)p.ldz + (unsigned)p.off_trans;
    // (the tile's residual lines and its slice of the output bias are fetched one tile ahead; the bias joins
    // behind the MFMAs, so no tile starts with a load it has to wait for)
    f32x4 res[2], bo;
    auto issue_res = [&](int nt, f32x4 (&dst)[2]) {
      const unsigned cc = (unsigned)min(nt * 32 + 4 * lc, p.n_trans4 - 4);
#pragma unroll
      for (int i = 0; i < 2; ++i) dst[i] = *reinterpret_cast<const f32x4*>(p.z + (ooff[i] + cc));
      bo = *reinterpret_cast<const f32x4*>(p.b_out + nt * 32 + 4 * lc);   // b_out is padded to 32 * nS3
    };
    issue_res(0, res);
    for (int nt = 0; nt < nS3; ++nt, ++g) {
      const int buf = g & 1;
      issue_n(p.Wout, p.ld_out, p.pl_out, min(nt + 1, nS3 - 1) * 32, st);
      f32x4 acc[2] = {zero4, zero4};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int x = 2 * ((4 * ks + lg) & 7);                 // row swizzle of chunk 4 ks + g
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const float* wf = &lds[buf][4 * ((4 * ks + lg) * 32 + u * 16 + (lj ^ x))];
          const bf16x8 w1 = *reinterpret_cast<const bf16x8*>(wf + 4 * (0 * NC * 32));
          const bf16x8 w2 = *reinterpret_cast<const bf16x8*>(wf + 4 * (1 * NC * 32));
          const bf16x8 w3 = *reinterpret_cast<const bf16x8*>(wf + 4 * (2 * NC * 32));
          C3_MFMA6(acc[u], w1, w2, w3, xp[ks][0], xp[ks][1], xp[ks][2]);
        }
      }
      __builtin_amdgcn_sched_group_barrier(0x100, 3 * C3_AHEAD, 0);
#pragma unroll
      for (int i = 0; i < 2 * KS; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
        if (i + C3_AHEAD < 2 * KS) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
        if (i < NST + 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      store_n(buf ^ 1, st);
      // accumulator tiles -> scratch -> lines: lane (j, g) of tile u holds features 16 u + 4 g + (0..3) of row j
#pragma unroll
      for (int u = 0; u < 2; ++u) *reinterpret_cast<f32x4*>(scr + lj * 32 + 4 * ((4 * u + lg) ^ swz(lj))) = acc[u];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int r = lr + 8 * i;
        const f32x4 a = *reinterpret_cast<const f32x4*>(scr + r * 32 + 4 * (lc ^ swz(r)));
        const f32x4 v = res[i] + p.sign * (a + bo);
        const int row = wrow0 + r, col = nt * 32 + 4 * lc;
        float* dst = p.out + (int64_t)row * p.ldz + p.off_trans + col;
        if (row < p.M) {
          if (col + 3 < p.n_trans) {
            *reinterpret_cast<f32x4*>(dst) = v;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (col + e < p.n_trans) dst[e] = v[e];
          }
        }
      }
      issue_res(min(nt + 1, nS3 - 1), res);
      __syncthreads();
    }
  };
  if (NH == 2) output_layer(X2); else output_layer(X1);
#ifdef USF_STAMP
  C3STAMP(t3);
  if (p.dbg && lane == 0) {
    unsigned long long* o = p.dbg + (size_t)((blockIdx.x % 512) * 8 + wave) * 8;
    o[0] = t1 - t0; o[1] = t2 - t1; o[2] = t3 - t2; o[3] = t3 - t0; o[5] = 1;
  }
#endif
}

#ifdef USF_STAMP
unsigned long long* g_c3dbg = nullptr;
#endif

bool coupling_bf16x3_eligible(const usf_coupling_desc* d) {
  if (!d->split_in || !d->split_out || d->n_hidden < 1 || d->n_hidden > 3) return false;
  for (int i = 0; i + 1 < d->n_hidden; ++i)
    if (!d->split_hid[i]) return false;
  int hmax = 0;
  for (int i = 0; i < d->n_hidden; ++i) hmax = d->hidden[i] > hmax ? d->hidden[i] : hmax;
  return hmax > 128 && hmax <= C3_HMAX && d->M >= 1024;     // instantiated for the 256-wide tile set only
}

int coupling_bf16x3_dispatch(const usf_coupling_desc* d, hipStream_t stream) {
  Cpl3Args a;
  a.z = d->z; a.out = d->out; a.ldz = d->ldz;
  a.M = (int)d->M; a.off_pass = (int)d->off_pass; a.n_pass = (int)d->n_pass; a.off_trans = (int)d->off_trans;
  a.n_trans = (int)d->n_trans; a.n_trans4 = (int)((d->n_trans + 3) / 4 * 4);
  a.Win = reinterpret_cast<const __bf16*>(d->split_in); a.ld_in = d->split_in_ld; a.pl_in = d->split_in_plane; a.b_in = d->b_in;
  for (int i = 0; i < 2; ++i) {
    const bool used = i + 1 < d->n_hidden;
    a.Whid[i] = reinterpret_cast<const __bf16*>(used ? d->split_hid[i] : d->split_in);
    a.b_hid[i] = used ? d->b_hid[i] : d->b_in;
  }
  a.ld_hid = d->split_hid_ld; a.pl_hid = d->split_hid_plane;
  a.Wout = reinterpret_cast<const __bf16*>(d->split_out); a.ld_out = d->split_out_ld; a.pl_out = d->split_out_plane; a.b_out = d->b_out;
  a.ctx = d->context; a.W_ctx = d->W_ctx; a.b_ctx = d->b_ctx;
  a.sign = d->sign; a.slope = d->slope; a.act = d->act;
  a.dbg = nullptr;
#ifdef USF_STAMP
  a.dbg = g_c3dbg;
#endif
  for (int i = 0; i < 3; ++i) a.hsave[i] = (i < d->n_hidden) ? d->hidden_out[i] : nullptr;
  for (int i = 0; i < 3; ++i) a.gate[i] = (i < d->n_hidden) ? d->gate[i] : nullptr;
  a.ld_gate = d->ld_gate;
  if (d->act == USF_ACT_GATE) {
    for (int i = 0; i < d->n_hidden; ++i)
      if (!d->gate[i] || !aligned16(d->gate[i])) { set_error("usf_coupling_additive_f32(bf16x3): USF_ACT_GATE needs gate[l] for every hidden layer (16-byte aligned)"); return -2; }
    if (d->ld_gate < C3_HMAX || (d->ld_gate & 3) || d->context) {
      set_error("usf_coupling_additive_f32(bf16x3): USF_ACT_GATE needs ld_gate >= 256, a multiple of 4, and no context");
      return -2;
    }
  }
  a.ld_hs = d->ld_hidden_out;
  {
    // all n_hidden slots or none (a NULL slot would be skipped silently and the caller would train on stale activations)
    int set = 0;
    for (int i = 0; i < d->n_hidden; ++i) set += d->hidden_out[i] != nullptr;
    if (set != 0 && set != d->n_hidden) {
      set_error("usf_coupling_additive_f32(bf16x3): hidden_out: either all n_hidden pointers or none");
      return -2;
    }
    if (set && (d->ld_hidden_out < C3_HMAX || (d->ld_hidden_out & 3))) {
      set_error("usf_coupling_additive_f32(bf16x3): hidden_out needs ld_hidden_out >= 256, a multiple of 4");
      return -2;
    }
    for (int i = 0; i < d->n_hidden; ++i)
      if (d->hidden_out[i] && !aligned16(d->hidden_out[i])) {
        set_error("usf_coupling_additive_f32(bf16x3): hidden_out[%d] must be 16-byte aligned", i);
        return -2;
      }
  }
  const int64_t kp = ((d->n_pass + 31) / 32) * 32;
  if (d->split_in_ld < kp || d->split_hid_ld < C3_HMAX * (d->n_hidden > 1) || d->split_out_ld < C3_HMAX || (d->split_in_ld & 7) ||
      (d->split_out_ld & 7) || !aligned16(d->split_in) || !aligned16(d->split_out)) {
    set_error("usf_coupling_additive_f32(bf16x3): split-plane padding contract violated");
    return -2;
  }
  const dim3 grid((unsigned)((d->M + C3_ROWS - 1) / C3_ROWS)), block(C3_NT);
  switch (d->n_hidden) {
    case 1: hipLaunchKernelGGL((coupling_bf16x3_kernel<1, 16>), grid, block, 0, stream, a); break;
    case 2: hipLaunchKernelGGL((coupling_bf16x3_kernel<2, 16>), grid, block, 0, stream, a); break;
    default: hipLaunchKernelGGL((coupling_bf16x3_kernel<3, 16>), grid, block, 0, stream, a); break;
  }
  return check_launch("usf_coupling_additive_f32(bf16x3)");
}

}  // namespace usf
