// Fused additive coupling layer ON PLANES (usf_coupling_planes): MaskedCoupling.forward / backward
// (transforms.py:277-306) with its whole conditioner MLP (networks.py:739-751) in one launch, the activation buffer z
// being a planes buffer (include/usflows_hip.h: pre-split bf16x3 / fp16x2 planes in MFMA-operand order).
//
// Same dataflow as usf_coupling_bf16x3.hip -- a wave owns 16 batch rows (= ONE row panel of the planes buffer) for the
// whole layer, hidden activations live in MFMA accumulators and become the next layer's B operand by a lane-local
// split, only weights travel through LDS -- minus everything that kernel does to get operands: the conditioning
// features arrive as ONE 16-byte load per lane, plane and 32-feature block (no line-shaped loads, no LDS scratch, no
// split in the K loop), and the transformed features are read (residual), updated and rewritten as planes in place,
// lane-locally (two neighbouring accumulator tiles are a lane's 8 slots of a chunk line).
//   * 512-thread block = 8 waves x 16 rows; a weight stage is a [256 x 32 k] k-slab (first / hidden layers) or a
//     [32 n x 256 k] n-tile (output layer) of NPL planes, double-buffered; hidden widths are padded to 256.
//   * weights: the planes images of the planes pipeline (K axis in slot order; rows beyond the real width zero).
//   * NPL = 3: bf16, six MFMAs per product; NPL = 2: fp16, three (range guard as in usf_planes.hip).
#include <stdlib.h>

#include "usf_common.h"

namespace usf {

typedef __bf16 cp_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 cp_f16x8 __attribute__((ext_vector_type(8)));
#define USF_CP_F16_GUARD 65000.0f

template <int NPL> struct CPlanes;
template <> struct CPlanes<3> {
  typedef cp_bf16x8 vec;
  static __device__ __forceinline__ f32x4 mfma(vec a, vec b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ void split(const f32x4 x0, const f32x4 x1, vec (&o)[3]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float x = (j < 4) ? x0[j & 3] : x1[j & 3];
      const __bf16 h = (__bf16)x;
      const float r = x - (float)h;
      const __bf16 m = (__bf16)r;
      o[0][j] = h; o[1][j] = m; o[2][j] = (__bf16)(r - (float)m);
    }
  }
  // one value into slot j of a chunk line's planes
  static __device__ __forceinline__ void split1(const float x, const int j, vec (&o)[3]) {
    const __bf16 h = (__bf16)x;
    const float r = x - (float)h;
    const __bf16 m = (__bf16)r;
    o[0][j] = h; o[1][j] = m; o[2][j] = (__bf16)(r - (float)m);
  }
  // six-term product, smallest terms first
  static __device__ __forceinline__ void mm(f32x4& acc, const vec (&w)[3], const vec (&a)[3]) {
    acc = mfma(w[2], a[0], acc); acc = mfma(w[1], a[1], acc); acc = mfma(w[0], a[2], acc);
    acc = mfma(w[1], a[0], acc); acc = mfma(w[0], a[1], acc); acc = mfma(w[0], a[0], acc);
  }
};
template <> struct CPlanes<2> {
  typedef cp_f16x8 vec;
  static __device__ __forceinline__ f32x4 mfma(vec a, vec b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ void split(const f32x4 x0, const f32x4 x1, vec (&o)[2]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float x = (j < 4) ? x0[j & 3] : x1[j & 3];
      const _Float16 h = (_Float16)x;
      o[0][j] = h; o[1][j] = (_Float16)(x - (float)h);
    }
  }
  static __device__ __forceinline__ void split1(const float x, const int j, vec (&o)[2]) {
    const _Float16 h = (_Float16)x;
    o[0][j] = h; o[1][j] = (_Float16)(x - (float)h);
  }
  static __device__ __forceinline__ void mm(f32x4& acc, const vec (&w)[2], const vec (&a)[2]) {
    acc = mfma(w[1], a[0], acc); acc = mfma(w[0], a[1], acc); acc = mfma(w[0], a[0], acc);
  }
};

struct CplPArgs {
  char* z; int z_nkb, npanels, M;
  int kb_p0, nk_p, kb_t0, nk_t;
  const char* Win; int64_t ld_in, pl_in; const float* b_in;
  const char* Whid[2]; int64_t ld_hid, pl_hid; const float* b_hid[2];
  const char* Wout; int64_t ld_out, pl_out; const float* b_out;
  float sign, slope; int act;
  int32_t* range_flag;
  unsigned long long* dbg;               // tuning builds (-DUSF_STAMP) only
  // training (ABI 33; bf16x3 only): planes buffers of 8 blocks per panel (hidden width 256).  hout[l]: receives hidden
  // layer l's activations (MODE 1) resp. the gradients at its pre-activations (MODE 2); gate[l]: MODE 2, the saved
  // activations whose sign gates layer l (leaky_relu_backward from the saved output; only plane 0 is read)
  char* hout[2]; const char* gate[2];
};

#ifdef USF_STAMP
#define CSTAMP(v) unsigned long long v = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0)
#else
#define CSTAMP(v)
#endif

// MODE 0: inference.  MODE 1 (training forward): the lane-local splits of the hidden activations -- the next layer's B
// operands -- are also stored as planes (hout[l]: the operands of the conditioner's weight gradients and the gates of the
// backward launch).  MODE 2 (training backward: the launch runs the conditioner's transposed chain on the gradient
// buffer, usf_coupling_planes_desc::gate): the activation is leaky_relu_backward from the saved activations gate[l], and
// hout[l] receives the gated values (the gradients at the pre-activations).
template <int NPL, int NH, int MODE = 0>
__global__ __launch_bounds__(512, 2) void coupling_planes_kernel(const CplPArgs p) {
  typedef CPlanes<NPL> PT;
  typedef typename PT::vec vec8;
  constexpr int T = 16;                     // hidden tiles (padded hidden width 256)
  constexpr int HP = 16 * T;
  constexpr int KS = T / 2;                 // 32-k steps over a hidden layer
  constexpr int NT = 512;
  constexpr int SLOTS = NPL * 4 * HP;       // 16-B slots per stage
  constexpr int NST = SLOTS / NT;           // float4 staged per thread per stage
  constexpr int NPP = NST / NPL;            // ... per plane
  constexpr int NC = HP / 8;                // 16-B chunks per n-tile row
  constexpr int NPR = (NPL == 3) ? 6 : 3;   // MFMAs per (tile, operand) product
#ifndef USF_CP_AHEAD
#define USF_CP_AHEAD 3
#endif
  constexpr int AHEAD = USF_CP_AHEAD;       // weight fragments are read this many tiles ahead of their MFMAs
  __shared__ __attribute__((aligned(16))) float lds[2][SLOTS * 4];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int lj = lane & 15, lg = lane >> 4;
  const int panel = blockIdx.x * 8 + wave;                       // this wave's 16 rows
  const int panc = min(panel, p.npanels - 1);
  const bool live = panel < p.npanels;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  constexpr unsigned CHB = NPL * 1024u;
  const size_t zbase = ((size_t)panc * p.z_nkb) * CHB + (size_t)lane * 16;

  // ---- weight stages (images as in usf_coupling_bf16x3.hip; offsets in bytes, 2-byte elements) ----
  int kdst[NPP];
  unsigned krow[NPP];
#pragma unroll
  for (int i = 0; i < NPP; ++i) {
    const int idx = tid + NT * i;
    const int r = idx >> 2, c = idx & 3;
    kdst[i] = 4 * (c * HP + (r ^ (2 * c)));
    krow[i] = (unsigned)r;
  }
  auto issue_k = [&](const char* W, int64_t ld, int64_t pl, int k0, f32x4 (&st)[NST]) {
#pragma unroll
    for (int q = 0; q < NPL; ++q)
#pragma unroll
      for (int i = 0; i < NPP; ++i)
        st[q * NPP + i] = *reinterpret_cast<const f32x4*>(W + 2 * (q * pl + (int64_t)krow[i] * ld + k0 + 8 * ((tid + NT * i) & 3)));
  };
  auto store_k = [&](int buf, const f32x4 (&st)[NST]) {
#pragma unroll
    for (int q = 0; q < NPL; ++q)
#pragma unroll
      for (int i = 0; i < NPP; ++i) *reinterpret_cast<f32x4*>(&lds[buf][q * 16 * HP + kdst[i]]) = st[q * NPP + i];
  };
  auto n_rc = [&](int i, int& r, int& c) { const int idx = tid + NT * i; r = idx / NC; c = idx % NC; };
  auto issue_n = [&](const char* W, int64_t ld, int64_t pl, int n0, f32x4 (&st)[NST]) {
#pragma unroll
    for (int q = 0; q < NPL; ++q)
#pragma unroll
      for (int i = 0; i < NPP; ++i) {
        int r, c; n_rc(i, r, c);
        st[q * NPP + i] = *reinterpret_cast<const f32x4*>(W + 2 * (q * pl + (int64_t)(n0 + r) * ld + 8 * c));
      }
  };
  auto store_n = [&](int buf, const f32x4 (&st)[NST]) {
#pragma unroll
    for (int q = 0; q < NPL; ++q)
#pragma unroll
      for (int i = 0; i < NPP; ++i) {
        int r, c; n_rc(i, r, c);
        *reinterpret_cast<f32x4*>(&lds[buf][4 * ((q * NC + c) * 32 + (r ^ (2 * (c & 7))))]) = st[q * NPP + i];
      }
  };

  // accumulators start at the layer bias
  f32x4 X1[T], X2[T];
#pragma unroll
  for (int t = 0; t < T; ++t) X1[t] = *reinterpret_cast<const f32x4*>(p.b_in + t * 16 + 4 * lg);

  // conditioning features: the B operand straight from the planes
  auto load_z = [&](int kb, vec8 (&dst)[NPL]) {
#pragma unroll
    for (int q = 0; q < NPL; ++q)
      dst[q] = *reinterpret_cast<const vec8*>(p.z + zbase + (size_t)(p.kb_p0 + kb) * CHB + q * 1024);
  };

  // one 32-k step over all hidden tiles: X[ht] += W(ht) . B; W fragments from a k-slab
  auto mfma_slab = [&](int buf, f32x4 (&X)[T], const vec8 (&b)[NPL]) {
    const float* wl = &lds[buf][4 * (lg * HP + (lj ^ (2 * lg)))];
#pragma unroll
    for (int ht = 0; ht < T; ++ht) {
      vec8 w[NPL];
#pragma unroll
      for (int q = 0; q < NPL; ++q) w[q] = *reinterpret_cast<const vec8*>(wl + 4 * ((q * 4) * HP + ht * 16));
      // issue priority alternates tile by tile so that the two waves of a SIMD advance in step (usf_coupling_bf16x3.hip)
#ifndef USF_CP_NOPRIO
      if (ht & 1) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(1);
#endif
      PT::mm(X[ht], w, b);
    }
  };
#define CP_PIN_SLAB(NLD, VPT)                                                                     \
  do {                                                                                            \
    __builtin_amdgcn_sched_group_barrier(0x100, NPL * AHEAD, 0);                                  \
    _Pragma("unroll") for (int ht_ = 0; ht_ < T; ++ht_) {                                         \
      __builtin_amdgcn_sched_group_barrier(0x008, NPR, 0);                                        \
      if (ht_ + AHEAD < T) __builtin_amdgcn_sched_group_barrier(0x100, NPL, 0);                   \
      if (ht_ < (NLD)) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                         \
      if (ht_ >= 3 && (VPT) > 0) __builtin_amdgcn_sched_group_barrier(0x002, (VPT), 0);           \
    }                                                                                             \
  } while (0)

  bool bad = false;
  auto guard = [&](const f32x4 v) {
    if (NPL == 2) {
#pragma unroll
      for (int j = 0; j < 4; ++j) bad = bad || !(fabsf(v[j]) < USF_CP_F16_GUARD);
    }
  };

  CSTAMP(c0);
  int g = 0;
  f32x4 st[NST];
  vec8 zp[NPL], zn[NPL];
  issue_k(p.Win, p.ld_in, p.pl_in, 0, st);
  load_z(0, zp);
  store_k(0, st);
  __syncthreads();

  // ================= phase 1: X1[h][row] += W_in[h][k] * z[row][k] over the conditioning blocks ============
  for (int s = 0; s + 1 < p.nk_p; ++s, ++g) {
    const int buf = g & 1;
    issue_k(p.Win, p.ld_in, p.pl_in, (s + 1) * 32, st);
    load_z(s + 1, zn);
    mfma_slab(buf, X1, zp);
    CP_PIN_SLAB(NST + NPL, 0);
    __builtin_amdgcn_sched_barrier(0);
    store_k(buf ^ 1, st);
#pragma unroll
    for (int q = 0; q < NPL; ++q) zp[q] = zn[q];
    __syncthreads();
  }
  {
    const int buf = g & 1;
    if (NH >= 2) issue_k(p.Whid[0], p.ld_hid, p.pl_hid, 0, st); else issue_n(p.Wout, p.ld_out, p.pl_out, 0, st);
    mfma_slab(buf, X1, zp);
    CP_PIN_SLAB(NST, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (NH >= 2) store_k(buf ^ 1, st); else store_n(buf ^ 1, st);
    __syncthreads();
    ++g;
  }
  const size_t hbase = (size_t)panc * 8 * CHB + (size_t)lane * 16;      // this lane's line in block 0 of a hidden planes buffer
  auto activate = [&](f32x4 (&X)[T], int l) {
    if (MODE == 2) {
      // lane (j, g)'s line of block ks holds slots 8 g .. 8 g + 7 = registers 0..3 of tiles 2 ks and 2 ks + 1
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const vec8 hv = *reinterpret_cast<const vec8*>(p.gate[l] + hbase + (size_t)ks * CHB);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          X[2 * ks][t] = gate_apply(X[2 * ks][t], (float)hv[t], p.slope);
          X[2 * ks + 1][t] = gate_apply(X[2 * ks + 1][t], (float)hv[4 + t], p.slope);
        }
      }
      return;
    }
#pragma unroll
    for (int ht = 0; ht < T; ++ht) {
#pragma unroll
      for (int t = 0; t < 4; ++t) X[ht][t] = act_apply(X[ht][t], p.act, p.slope);
      guard(X[ht]);
    }
  };
  auto save_hidden = [&](int l, int ks, const vec8 (&x)[NPL]) {
    if (MODE != 0 && live) {
#pragma unroll
      for (int q = 0; q < NPL; ++q) *reinterpret_cast<vec8*>(p.hout[l] + hbase + (size_t)ks * CHB + q * 1024) = x[q];
    }
  };
  activate(X1, 0);
  CSTAMP(c1);

  // ================= phase 2: Xout[h2][row] += W_h[h2][h1'] * Xin[h1'][row] ====================
  auto hidden_layer = [&](f32x4 (&Xin)[T], f32x4 (&Xout)[T], int l) {
    vec8 xc[NPL], xn[NPL];
    PT::split(Xin[0], Xin[1], xc);               // slot order {4g.., 16+4g..}: the weights' K order
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      save_hidden(l, ks, xc);
      const int buf = g & 1;
      const bool last = (ks + 1 == KS);
      const bool next_is_hidden = (l + 2 < NH);
      if (!last) issue_k(p.Whid[l], p.ld_hid, p.pl_hid, (ks + 1) * 32, st);
      else if (next_is_hidden) issue_k(p.Whid[l + 1], p.ld_hid, p.pl_hid, 0, st);
      else issue_n(p.Wout, p.ld_out, p.pl_out, 0, st);
      mfma_slab(buf, Xout, xc);
      if (!last) PT::split(Xin[2 * ks + 2], Xin[2 * ks + 3], xn);
      CP_PIN_SLAB(NST, (NPL == 3) ? 5 : 3);
      __builtin_amdgcn_sched_barrier(0);
      if (!last || next_is_hidden) store_k(buf ^ 1, st); else store_n(buf ^ 1, st);
      if (!last) {
#pragma unroll
        for (int q = 0; q < NPL; ++q) xc[q] = xn[q];
      }
      __syncthreads();
      ++g;
    }
    activate(Xout, l + 1);
  };
  if (NH >= 2) {
#pragma unroll
    for (int t = 0; t < T; ++t) X2[t] = *reinterpret_cast<const f32x4*>(p.b_hid[0] + t * 16 + 4 * lg);
    hidden_layer(X1, X2, 0);
  }
  if (NH >= 3) {
#pragma unroll
    for (int t = 0; t < T; ++t) X1[t] = *reinterpret_cast<const f32x4*>(p.b_hid[1] + t * 16 + 4 * lg);
    hidden_layer(X2, X1, 1);
  }

  CSTAMP(c2);
  // ================= phase 3: z_T[row][n] += sign * (b_out[n] + sum_h X[h][row] W_out[n][h']) on the transformed blocks =====
  // (deferring a block's epilogue -- residual rebuilt from its planes, fma, three-way split, plane stores: ~80 vector
  // instructions per lane -- into the next block's MFMA issue slots was measured in round 4 and did not pay in the flow:
  // profiles/r04_tuning_experiments.md section 2; the code is gone)
  auto output_layer = [&](f32x4 (&X)[T]) {
    vec8 xp[KS][NPL];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) { PT::split(X[2 * ks], X[2 * ks + 1], xp[ks]); save_hidden(NH - 1, ks, xp[ks]); }
    // residual of output block nt (the lane's own chunk line) and its slice of the output bias, one block ahead
    vec8 res[NPL];
    f32x4 bo[2];
    auto issue_res = [&](int nt) {
#pragma unroll
      for (int q = 0; q < NPL; ++q)
        res[q] = *reinterpret_cast<const vec8*>(p.z + zbase + (size_t)(p.kb_t0 + nt) * CHB + q * 1024);
#pragma unroll
      for (int u = 0; u < 2; ++u) bo[u] = *reinterpret_cast<const f32x4*>(p.b_out + nt * 32 + 16 * u + 4 * lg);
    };
    // lane-local epilogue: the two tiles are the lane's 8 slots of its chunk line of block kb_t0 + nt
    auto epilogue = [&](int nt, const auto& ac, const auto& rs) {
      f32x4 v[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float r = (float)rs[0][4 * u + e] + (float)rs[1][4 * u + e];
          if (NPL == 3) r = r + (float)rs[NPL - 1][4 * u + e];
          v[u][e] = __builtin_fmaf(p.sign, ac[u][e], r);       // (sign = +-1: the product is exact, one rounding either way)
        }
        guard(v[u]);
      }
      vec8 o[NPL];
      PT::split(v[0], v[1], o);
      if (live) {
#pragma unroll
        for (int q = 0; q < NPL; ++q)
          *reinterpret_cast<vec8*>(p.z + zbase + (size_t)(p.kb_t0 + nt) * CHB + q * 1024) = o[q];
      }
    };
    issue_res(0);
    // one output block: its MFMAs, then its epilogue
    auto block = [&](int nt) {
      const int buf = g & 1;
      issue_n(p.Wout, p.ld_out, p.pl_out, min(nt + 1, p.nk_t - 1) * 32, st);
      f32x4 acc[2] = {bo[0], bo[1]};                         // the output bias is the accumulators' starting value
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int x = 2 * ((4 * ks + lg) & 7);                 // row swizzle of chunk 4 ks + g
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const float* wf = &lds[buf][4 * ((4 * ks + lg) * 32 + u * 16 + (lj ^ x))];
          vec8 w[NPL];
#pragma unroll
          for (int q = 0; q < NPL; ++q) w[q] = *reinterpret_cast<const vec8*>(wf + 4 * (q * NC * 32));
          PT::mm(acc[u], w, xp[ks]);
        }
      }
      __builtin_amdgcn_sched_group_barrier(0x100, NPL * AHEAD, 0);
#pragma unroll
      for (int i = 0; i < 2 * KS; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, NPR, 0);
        if (i + AHEAD < 2 * KS) __builtin_amdgcn_sched_group_barrier(0x100, NPL, 0);
        if (i < NST + NPL + 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      store_n(buf ^ 1, st);
      epilogue(nt, acc, res);
      issue_res(min(nt + 1, p.nk_t - 1));
      __syncthreads();
      ++g;
    };
    for (int nt = 0; nt < p.nk_t; ++nt) block(nt);
  };
  if (NH == 2) output_layer(X2); else output_layer(X1);
#ifdef USF_STAMP
  CSTAMP(c3);
  if (p.dbg && lane == 0) {
    unsigned long long* o = p.dbg + (size_t)((blockIdx.x % 2048) * 8 + wave) * 4;
    o[0] = c1 - c0; o[1] = c2 - c1; o[2] = c3 - c2; o[3] = 1;
  }
#endif
  if (NPL == 2 && p.range_flag && bad && live && (16 * panel + lj) < p.M) atomicOr(p.range_flag, 1);
#undef CP_PIN_SLAB
}


// (Round 3's 32-rows-per-wave variant on the 512-register budget -- half the LDS fragment reads per product, 10-13 % faster
// stand-alone, never faster between the flow's GEMMs (18.77 vs 18.93 ms per step: profiles/r03_tuning_experiments.md
// section 1) -- was a selectable dead end for two rounds and is removed; its measurements stay in the profiles.)

#ifdef USF_STAMP
unsigned long long* g_cdbg = nullptr;
#endif

int coupling_planes(const usf_coupling_planes_desc* d, hipStream_t stream) {
  if (!d) { set_error("usf_coupling_planes: null descriptor"); return -1; }
  if (d->M < 0 || d->M > 0x7fffffff || d->z_nkb <= 0 || d->n_hidden < 1 || d->n_hidden > 3 || d->nk_p <= 0 || d->nk_t <= 0 ||
      d->kb_p0 < 0 || d->kb_t0 < 0 || d->kb_p0 + d->nk_p > d->z_nkb || d->kb_t0 + d->nk_t > d->z_nkb) {
    set_error("usf_coupling_planes: bad sizes / block ranges");
    return -2;
  }
  if (d->M == 0) return 0;
  if (d->format != USF_PLANES_BF16X3 && d->format != USF_PLANES_F16X2) { set_error("usf_coupling_planes: unknown format %d", d->format); return -2; }
  if (!d->z || !d->W_in || !d->W_out || !d->b_in || !d->b_out) { set_error("usf_coupling_planes: null pointer"); return -1; }
  for (int i = 0; i + 1 < d->n_hidden; ++i)
    if (!d->W_hid[i] || !d->b_hid[i]) { set_error("usf_coupling_planes: hidden layer %d missing", i); return -1; }
  if (d->hidden_padded != 256) { set_error("usf_coupling_planes: weights must be padded to a hidden width of 256 (got %d)", d->hidden_padded); return -2; }
  if (d->ldw_in < 32 * d->nk_p || (d->n_hidden > 1 && d->ldw_hid < 256) || d->ldw_out < 256 || (d->ldw_in & 7) || (d->ldw_hid & 7) ||
      (d->ldw_out & 7) || !aligned16(d->z) || !aligned16(d->W_in) || !aligned16(d->W_out) || !aligned16(d->b_in) || !aligned16(d->b_out)) {
    set_error("usf_coupling_planes: weight-image contract violated (ldw_in >= 32 nk_p, ldw_hid / ldw_out >= 256, multiples of 8, 16-byte aligned)");
    return -2;
  }
  const int64_t npl = d->format == USF_PLANES_F16X2 ? 2 : 3;
  const int64_t npanels = (d->M + 15) / 16;
  if (2 * npl * d->w_in_plane >= (1LL << 40)) { set_error("usf_coupling_planes: operand too large"); return -3; }
  CplPArgs a;
  a.z = reinterpret_cast<char*>(d->z); a.z_nkb = (int)d->z_nkb; a.npanels = (int)npanels; a.M = (int)d->M;
  a.kb_p0 = (int)d->kb_p0; a.nk_p = (int)d->nk_p; a.kb_t0 = (int)d->kb_t0; a.nk_t = (int)d->nk_t;
  a.Win = reinterpret_cast<const char*>(d->W_in); a.ld_in = d->ldw_in; a.pl_in = d->w_in_plane; a.b_in = d->b_in;
  for (int i = 0; i < 2; ++i) {
    const bool used = i + 1 < d->n_hidden;
    a.Whid[i] = reinterpret_cast<const char*>(used ? d->W_hid[i] : d->W_in);
    a.b_hid[i] = used ? d->b_hid[i] : d->b_in;
  }
  a.ld_hid = d->ldw_hid; a.pl_hid = d->w_hid_plane;
  a.Wout = reinterpret_cast<const char*>(d->W_out); a.ld_out = d->ldw_out; a.pl_out = d->w_out_plane; a.b_out = d->b_out;
  a.sign = d->sign; a.slope = d->slope; a.act = d->act; a.range_flag = d->range_flag;
  a.dbg = nullptr;
#ifdef USF_STAMP
  a.dbg = g_cdbg;
#endif
  const dim3 grid((unsigned)((npanels + 7) / 8));
  // ---- training forms (ABI 33): hidden_out / gate ----
  const bool want_h = d->hidden_out[0] != nullptr || d->hidden_out[1] != nullptr;
  const bool gated = d->act == USF_ACT_GATE;
  if (d->act != USF_ACT_NONE && d->act != USF_ACT_LEAKY_RELU && !gated) { set_error("usf_coupling_planes: bad act %d", d->act); return -2; }
  for (int i = 0; i < 2; ++i) { a.hout[i] = nullptr; a.gate[i] = nullptr; }
  if (want_h || gated) {
    if (npl != 3 || d->n_hidden > 2) { set_error("usf_coupling_planes: hidden_out / USF_ACT_GATE need the bf16x3 format and n_hidden <= 2"); return -2; }
    if (gated && !want_h) { set_error("usf_coupling_planes: USF_ACT_GATE needs hidden_out (the gradients at the hidden pre-activations)"); return -2; }
    for (int i = 0; i < d->n_hidden; ++i) {
      if (!d->hidden_out[i] || !aligned16(d->hidden_out[i]) || (gated && (!d->gate[i] || !aligned16(d->gate[i])))) {
        set_error("usf_coupling_planes: hidden_out[l] (and gate[l] with USF_ACT_GATE) must be set for every hidden layer, 16-byte aligned");
        return -2;
      }
      a.hout[i] = reinterpret_cast<char*>(d->hidden_out[i]);
      a.gate[i] = reinterpret_cast<const char*>(gated ? d->gate[i] : nullptr);
    }
    const dim3 block(512);
#define USF_CPT(NH_, MODE_) hipLaunchKernelGGL((coupling_planes_kernel<3, NH_, MODE_>), grid, block, 0, stream, a)
    if (gated) { if (d->n_hidden == 1) USF_CPT(1, 2); else USF_CPT(2, 2); }
    else { if (d->n_hidden == 1) USF_CPT(1, 1); else USF_CPT(2, 1); }
#undef USF_CPT
    return check_launch("usf_coupling_planes");
  }
  const dim3 block(512);
#define USF_CPL(NPL_, NH_) hipLaunchKernelGGL((coupling_planes_kernel<NPL_, NH_>), grid, block, 0, stream, a)
  if (npl == 2) {
    switch (d->n_hidden) { case 1: USF_CPL(2, 1); break; case 2: USF_CPL(2, 2); break; default: USF_CPL(2, 3); break; }
  } else {
    switch (d->n_hidden) { case 1: USF_CPL(3, 1); break; case 2: USF_CPL(3, 2); break; default: USF_CPL(3, 3); break; }
  }
#undef USF_CPL
  return check_launch("usf_coupling_planes");
}

}  // namespace usf
