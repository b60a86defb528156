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

template <bool B> struct CpBoolT { static constexpr bool value = B; };

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
  // The epilogue of output block nt (residual rebuilt from its planes, fma, three-way split, plane stores: ~80 vector
  // instructions per lane) is deferred by one block: it sits in the instruction stream of block nt + 1 and is pinned, five
  // instructions per MFMA group, into the issue slots the matrix instructions leave free (-DUSF_CP_PIPE_EPI=1; round 4: an
  // experiment that did NOT pay in the flow and is off by default) -- otherwise both waves of a SIMD run it back to back
  // behind their MFMAs.  The stores are raw buffer stores
  // whose resource is empty for a wave beyond the last panel (dropped by the hardware): no branch inside the pinned region.
  // (NH == 3 runs at the 256-register budget: it keeps the epilogue behind its own block.)
#ifndef USF_CP_PIPE_EPI
#define USF_CP_PIPE_EPI 0          // measured in the flow (profiles/r04_tuning_experiments.md): 6.53 ms of couplings per step with, 6.46 without
#endif
  constexpr bool PIPE = (USF_CP_PIPE_EPI != 0) && NH <= 2;
  typedef unsigned cp_u32x4 __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t zrs = __builtin_amdgcn_make_buffer_rsrc(
      p.z, 0, live ? (int)((size_t)p.npanels * p.z_nkb * CHB) : 0, 0x00020000);
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
      if (PIPE) {
        const unsigned off = (unsigned)(zbase + (size_t)(p.kb_t0 + nt) * CHB);
#pragma unroll
        for (int q = 0; q < NPL; ++q)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(cp_u32x4, o[q]), zrs, (int)(off + q * 1024u), 0, 0);
      } else if (live) {
#pragma unroll
        for (int q = 0; q < NPL; ++q)
          *reinterpret_cast<vec8*>(p.z + zbase + (size_t)(p.kb_t0 + nt) * CHB + q * 1024) = o[q];
      }
    };
    issue_res(0);
    f32x4 accp[PIPE ? 2 : 1];
    vec8 resp[PIPE ? NPL : 1];
    // one output block: its MFMAs -- and, pipelined, the previous block's epilogue in their shadow
    auto block = [&](int nt, auto with_prev) {
      constexpr bool PREV = decltype(with_prev)::value;
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
      if constexpr (PREV) epilogue(nt - 1, accp, resp);
      __builtin_amdgcn_sched_group_barrier(0x100, NPL * AHEAD, 0);
#pragma unroll
      for (int i = 0; i < 2 * KS; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, NPR, 0);
        if (i + AHEAD < 2 * KS) __builtin_amdgcn_sched_group_barrier(0x100, NPL, 0);
        if (i < NST + NPL + 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        if (PREV && i >= 1) __builtin_amdgcn_sched_group_barrier(0x002, (NPL == 3) ? 6 : 4, 0);
        if (PREV && i >= 2 * KS - NPL) __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      store_n(buf ^ 1, st);
      if constexpr (!PIPE) {
        epilogue(nt, acc, res);
      } else {
        accp[0] = acc[0]; accp[1] = acc[1];
#pragma unroll
        for (int q = 0; q < NPL; ++q) resp[q] = res[q];
      }
      issue_res(min(nt + 1, p.nk_t - 1));
      __syncthreads();
      ++g;
    };
    if constexpr (PIPE) {
      block(0, CpBoolT<false>());
      for (int nt = 1; nt < p.nk_t; ++nt) block(nt, CpBoolT<true>());
      epilogue(p.nk_t - 1, accp, resp);
    } else {
      for (int nt = 0; nt < p.nk_t; ++nt) block(nt, CpBoolT<false>());
    }
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


// ------------------------------------------------------------------------------------------------------------
// coupling_planes_w32_kernel: the same layer with 32 batch rows (two row panels) per wave and ONE wave per SIMD
// (256-thread blocks of 128 rows on the 512-register budget).  Why (profiles/r02_tuning_experiments.md section 3 and the
// round-2 review): with 16-row waves every weight fragment read from LDS feeds 6 MFMAs (the planes GEMM: 12) -- twice
// the LDS read bytes per product on a chip whose clock is set by the energy per product.  Here a fragment feeds the
// wave's two batch tiles (12 MFMAs); both hidden layers' accumulators of 32 rows (2 x 128 registers) fit because the
// wave owns the whole register file of its SIMD.  Weight stages travel through a ring of three LDS buffers with ONE
// barrier in the MIDDLE of a stage (the planes GEMM's loop structure): a stage's staging stores sit in the first half
// of the stage in front of it, its global loads half a stage earlier still, and its first three fragments are read
// under the last MFMAs of the stage before -- across phase boundaries too.  The stage list of a layer is linear: nk_p
// k-slabs of W_in, 8 k-slabs per hidden layer, nk_t n-tiles of W_out.  Per accumulator the products are summed in the
// order of coupling_planes_kernel: the two kernels give bit-identical results.
// ------------------------------------------------------------------------------------------------------------
template <bool B> struct CpBool { static constexpr bool value = B; };
template <int I> struct CpInt { static constexpr int value = I; };

template <int NPL, int NH>
__global__ __launch_bounds__(256) void coupling_planes_w32_kernel(const CplPArgs p) {
  typedef CPlanes<NPL> PT;
  typedef typename PT::vec vec8;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  constexpr int T = 16;                     // tiles per stage (hidden width 256 / n-tile of 32 outputs x 8 k-steps)
  constexpr int HP = 16 * T;
  constexpr int KS = T / 2;
  constexpr int NT = 256;
  constexpr int SLOTS = NPL * 4 * HP;       // 16-B slots per stage
  constexpr int STGF = SLOTS * 4;           // floats per stage
  constexpr int NPP = (4 * HP) / NT;        // staged 16-B pieces per thread and plane
  constexpr int NST = NPL * NPP;
  constexpr int NB = 3;
  constexpr int PLF = 16 * HP;              // floats between the planes of a stage image (both image kinds)
#ifndef USF_CPW_AH
#define USF_CPW_AH 1
#endif
#ifndef USF_CPW_ABL
#define USF_CPW_ABL 0      // tuning builds (wrong results): 1 no operand / residual loads, 2 no staging stores, 4 no weight loads,
#endif                     // 8 no mid-stage barriers, 16 no plane stores, 32 no fragment reads, 64 no side VALU (act / split / epilogue)
  constexpr int AH = USF_CPW_AH;            // fragments are read this many tiles ahead (tile i lives in register set i % 4)
  static_assert(AH >= 1 && AH <= 3, "four fragment sets");
  constexpr unsigned CHB = NPL * 1024u;
  constexpr int NPR2 = (NPL == 3) ? 12 : 6; // MFMAs per tile (both batch tiles)
  __shared__ __attribute__((aligned(16))) float lds[NB * STGF];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int lj = lane & 15, lg = lane >> 4;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const float slope_eff = (p.act == USF_ACT_LEAKY_RELU) ? p.slope : 1.0f;     // x > 0 ? x : x * 1 == x

  // ---- z through a buffer resource that covers exactly this block's valid panels: rows beyond M load zeros and
  // their stores are dropped by the hardware (no branch anywhere in the stage bodies) ----
  const int panel0 = blockIdx.x * 8;
  const int nvalid = min(8, p.npanels - panel0);
  const unsigned panel_bytes = (unsigned)p.z_nkb * CHB;
  const __amdgpu_buffer_rsrc_t zrs = __builtin_amdgcn_make_buffer_rsrc(
      p.z + (size_t)panel0 * panel_bytes, 0, (int)((unsigned)nvalid * panel_bytes), 0x00020000);
  unsigned zoff[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) zoff[b] = (unsigned)(2 * wave + b) * panel_bytes + (unsigned)lane * 16u;
  auto load_zblk = [&](int kb, vec8 (&dst)[2][NPL]) {
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int q = 0; q < NPL; ++q)
        dst[b][q] = __builtin_bit_cast(vec8, __builtin_amdgcn_raw_buffer_load_b128(zrs, (int)(zoff[b] + (unsigned)kb * CHB + (unsigned)q * 1024u), 0, 0));
  };
  auto store_zblk = [&](int kb, int b, const vec8 (&src)[NPL]) {
#pragma unroll
    for (int q = 0; q < NPL; ++q)
      if (USF_CPW_ABL & 16) asm volatile("" :: "v"(src[q])); else
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, src[q]), zrs, (int)(zoff[b] + (unsigned)kb * CHB + (unsigned)q * 1024u), 0, 0);
  };

  // ---- weight stages: the layer's stage list is linear (nk_p k-slabs of W_in, 8 k-slabs per hidden layer, nk_t
  // n-tiles of W_out).  Which matrix / image kind a stage has is a compile-time fact everywhere except in the loop of
  // phase 1, whose last two stages stage the first two stages of the next phase: selects there, no branch ----
  // (pointers made opaque: a select of two kernel-argument loads is turned into a load from a selected address, i.e. a
  //  scalar-load latency in front of the MFMAs of every stage)
  auto opaque = [](const char* q) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)q);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((uintptr_t)q >> 32));
    return reinterpret_cast<const char*>(((uintptr_t)hi << 32) | (uintptr_t)lo);
  };
  const char* const wp_in = opaque(p.Win);
  const char* const wp_a1 = opaque(NH >= 2 ? p.Whid[0] : p.Wout);        // the matrix behind phase 1
  const unsigned ld_in = (unsigned)__builtin_amdgcn_readfirstlane((int)p.ld_in), pl2_in = 2u * (unsigned)__builtin_amdgcn_readfirstlane((int)p.pl_in);
  const unsigned ld_a1 = (unsigned)__builtin_amdgcn_readfirstlane((int)(NH >= 2 ? p.ld_hid : p.ld_out));
  const unsigned pl2_a1 = 2u * (unsigned)__builtin_amdgcn_readfirstlane((int)(NH >= 2 ? p.pl_hid : p.pl_out));
  // staging geometry of this thread: k-slab image piece i = (row (tid >> 2) + 64 i, chunk tid & 3);
  // n-tile image piece i = (row (tid >> 5) + 8 i, chunk tid & 31)
  int kdst[NPP], ndst[NPP];
#pragma unroll
  for (int i = 0; i < NPP; ++i) {
    const int rk = (tid >> 2) + 64 * i, ck = tid & 3;
    kdst[i] = 4 * (ck * HP + (rk ^ (2 * ck)));
    const int rn = (tid >> 5) + 8 * i, cn = tid & 31;
    ndst[i] = 4 * (cn * 32 + (rn ^ (2 * (cn & 7))));
  }
  f32x4 st[NST];
  // global loads of one stage: k-slab at column x0 (is_n false) or n-tile at row x0 (is_n true) of W
  auto issue_w = [&](const char* W, unsigned ld, unsigned pl2, unsigned x0, bool is_n) {
    const unsigned cpart = is_n ? 8u * (unsigned)(tid & 31) : x0 + 8u * (unsigned)(tid & 3);
#pragma unroll
    for (int i = 0; i < NPP; ++i) {
      const unsigned r = is_n ? x0 + (unsigned)((tid >> 5) + 8 * i) : (unsigned)((tid >> 2) + 64 * i);
      const unsigned off = 2u * (r * ld + cpart);
#pragma unroll
      for (int q = 0; q < NPL; ++q) {
        typedef const f32x4 __attribute__((address_space(1)))* gptr;
        st[q * NPP + i] = *(gptr)(uintptr_t)(W + (size_t)((unsigned)q * pl2 + off));
      }
    }
  };
  // staging store of piece j (= q * NPP + i) into ring slot `ring`
  auto store_piece = [&](int ring, int j, bool is_n) {     // (ring < 0: slot -ring - 1, exempt from the ablation switch)
    const int i = j % NPP, q = j / NPP;
    if ((USF_CPW_ABL & 2) && ring >= 0) { asm volatile("" :: "v"(st[j])); return; }
    if (ring < 0) ring = -ring - 1;
    *reinterpret_cast<f32x4*>(lds + ring * STGF + q * PLF + (is_n ? ndst[i] : kdst[i])) = st[j];
  };
  // fragment addresses: k-slab image tile ht: kfr + 64 ht; n-tile image tile 2 ks + u: nfr[ks & 1] + 512 ks + 64 u
  const int kfr = 4 * (lg * HP + (lj ^ (2 * lg)));
  int nfr[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) nfr[e] = 4 * (32 * lg + (lj ^ (2 * ((4 * e + lg) & 7))));
  auto frag = [&](const float* a, vec8 (&w)[NPL]) {
#pragma unroll
    for (int q = 0; q < NPL; ++q) w[q] = *reinterpret_cast<const vec8*>(a + q * PLF);
  };
  auto tile_ptr = [&](int ring, int i, bool is_n) -> const float* {
    return lds + ring * STGF + (is_n ? nfr[(i >> 1) & 1] + 512 * (i >> 1) + 64 * (i & 1) : kfr + 64 * i);
  };
  auto next_ring = [](int r) { return r == NB - 1 ? 0 : r + 1; };

  // two batch tiles against one weight fragment set: twelve (six) MFMAs, the two accumulators alternating; in two
  // halves (a tile = two half-tile chunks of the instruction stream, below)
  auto mm2a = [&](f32x4& a0, f32x4& a1, const vec8 (&w)[NPL], const vec8 (&b0)[NPL], const vec8 (&b1)[NPL]) {
    if (NPL == 3) {
      a0 = PT::mfma(w[NPL - 1], b0[0], a0); a1 = PT::mfma(w[NPL - 1], b1[0], a1);
      a0 = PT::mfma(w[1], b0[1], a0); a1 = PT::mfma(w[1], b1[1], a1);
      a0 = PT::mfma(w[0], b0[NPL - 1], a0); a1 = PT::mfma(w[0], b1[NPL - 1], a1);
    } else {
      a0 = PT::mfma(w[1], b0[0], a0); a1 = PT::mfma(w[1], b1[0], a1);
      a0 = PT::mfma(w[0], b0[1], a0); a1 = PT::mfma(w[0], b1[1], a1);
    }
  };
  auto mm2b = [&](f32x4& a0, f32x4& a1, const vec8 (&w)[NPL], const vec8 (&b0)[NPL], const vec8 (&b1)[NPL]) {
    if (NPL == 3) {
      a0 = PT::mfma(w[1], b0[0], a0); a1 = PT::mfma(w[1], b1[0], a1);
      a0 = PT::mfma(w[0], b0[1], a0); a1 = PT::mfma(w[0], b1[1], a1);
    }
    a0 = PT::mfma(w[0], b0[0], a0); a1 = PT::mfma(w[0], b1[0], a1);
  };
  constexpr int NMA = (NPL == 3) ? 6 : 4;   // MFMAs of the two half tiles
  constexpr int NMB = (NPL == 3) ? 6 : 2;

  bool bad = false;
  auto guard1 = [&](float v) { if (NPL == 2) bad = bad || !(fabsf(v) < USF_CP_F16_GUARD); };
  auto act1 = [&](float v) { v = v > 0.0f ? v : v * slope_eff; guard1(v); return v; };
  // lane-local split of values (j, j + 1) of a chunk line (j even: the pair shares a dword of every plane)
  auto split_pair = [&](float x0, float x1, int j, vec8 (&o)[NPL]) {
    PT::split1(x0, j, o);
    PT::split1(x1, j + 1, o);
  };

  // The instruction stream of a stage is written as 32 half-tile CHUNKS separated by scheduling fences: chunk A of tile i
  // = the first half of its MFMAs + the fragment reads of tile i + AH + a piece of side work, chunk B = the other MFMAs
  // + another piece (staging stores in front of the barrier, the weight loads of stage g + 2 behind it, operand loads,
  // the next operand's activation + split, the previous n-tile's epilogue).  Inside a chunk the pins alternate one
  // MFMA with two vector instructions (a 16x16x32 MFMA leaves the SIMD's issue port free for 8 of its 16 cycles) and
  // put memory instructions behind them.  One wave per SIMD: whatever is not under an MFMA is idle matrix pipe.
  auto fence = [&]() { __builtin_amdgcn_sched_barrier(0); };
  auto pins = [&](int nm, bool reads_first = false) {
    // (literal counts; the chain folds)
    if (reads_first) __builtin_amdgcn_sched_group_barrier(0x100, NPL, 0);
    if (nm >= 1) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 2, 0); }
    if (nm >= 2) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 2, 0); }
    if (nm >= 3) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 2, 0); }
    if (nm >= 4) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 2, 0); }
    if (nm >= 5) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 2, 0); }
    if (nm >= 6) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 2, 0); }
    if (!reads_first) __builtin_amdgcn_sched_group_barrier(0x100, NPL, 0);      // fragment reads
    __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);        // loads
    __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);        // staging stores
    __builtin_amdgcn_sched_group_barrier(0x040, NPL, 0);      // plane stores
  };
#if defined(USF_STAMP) && USF_STAMP >= 2
  unsigned long long bwait = 0, bw_mark[3] = {0, 0, 0}, tstage = 0, smax = 0, smin = ~0ull;
#endif
  auto mid_barrier = [&]() {
    __builtin_amdgcn_sched_barrier(0);
#if defined(USF_STAMP) && USF_STAMP >= 2
    const unsigned long long ta = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
    if (tstage) { const unsigned long long dt = ta - tstage; smax = dt > smax ? dt : smax; smin = dt < smin ? dt : smin; }
    tstage = ta;
#endif
    if (!(USF_CPW_ABL & 8)) __syncthreads();
#if defined(USF_STAMP) && USF_STAMP >= 2
    __builtin_amdgcn_sched_barrier(0);
    bwait += __builtin_amdgcn_s_memtime() - ta;
#endif
    __builtin_amdgcn_sched_barrier(0);
  };

  // accumulators: X1 starts at the first layer's bias
  f32x4 X1[T][2], X2[T][2];
#pragma unroll
  for (int t = 0; t < T; ++t) { X1[t][0] = *reinterpret_cast<const f32x4*>(p.b_in + t * 16 + 4 * lg); X1[t][1] = X1[t][0]; }

  CSTAMP(c0);
  constexpr bool A1N = (NH < 2);             // the stage behind phase 1 is an n-tile (no hidden layer)
  // ---- pipeline fill: stage 0 staged, stage 1 in the staging registers, first fragments and operands in flight ----
  vec8 zp[2][NPL], zn[2][NPL];
  issue_w(wp_in, ld_in, pl2_in, 0, false);
  load_zblk(p.kb_p0, zp);
#pragma unroll
  for (int j = 0; j < NST; ++j) {
    store_piece(-1, j, false);
    if (USF_CPW_ABL & 6) { store_piece(-2, j, false); store_piece(-3, j, false); }       // ablation builds: sane bytes in every slot
  }
  issue_w(wp_in, ld_in, pl2_in, 32, false);  // (nk_p >= 2: the host routes narrower layers to the 16-row kernel)
  __syncthreads();
  vec8 fr[4][NPL];                            // weight fragments: tile i lives in set i % 4; tiles 0 .. AH - 1 are in flight on stage entry
#pragma unroll
  for (int i = 0; i < AH; ++i) frag(tile_ptr(0, i, false), fr[i]);
  int ring = 0;

  // (MFMAs first in program order: at the head of a loop body the compiler waits for EVERY outstanding LDS operation
  //  before the first MFMA -- its counters are merged conservatively across the back edge -- and a read issued in front
  //  of that wait would expose its latency in every stage.)
#define W32_HALF_A(I_, NXPTR_, A0_, A1_, B0_, B1_)                                                \
  do {                                                                                            \
    if ((I_) == 0) {                      /* stage head: MFMAs first (see above) */               \
      mm2a(A0_, A1_, fr[(I_) % 4], B0_, B1_);                                                     \
      if (!(USF_CPW_ABL & 32)) frag(NXPTR_, fr[((I_) + AH) % 4]);                                 \
    } else {                              /* elsewhere the reads lead: AH tiles + half a tile of latency cover */ \
      if (!(USF_CPW_ABL & 32)) frag(NXPTR_, fr[((I_) + AH) % 4]);                                 \
      mm2a(A0_, A1_, fr[(I_) % 4], B0_, B1_);                                                     \
    }                                                                                             \
  } while (0)
#define W32_HALF_B(I_, A0_, A1_, B0_, B1_) mm2b(A0_, A1_, fr[(I_) % 4], B0_, B1_)
  // side work every stage carries in the B chunks in front of its barrier: staging piece j is stored (weights of stage
  // g + 1, loaded a stage ago) and its registers are re-loaded at once (weights of stage g + 2: a full stage of latency
  // cover).  IN PROGRAM ORDER between the fragment reads: the ring slots are runtime values, LDS accesses keep their
  // order.  Why in the first half: vmcnt counts loads and stores together, in issue order -- a staging store waits for
  // everything older than its load, and with the loads in front of the stage's plane stores that is never a store a
  // few hundred cycles old.
  auto issue_piece = [&](const char* W, unsigned ld, unsigned pl2, unsigned x0, bool is_n, int j) {
    const int i = j % NPP, q = j / NPP;
    if (USF_CPW_ABL & 4) return;
    const unsigned cpart = is_n ? 8u * (unsigned)(tid & 31) : x0 + 8u * (unsigned)(tid & 3);
    const unsigned r = is_n ? x0 + (unsigned)((tid >> 5) + 8 * i) : (unsigned)((tid >> 2) + 64 * i);
    typedef const f32x4 __attribute__((address_space(1)))* gptr;       // a pointer rebuilt from integers is FLAT to the compiler: flat
    st[j] = *(gptr)(uintptr_t)(W + (size_t)((unsigned)q * pl2 + 2u * (r * ld + cpart)));   // loads count in lgkmcnt too, out of order
  };
  auto stage_side = [&](int i, int rn, bool next_n, const char* W2, unsigned ld2, unsigned pl22, unsigned x02, bool n2) {
    if (i < NST - T / 2) {
      store_piece(rn, 2 * i, next_n); store_piece(rn, 2 * i + 1, next_n);
      issue_piece(W2, ld2, pl22, x02, n2, 2 * i); issue_piece(W2, ld2, pl22, x02, n2, 2 * i + 1);
    } else if (i < T / 2) {
      store_piece(rn, i + NST - T / 2, next_n);
      issue_piece(W2, ld2, pl22, x02, n2, i + NST - T / 2);
    }
  };
  auto zload_piece = [&](int kb, int j, vec8 (&dst)[2][NPL]) {          // j = b * NPL + q
    const int b = j / NPL, q = j % NPL;
    if (USF_CPW_ABL & 1) { dst[b][q] = zp[b][q]; return; }
    dst[b][q] = __builtin_bit_cast(vec8, __builtin_amdgcn_raw_buffer_load_b128(zrs, (int)(zoff[b] + (unsigned)kb * CHB + (unsigned)q * 1024u), 0, 0));
  };

  // one k-slab stage out of ring slot `ring`: X[ht] += W[ht] . (b0 | b1).  sideA(i): the stage's own piece of side work
  // for chunk A of tile i; next_n: image kind of the following stage; (W2, ld2, pl22, x02, n2): the stage after that
  auto kslab_stage = [&](f32x4 (&X)[T][2], const vec8 (&b0)[NPL], const vec8 (&b1)[NPL], auto sideA, bool next_n,
                         const char* W2, unsigned ld2, unsigned pl22, unsigned x02, bool n2) {
    const int rn = next_ring(ring);
#pragma unroll
    for (int ht = 0; ht < T; ++ht) {
      const float* nx = (ht + AH < T) ? tile_ptr(ring, ht + AH, false) : tile_ptr(rn, ht + AH - T, next_n);
      W32_HALF_A(ht, nx, X[ht][0], X[ht][1], b0, b1);
      sideA(ht);
      pins(NMA, ht > 0);
      fence();
      W32_HALF_B(ht, X[ht][0], X[ht][1], b0, b1);
      stage_side(ht, rn, next_n, W2, ld2, pl22, x02, n2);
      pins(NMB);
      fence();
      if (ht == T / 2 - 1) mid_barrier();
    }
    ring = rn;
  };

  // ================= phase 1: X1[h][row] += W_in[h][k] * z[row][k] over the conditioning blocks ============
  // stage s: next stage's kind and the weights of stage s + 2 are runtime facts near the end of the phase; the operand
  // planes of slab s + 1 are loaded under tiles 1 .. 2 NPL (behind the first staging stores: those wait for the staged
  // weights with a conservative vmcnt(0) at the head of a loop body, which must not cover a load issued a moment ago)
  auto p1_stage = [&](int s, const vec8 (&zc)[2][NPL], vec8 (&zo)[2][NPL]) {
    const bool next_n = A1N && (s + 1 >= p.nk_p);
    const bool beyond = s + 2 >= p.nk_p;                         // stage s + 2 belongs to the next phase
    const unsigned j = (unsigned)(s + 2 - p.nk_p);               // ... as its stage j (0 or 1)
    const char* W = beyond ? wp_a1 : wp_in;
    const unsigned ld = beyond ? ld_a1 : ld_in, pl2 = beyond ? pl2_a1 : pl2_in;
    const unsigned x0 = beyond ? 32u * (A1N ? min(j, (unsigned)p.nk_t - 1u) : j) : 32u * (unsigned)(s + 2);
    const int kbn = p.kb_p0 + min(s + 1, p.nk_p - 1);
    kslab_stage(X1, zc[0], zc[1], [&](int i) { if (i >= 1 && i <= 2 * NPL) zload_piece(kbn, i - 1, zo); }, next_n,
                W, ld, pl2, x0, A1N && beyond);
  };
  for (int s = 0; s < p.nk_p; s += 2) {
    p1_stage(s, zp, zn);
    if (s + 1 < p.nk_p) p1_stage(s + 1, zn, zp);
  }
  CSTAMP(c1);
#if defined(USF_STAMP) && USF_STAMP >= 2
  bw_mark[0] = bwait;
#endif

  // ================= phase 2: Xout[h2][row] += W_h[h2][h1'] * act(Xin)[h1'][row] ====================
  // the activation and the split of the next k-step's operand ride under the current step's MFMAs, two values (one
  // dword of every plane) per chunk: piece k = 4 b + jj covers slots 2 jj, 2 jj + 1 of batch tile b
  auto act_split_piece = [&](f32x4 (&X)[T][2], int ks, int k, vec8 (&o)[2][NPL]) {
    const int b = k >> 2, j = 2 * (k & 3);
    if ((USF_CPW_ABL & 64) && ks > 0) { if (k == 0) { for (int bb = 0; bb < 2; ++bb) for (int q = 0; q < NPL; ++q) o[bb][q] = zp[bb][q]; } return; }
    split_pair(act1(X[2 * ks + (j >> 2)][b][j & 3]), act1(X[2 * ks + (j >> 2)][b][(j & 3) + 1]), j, o[b]);
  };
  auto hidden_layer = [&](f32x4 (&Xin)[T][2], f32x4 (&Xout)[T][2], auto layer_tag) {
    constexpr int L = decltype(layer_tag)::value;          // hidden layer index: weights W_hid[L]
    constexpr bool MORE = (L + 2 < NH);                    // another hidden layer follows
    vec8 xa[2][NPL], xb[2][NPL];
#pragma unroll
    for (int k = 0; k < 8; ++k) act_split_piece(Xin, 0, k, xa);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const bool next_n = !(ks + 1 < KS || MORE);
      const bool own = ks + 2 < KS;
      const char* W2 = own ? p.Whid[L] : (MORE ? p.Whid[L + 1 < 2 ? L + 1 : 1] : p.Wout);
      const unsigned ld2 = (unsigned)((own || MORE) ? p.ld_hid : p.ld_out), pl22 = 2u * (unsigned)((own || MORE) ? p.pl_hid : p.pl_out);
      const bool n2 = !(own || MORE);
      const unsigned x02 = own ? 32u * (ks + 2) : (n2 ? 32u * (unsigned)min(ks + 2 - KS, p.nk_t - 1) : 32u * (ks + 2 - KS));
      if (ks & 1) kslab_stage(Xout, xb[0], xb[1], [&](int i) { if (ks + 1 < KS && i >= 2 && i < 10) act_split_piece(Xin, ks + 1, i - 2, xa); }, next_n, W2, ld2, pl22, x02, n2);
      else        kslab_stage(Xout, xa[0], xa[1], [&](int i) { if (ks + 1 < KS && i >= 2 && i < 10) act_split_piece(Xin, ks + 1, i - 2, xb); }, next_n, W2, ld2, pl22, x02, n2);
    }
  };
  if (NH >= 2) {
#pragma unroll
    for (int t = 0; t < T; ++t) { X2[t][0] = *reinterpret_cast<const f32x4*>(p.b_hid[0] + t * 16 + 4 * lg); X2[t][1] = X2[t][0]; }
    hidden_layer(X1, X2, CpInt<0>());
  }
  if (NH >= 3) {
#pragma unroll
    for (int t = 0; t < T; ++t) { X1[t][0] = *reinterpret_cast<const f32x4*>(p.b_hid[1] + t * 16 + 4 * lg); X1[t][1] = X1[t][0]; }
    hidden_layer(X2, X1, CpInt<1>());
  }
  CSTAMP(c2);
#if defined(USF_STAMP) && USF_STAMP >= 2
  bw_mark[1] = bwait;
#endif

  // ================= phase 3: z_T[row][n] += sign * (b_out[n] + sum_h X[h][row] W_out[n][h']) on the transformed blocks =====
  auto output_layer = [&](f32x4 (&X)[T][2]) {
    vec8 xp[KS][2][NPL];
    // Register plan (the stage bodies of this phase are the tightest: 192 registers of operand planes): the output bias
    // is the accumulators' starting value; the residual planes of n-tile nt are loaded under tiles 1 .. 2 NPL of stage
    // nt and read by its epilogue under stage nt + 1 (two sets that alternate with the accumulators: a full stage of
    // latency cover -- a set re-loaded in the stage that reads it would wait for HBM with a single wave on the SIMD).
    vec8 rA[2][NPL], rB[2][NPL];            // residual planes [batch tile][plane]
    f32x4 bn[2];                            // bias slices of the NEXT n-tile
    f32x4 aA[2][2], aB[2][2];               // accumulators [u][batch tile]: the two sets alternate between "being computed" and "finished"
    auto load_bias = [&](int nt) {
#pragma unroll
      for (int u = 0; u < 2; ++u) bn[u] = *reinterpret_cast<const f32x4*>(p.b_out + min(nt, p.nk_t - 1) * 32 + 16 * u + 4 * lg);
    };

    // epilogue of a finished n-tile (residual from the planes, update, split, store in place -- all lane-local) in pieces
    // that ride in the chunks of the following stage: value v = 8 b + 4 u + e in chunk A of tile v; the split of the
    // pair (v - 1, v) in chunk B of odd tiles; a batch tile's three plane stores behind its last pair (tiles 7, 15)
    float ev[8];                            // (one batch tile at a time: tile b's stores are issued before tile b + 1 starts)
    vec8 eo[NPL];
    auto epi_value = [&](int v, const f32x4 (&a)[2][2], const vec8 (&rr_)[2][NPL]) {
      const int b = v >> 3, u = (v >> 2) & 1, e = v & 3;
      if (USF_CPW_ABL & 64) { ev[4 * u + e] = a[u][b][e]; return; }
      const vec8 (&r)[NPL] = rr_[b];
      float rr = (float)r[0][4 * u + e] + (float)r[1][4 * u + e];
      if (NPL == 3) rr = rr + (float)r[NPL - 1][4 * u + e];
      ev[4 * u + e] = __builtin_fmaf(p.sign, a[u][b][e], rr);
      guard1(ev[4 * u + e]);
    };
    auto epi_pair = [&](int nt, int v) {              // v odd: values v - 1, v
      const int b = v >> 3, j = (v & 7) - 1;
      if (USF_CPW_ABL & 64) { if ((v & 7) == 7) { for (int q = 0; q < NPL; ++q) eo[q] = zp[b][q]; eo[0][0] = (decltype(eo[0][0] + eo[0][0]))ev[0]; store_zblk(p.kb_t0 + nt, b, eo); } return; }
      split_pair(ev[j], ev[j + 1], j, eo);
      if ((v & 7) == 7) store_zblk(p.kb_t0 + nt, b, eo);
    };
    // the operand planes of k-step ks (FIRST n-tile only): piece k = 4 b + jj as in phase 2
    auto xp_piece = [&](int ks, int k) {
      const int b = k >> 2, j = 2 * (k & 3);
      split_pair(act1(X[2 * ks + (j >> 2)][b][j & 3]), act1(X[2 * ks + (j >> 2)][b][(j & 3) + 1]), j, xp[ks][b]);
    };
    // one n-tile stage
    auto ntile_stage = [&](int nt, auto first_tag, f32x4 (&acc)[2][2], vec8 (&rnew)[2][NPL], const f32x4 (&accp)[2][2],
                           const vec8 (&rcur)[2][NPL]) {
      constexpr bool FIRST = decltype(first_tag)::value;
      const int rn = next_ring(ring);
      const unsigned x02 = 32u * (unsigned)min(nt + 2, p.nk_t - 1);
#pragma unroll
      for (int u = 0; u < 2; ++u) { acc[u][0] = bn[u]; acc[u][1] = bn[u]; }
#pragma unroll
      for (int i = 0; i < T; ++i) {
        const int ks = i >> 1, u = i & 1;
        const float* nx = (i + AH < T) ? tile_ptr(ring, i + AH, true) : tile_ptr(rn, i + AH - T, true);
        W32_HALF_A(i, nx, acc[u][0], acc[u][1], xp[ks][0], xp[ks][1]);
        if (i >= 1 && i <= 2 * NPL) zload_piece(p.kb_t0 + nt, i - 1, rnew);           // this n-tile's residual planes
        if (FIRST) { if (ks + 1 < KS) { xp_piece(ks + 1, 4 * u); xp_piece(ks + 1, 4 * u + 1); } }
        else epi_value(i, accp, rcur);
        if (i == 2 * NPL + 1) load_bias(nt + 1);
        pins(NMA, i > 0);
        fence();
        W32_HALF_B(i, acc[u][0], acc[u][1], xp[ks][0], xp[ks][1]);
        stage_side(i, rn, true, p.Wout, (unsigned)p.ld_out, 2u * (unsigned)p.pl_out, x02, true);
        if (FIRST) { if (ks + 1 < KS) { xp_piece(ks + 1, 4 * u + 2); xp_piece(ks + 1, 4 * u + 3); } }
        else if (i & 1) epi_pair(nt - 1, i);
        pins(NMB);
        fence();
        if (i == T / 2 - 1) mid_barrier();
      }
      ring = rn;
    };
    load_bias(0);
#pragma unroll
    for (int k = 0; k < 8; ++k) xp_piece(0, k);
    ntile_stage(0, CpBool<true>(), aA, rA, aB, rB);
    int nt = 1;
    for (; nt < p.nk_t; nt += 2) {
      ntile_stage(nt, CpBool<false>(), aB, rB, aA, rA);
      if (nt + 1 < p.nk_t) ntile_stage(nt + 1, CpBool<false>(), aA, rA, aB, rB);
    }
    // the last n-tile sits in set A when nk_t is odd, in set B when it is even
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      if (p.nk_t & 1) epi_value(v, aA, rA); else epi_value(v, aB, rB);
      if (v & 1) epi_pair(p.nk_t - 1, v);
    }
  };
  if (NH == 2) output_layer(X2); else output_layer(X1);
#undef W32_HALF_A
#undef W32_HALF_B
#ifdef USF_STAMP
  CSTAMP(c3);
  if (p.dbg && lane == 0) {
    unsigned long long* o = p.dbg + (size_t)((blockIdx.x % 2048) * 8 + wave) * 4;
    o[0] = c1 - c0; o[1] = c2 - c1; o[2] = c3 - c2; o[3] = 1;
#if USF_STAMP >= 2
    unsigned long long* o2 = p.dbg + (size_t)2048 * 8 * 4 + (size_t)((blockIdx.x % 2048) * 8 + wave) * 8;
    o2[0] = bw_mark[0]; o2[1] = bw_mark[1] - bw_mark[0]; o2[2] = bwait - bw_mark[1]; o2[3] = smin; o2[4] = smax;
#endif
  }
#endif
  // (rows beyond M load zeros, so a flag raised by such a row is a bias-only value beyond fp16's range: the pass is
  //  voided and redone in bf16x3, which is safe)
  if (NPL == 2 && p.range_flag && bad) atomicOr(p.range_flag, 1);
}

#ifdef USF_STAMP
unsigned long long* g_cdbg = nullptr;
#endif
int g_cp_w32 = -1;                        // usf_coupling_planes_select (-1: the environment's USF_CP_W32, default 0)

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
  static int w32_env = -1;
  if (w32_env < 0) { const char* e = getenv("USF_CP_W32"); w32_env = e ? atoi(e) : 0; }   // 1: the 32-row-wave kernel (usf_coupling_planes_select)
  const int w32 = g_cp_w32 >= 0 ? g_cp_w32 : w32_env;
  if (w32 && d->n_hidden <= 2 && d->nk_p >= 2 && 2 * npl * d->w_in_plane < (1LL << 31) && 2 * npl * d->w_hid_plane < (1LL << 31) &&
      2 * npl * d->w_out_plane < (1LL << 31) && npanels * 0 + d->z_nkb * npl * 1024 * 8 < (1LL << 31)) {
    const dim3 block(256);
#define USF_CPW(NPL_, NH_) hipLaunchKernelGGL((coupling_planes_w32_kernel<NPL_, NH_>), grid, block, 0, stream, a)
    if (npl == 2) { if (d->n_hidden == 1) USF_CPW(2, 1); else USF_CPW(2, 2); }
    else { if (d->n_hidden == 1) USF_CPW(3, 1); else USF_CPW(3, 2); }
#undef USF_CPW
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
