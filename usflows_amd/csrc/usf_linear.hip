// usf_linear_f32: fused dense layer  C = epi( pro(A) @ W^T )  on exact-f32 MFMA (gfx950).
//
// Design (DESIGN.md section "linear kernel"):
//  * v_mfma_f32_32x32x2_f32: MFMA-A = activations (i = batch row), MFMA-B = weights (j = output
//    feature), so the accumulator has the output feature on the lane -> coalesced C rows.
//  * k-permutation: for every 8 consecutive k, lane (i, h=lane>>5) owns k = 8q+4h .. 8q+4h+3 and
//    feeds them to 4 successive MFMAs.  Both operands are K-contiguous in memory (torch Linear
//    layout [N,K]; activations [M,K]), so every fragment is ONE 16-byte access per lane.
//  * waves tile M only (WM x 1): a wave owns its rows, so activation fragments go global ->
//    registers directly (prefetched one K-slab ahead, prologue applied in registers); only the
//    weight slab, shared by all waves, is staged through LDS (register-staged, double-buffered,
//    rows padded to an odd number of 16-B slots => conflict-free ds_read_b128).
//  * XCD-aware block map: the column-blocks of one row panel run on one XCD so the panel's
//    activations are served from that XCD's L2 after the first touch.
#include "usf_common.h"

namespace usf {

struct LinArgs {
  const float* A; const float* W; const float* bias; const float* pre_div; const float* pre_sub;
  const float* residual; const float* addend; const float* post_mul; float* C;
  int64_t lda, ldw, ldr, ldadd, ldc;
  int M, N, K;
  int nbm, nbn;
  float res_sign, slope;
  int act;
};

template <int TM, int TN, int WM, int BK>
__global__ __launch_bounds__(WM * 64, 2) void linear_kernel(const LinArgs p) {
  constexpr int NT = WM * 64;
  constexpr int BM = WM * TM * 32;
  constexpr int BN = TN * 32;
  constexpr int LDS_LD = BK + 4;            // odd number of 16-B slots per row
  constexpr int QS = BK / 8;                // 8-k steps per slab
  constexpr int WCH = BK / 4;               // float4 chunks per weight row per slab
  constexpr int NWV = (BN * WCH + NT - 1) / NT;  // float4 staged per thread per slab
  static_assert((LDS_LD / 4) % 2 == 1, "row stride must be an odd number of 16-B slots");

  __shared__ __attribute__((aligned(16))) float lds[2][BN * LDS_LD];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int li = lane & 31;
  const int lh = lane >> 5;

  // ---- block -> (row panel, column block); blocks b and b+8 share an XCD -----------------
  const int bid = blockIdx.x;
  const int xcd = bid & 7;
  const int seq = bid >> 3;
  const int panel = (seq / p.nbn) * 8 + xcd;
  const int bn = seq % p.nbn;
  if (panel >= p.nbm) return;               // uniform per block: no barrier was reached yet
  const int row0 = panel * BM + wave * (TM * 32);
  const int n0 = bn * BN;

  // ---- per-lane activation fragment sources -----------------------------------------------
  const float* aptr[TM];
  bool avalid[TM];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int r = row0 + tm * 32 + li;
    avalid[tm] = r < p.M;
    aptr[tm] = p.A + (int64_t)(avalid[tm] ? r : 0) * p.lda + 4 * lh;
  }
  const bool has_pro = (p.pre_div != nullptr) || (p.pre_sub != nullptr);

  auto load_a = [&](int k0, f32x4 (&dst)[TM][QS]) {
#pragma unroll
    for (int q = 0; q < QS; ++q) {
      const int k = k0 + 8 * q + 4 * lh;
      const bool kin = k < p.K;
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (kin && avalid[tm]) v = *reinterpret_cast<const f32x4*>(aptr[tm] + k0 + 8 * q);
        dst[tm][q] = v;
      }
      if (has_pro && kin) {
        if (p.pre_div) {
          const f32x4 d = *reinterpret_cast<const f32x4*>(p.pre_div + k);
#pragma unroll
          for (int tm = 0; tm < TM; ++tm) dst[tm][q] = dst[tm][q] / d;
        }
        if (p.pre_sub) {
          const f32x4 s = *reinterpret_cast<const f32x4*>(p.pre_sub + k);
#pragma unroll
          for (int tm = 0; tm < TM; ++tm) dst[tm][q] = dst[tm][q] - s;
        }
      }
    }
  };

  // ---- weight slab staging (global -> regs -> LDS) ----------------------------------------
  auto load_w = [&](int k0, f32x4 (&dst)[NWV]) {
#pragma unroll
    for (int i = 0; i < NWV; ++i) {
      const int idx = tid + i * NT;
      const int r = idx / WCH;
      const int c = idx % WCH;
      const int n = n0 + r;
      const int k = k0 + 4 * c;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (idx < BN * WCH && n < p.N && k < p.K)
        v = *reinterpret_cast<const f32x4*>(p.W + (int64_t)n * p.ldw + k);
      dst[i] = v;
    }
  };
  auto store_w = [&](int buf, const f32x4 (&src)[NWV]) {
#pragma unroll
    for (int i = 0; i < NWV; ++i) {
      const int idx = tid + i * NT;
      const int r = idx / WCH;
      const int c = idx % WCH;
      if (idx < BN * WCH) *reinterpret_cast<f32x4*>(&lds[buf][r * LDS_LD + 4 * c]) = src[i];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

  f32x4 a_cur[TM][QS], a_nxt[TM][QS];
  f32x4 wst[NWV];

  const int nslab = (p.K + BK - 1) / BK;
  load_w(0, wst);
  load_a(0, a_cur);
  store_w(0, wst);
  __syncthreads();

  for (int s = 0; s < nslab; ++s) {
    const int buf = s & 1;
    const bool more = (s + 1) < nslab;
    if (more) {
      load_w((s + 1) * BK, wst);
      load_a((s + 1) * BK, a_nxt);
    }
    const float* wl = &lds[buf][li * LDS_LD + 4 * lh];
#pragma unroll
    for (int q = 0; q < QS; ++q) {
      if (s * BK + 8 * q < p.K) {           // uniform: skip all-zero 8-k steps of the K tail
        f32x4 b[TN];
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
          b[tn] = *reinterpret_cast<const f32x4*>(wl + tn * 32 * LDS_LD + 8 * q);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
              acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[tm][q][t], b[tn][t], acc[tm][tn], 0, 0, 0);
      }
    }
    if (more) {
      store_w(buf ^ 1, wst);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int q = 0; q < QS; ++q) a_cur[tm][q] = a_nxt[tm][q];
    }
    __syncthreads();
  }

  // ---- epilogue: C layout of 32x32 f32: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) --
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int col = n0 + tn * 32 + li;
    if (col >= p.N) continue;
    const float bv = p.bias ? p.bias[col] : 0.f;
    const float pm = p.post_mul ? p.post_mul[col] : 1.f;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row0 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row < p.M) {
          float v = acc[tm][tn][r] + bv;
          if (p.addend) v = v + p.addend[(int64_t)row * p.ldadd + col];
          v = act_apply(v, p.act, p.slope);
          if (p.residual) v = p.residual[(int64_t)row * p.ldr + col] + p.res_sign * v;
          if (p.post_mul) v = v * pm;
          p.C[(int64_t)row * p.ldc + col] = v;
        }
      }
    }
  }
}

template <int TM, int TN, int WM, int BK>
static int launch_linear(const LinArgs& a0, hipStream_t stream) {
  LinArgs a = a0;
  constexpr int BM = WM * TM * 32, BN = TN * 32;
  a.nbm = (a.M + BM - 1) / BM;
  a.nbn = (a.N + BN - 1) / BN;
  const int64_t panels8 = ((int64_t)(a.nbm + 7) / 8) * 8;
  const int64_t grid = panels8 * a.nbn;
  if (grid > 0x7fffffffLL) { set_error("usf_linear_f32: grid too large"); return -3; }
  hipLaunchKernelGGL((linear_kernel<TM, TN, WM, BK>), dim3((unsigned)grid), dim3(WM * 64), 0, stream, a);
  return check_launch("usf_linear_f32");
}

int linear_dispatch(const usf_linear_desc* d, hipStream_t stream) {
  if (!d) { set_error("usf_linear_f32: null descriptor"); return -1; }
  if (d->M < 0 || d->N <= 0 || d->K <= 0 || d->M > 0x7fffffff || d->N > 0x7fffffff || d->K > 0x7fffffff) {
    set_error("usf_linear_f32: bad sizes M=%lld N=%lld K=%lld", (long long)d->M, (long long)d->N, (long long)d->K);
    return -2;
  }
  if (d->M == 0) return 0;
  if (!d->A || !d->W || !d->C) { set_error("usf_linear_f32: null A/W/C"); return -1; }
  if ((d->K & 3) || (d->lda & 3) || (d->ldw & 3) || d->lda < d->K || d->ldw < d->K || d->ldc < d->N ||
      (d->residual && d->ldr < d->N) || (d->addend && d->ldadd < d->N)) {
    set_error("usf_linear_f32: K/lda/ldw must be multiples of 4 and strides >= extents "
              "(K=%lld lda=%lld ldw=%lld ldc=%lld)", (long long)d->K, (long long)d->lda, (long long)d->ldw,
              (long long)d->ldc);
    return -2;
  }
  if (!aligned16(d->A) || !aligned16(d->W) || (d->pre_div && !aligned16(d->pre_div)) ||
      (d->pre_sub && !aligned16(d->pre_sub))) {
    set_error("usf_linear_f32: A/W/pre_div/pre_sub must be 16-byte aligned");
    return -2;
  }
  if (d->act != USF_ACT_NONE && d->act != USF_ACT_LEAKY_RELU) { set_error("usf_linear_f32: bad act"); return -2; }
  LinArgs a;
  a.A = d->A; a.W = d->W; a.bias = d->bias; a.pre_div = d->pre_div; a.pre_sub = d->pre_sub;
  a.residual = d->residual; a.addend = d->addend; a.post_mul = d->post_mul; a.C = d->C;
  a.lda = d->lda; a.ldw = d->ldw; a.ldr = d->ldr; a.ldadd = d->ldadd; a.ldc = d->ldc;
  a.M = (int)d->M; a.N = (int)d->N; a.K = (int)d->K;
  a.nbm = a.nbn = 0;
  a.res_sign = d->res_sign; a.slope = d->slope; a.act = d->act;

  // tile choice: 256-row panels; 160- or 128-wide column blocks, whichever pads N less
  // (784 -> 5 x 160 = 800; 256 -> 2 x 128). Small problems take the 64x64 tile.
  if (a.M <= 64 || a.N <= 64) return launch_linear<1, 2, 2, 16>(a, stream);
  const int pad160 = ((a.N + 159) / 160) * 160 - a.N;
  const int pad128 = ((a.N + 127) / 128) * 128 - a.N;
  if (pad160 < pad128) return launch_linear<2, 5, 4, 16>(a, stream);
  return launch_linear<2, 4, 4, 16>(a, stream);
}

}  // namespace usf
