// usf_linear_f32 for SMALL batches (M <= 768: measured cross-over against the narrow tiled kernel): the reference's evaluation loops feed the flow in chunks of 100
// (explib/hyperopt.py:273-278) and Flow.fit defaults to batch_size = 32 (flows.py:113).  At those sizes a layer is
// a weight-streaming problem -- 2 M flops per row against 2.4 MB of weights -- and what matters is the latency of
// one launch, not its FLOP rate: the tiled kernels put 5-13 blocks on the chip, each walking K in 25 serial,
// barrier-separated slabs (20-50 us per layer).  Here:
//
//   * one block per (16 output features, 64 batch rows): N = 784, M = 256 -> 49 x 4 = 196 blocks; no LDS staging,
//     no barrier in the K loop; per block 64 activation rows (200 KB from L2) and 16 weight rows (50 KB) pass the
//     CU's vector memory path -- with all rows in one block that path, not HBM, set the time;
//   * a wave owns 2 batch tiles of 16 rows and one of KS = 4 interleaved K ranges (in-block split-K: 2 x 4 waves),
//     so a wave's dependent MFMA chain is K / 16 x 2 long;
//   * exact-f32 MFMA (v_mfma_f32_16x16x4_f32) with the k-permutation of usf_linear.hip: lane group g owns 4
//     consecutive k, i.e. every operand fragment is one 16-byte global load (weights from HBM once per block,
//     activations from L2); loads of 4 k-steps (64 k) are issued together and double-buffered, so ~12 KB per wave are
//     in flight -- the whole weight matrix is on the wire within the first microsecond;
//   * the KS partial accumulators meet in LDS once, after the loop; the k-range-0 waves apply the epilogue
//     (bias / addend / LeakyReLU / residual / post_mul, same order as the tiled kernels) and store 4 consecutive
//     features per lane.
#include <stdlib.h>

#include "usf_common.h"

namespace usf {

struct SkArgs {
  const float* A; const float* W; const float* bias; const float* pre_div; const float* pre_sub;
  const float* residual; const float* addend; const float* post_mul; float* C;
  int64_t lda, ldw, ldr, ldadd, ldc;
  int M, N, K;
  int mw, ks;                  // waves along M (2 batch tiles each), K ranges
  float res_sign, slope;
  int act, vec_ok;
};

// SK_G: k-steps (of 16) fetched together (4; 8 is a tuning instance: USF_SKINNY_G)
template <int SK_G>
__global__ __launch_bounds__(512) void linear_skinny_kernel(const SkArgs p) {
  __shared__ f32x4 red[8][2][64];              // [wave][batch tile][lane] partial accumulators
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int wml = wave % p.mw, wk = wave / p.mw;
  const int wm = blockIdx.y * p.mw + wml;      // 32-row group of this wave
  const int n0 = blockIdx.x * 16;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const bool pro = p.pre_div != nullptr || p.pre_sub != nullptr;

  // operand rows: W row n0 + li (clamped), activation rows of the wave's two batch tiles (clamped)
  const float* wrow = p.W + (int64_t)min(n0 + li, p.N - 1) * p.ldw;
  const float* arow[2];
  bool tile_on[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int m = wm * 32 + t * 16 + li;
    arow[t] = p.A + (int64_t)min(m, p.M - 1) * p.lda;
    tile_on[t] = (wm * 32 + t * 16) < p.M;     // wave-uniform
  }
  f32x4 acc[2] = {zero4, zero4};

  // this wave's k-steps: s = wk, wk + ks, ...  (16 k each); a group = SK_G consecutive ones of them
  const int nstep = (p.K + 15) / 16;
  const int my_steps = (nstep - wk + p.ks - 1) / p.ks;
  // two register buffers with compile-time names (a runtime buffer index would push them to scratch)
  f32x4 w0[SK_G], a0[2][SK_G], w1[SK_G], a1[2][SK_G];
  auto fetch = [&](f32x4 (&wv)[SK_G], f32x4 (&av)[2][SK_G], int s0) {
#pragma unroll
    for (int u = 0; u < SK_G; ++u) {
      const int step = wk + (s0 + u) * p.ks;
      const int kq = step * 16 + 4 * lg;
      const bool ok = (s0 + u) < my_steps && kq < p.K;
      const int kc = ok ? kq : 0;
      const f32x4 w = *reinterpret_cast<const f32x4*>(wrow + kc);
      wv[u] = ok ? w : zero4;                  // (a zero weight fragment silences an out-of-range step)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        f32x4 a = zero4;
        if (tile_on[t]) {
          a = *reinterpret_cast<const f32x4*>(arow[t] + kc);
          if (pro) {
            if (p.pre_div) a = a / *reinterpret_cast<const f32x4*>(p.pre_div + kc);
            if (p.pre_sub) a = a - *reinterpret_cast<const f32x4*>(p.pre_sub + kc);
          }
        }
        av[t][u] = a;
      }
    }
  };
  auto compute = [&](const f32x4 (&wv)[SK_G], const f32x4 (&av)[2][SK_G]) {
#pragma unroll
    for (int u = 0; u < SK_G; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int t = 0; t < 2; ++t)
          if (tile_on[t]) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[u][e], av[t][u][e], acc[t], 0, 0, 0);
  };
  const int ngroup = (my_steps + SK_G - 1) / SK_G;
  if (ngroup > 0) fetch(w0, a0, 0);
  for (int gi = 0; gi < ngroup; gi += 2) {
    if (gi + 1 < ngroup) fetch(w1, a1, (gi + 1) * SK_G);
    compute(w0, a0);
    if (gi + 1 < ngroup) {
      if (gi + 2 < ngroup) fetch(w0, a0, (gi + 2) * SK_G);
      compute(w1, a1);
    }
  }

  // in-block reduction over the K ranges
  if (wk > 0) {
    red[wave][0][lane] = acc[0];
    red[wave][1][lane] = acc[1];
  }
  __syncthreads();
  if (wk > 0) return;
  for (int q = 1; q < p.ks; ++q) {
    acc[0] = acc[0] + red[q * p.mw + wml][0][lane];
    acc[1] = acc[1] + red[q * p.mw + wml][1][lane];
  }
  // accumulator layout (C^T tile): lane (j = li, g = lg) holds features n0 + 4 g + (0..3) of batch row 16 t + j
  const int col = n0 + 4 * lg;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int row = wm * 32 + t * 16 + li;
    if (!tile_on[t] || row >= p.M || col >= p.N) continue;
    f32x4 v = acc[t];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = min(col + j, p.N - 1);
      float x = v[j];
      if (p.bias) x += p.bias[c];
      if (p.act == USF_ACT_GATE) x = gate_apply(x, p.addend[(int64_t)row * p.ldadd + c], p.slope);
      else {
        if (p.addend) x += p.addend[(int64_t)row * p.ldadd + c];
        x = act_apply(x, p.act, p.slope);
      }
      if (p.residual) x = p.residual[(int64_t)row * p.ldr + c] + p.res_sign * x;
      if (p.post_mul) x *= p.post_mul[c];
      v[j] = x;
    }
    float* dst = p.C + (int64_t)row * p.ldc + col;
    if (p.vec_ok && col + 3 < p.N) {
      *reinterpret_cast<f32x4*>(dst) = v;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (col + j < p.N) dst[j] = v[j];
    }
  }
}

bool linear_skinny_eligible(const usf_linear_desc* d) {
  return d->M <= tuning("skinny_max", 768);     // tuning aid (cross-over against the tiled kernels)
}

int linear_skinny_dispatch(const usf_linear_desc* d, hipStream_t stream) {
  SkArgs a;
  a.A = d->A; a.W = d->W; a.bias = d->bias; a.pre_div = d->pre_div; a.pre_sub = d->pre_sub;
  a.residual = d->residual; a.addend = d->addend; a.post_mul = d->post_mul; a.C = d->C;
  a.lda = d->lda; a.ldw = d->ldw; a.ldr = d->ldr; a.ldadd = d->ldadd; a.ldc = d->ldc;
  a.M = (int)d->M; a.N = (int)d->N; a.K = (int)d->K;
  a.res_sign = d->res_sign; a.slope = d->slope; a.act = d->act;
  a.vec_ok = ((d->ldc & 3) == 0 && aligned16(d->C)) ? 1 : 0;
  a.mw = a.M > 32 ? 2 : 1;                                  // waves along M per block (64 rows)
  const int nstep = (a.K + 15) / 16;
  // waves per block <= 8.  (Cutting K in 8 for batches of <= 32 rows -- one fetch round per wave instead of three --
  // measured no faster: 8.8 vs 8.2 us per launch in Flow.fit at batch 32; USF_SKINNY_KS_SMALL: tuning aid)
  int ks_small = (int)tuning("skinny_ks_small", 4);
  if (ks_small < 1 || ks_small > 8) ks_small = 4;
  int ks = a.mw == 1 ? ks_small : 4;
  if (ks > nstep) ks = nstep;
  a.ks = ks;
  const dim3 grid((unsigned)((a.N + 15) / 16), (unsigned)((a.M + 32 * a.mw - 1) / (32 * a.mw)));
  const int g8 = tuning("skinny_g", 4) == 8 ? 1 : 0;
  if (g8) hipLaunchKernelGGL(linear_skinny_kernel<8>, grid, dim3(64 * a.mw * a.ks), 0, stream, a);
  else hipLaunchKernelGGL(linear_skinny_kernel<4>, grid, dim3(64 * a.mw * a.ks), 0, stream, a);
  return check_launch("usf_linear_f32(skinny)");
}

}  // namespace usf
