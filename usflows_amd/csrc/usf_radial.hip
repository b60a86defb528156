// The Lp-radial base density of the reference's live configurations in ONE pass, forward and backward:
//   RadialDistribution.log_prob (src/usflows/distributions.py:501-511):  r = ||z - loc||_p over the event dims,
//       logp = norm_distribution.log_prob(r) - log_delta_volume(p, r)                       (:513-549)
//   norm_distribution = LogNormal (:181-197, experiments/mnist/mnist.yaml:79-92, cifar/cifar.yaml) or GammaMM (:674-707,
//       experiments/fashion/fashionclasses_veriflow.yaml:79-93) -- generally a K-component mixture (K <= 64) of
//       torch LogNormal / Gamma components whose positive parameters are stored through softplus (:117-197, 730-795).
// HBM-bound: one wave per row reduces the radius with 16-byte lane loads (4 D bytes per sample), then lanes 0..K-1 of the
// same wave evaluate one mixture component each and a wave log-sum-exp finishes the row -- O(K) work per sample in fp64,
// no second launch, no host round trip (the torch formulation builds a validating distribution object per call, which
// reads a flag back to the host).  The backward kernel produces d/dz, and per-block partial sums of the gradients of
// loc-independent parameters (component parameters, mixture logits) that a finishing launch adds in a fixed order
// (bit-reproducible); d/dloc = -colsum(d/dz) reuses usf_colsum_f32's kernel.
#include "usf_common.h"

namespace usf {

int colsum(const float* Y, int64_t ldy, int64_t M, int64_t N, float* out, float alpha, float beta, float* workspace,
           int64_t workspace_floats, hipStream_t stream);

constexpr int RAD_MAXK = 64;
constexpr int RAD_WPB = 4;                 // waves (rows in flight) per block

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_max_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}

// torch.nn.functional.softplus (beta 1, threshold 20) and its derivative
__device__ __forceinline__ double softplus_d(double x) { return x > 20.0 ? x : log1p(exp(x)); }
__device__ __forceinline__ double softplus_grad_d(double x) { return x > 20.0 ? 1.0 : 1.0 / (1.0 + exp(-x)); }

// digamma for x > 0: upward recurrence to x >= 8, then the asymptotic series (|error| < 1e-14 there)
__device__ __forceinline__ double digamma_d(double x) {
  double acc = 0.0;
  while (x < 8.0) { acc -= 1.0 / x; x += 1.0; }
  const double i = 1.0 / x, i2 = i * i;
  return acc + log(x) - 0.5 * i -
         i2 * (1.0 / 12.0 - i2 * (1.0 / 120.0 - i2 * (1.0 / 252.0 - i2 * (1.0 / 240.0 - i2 * (1.0 / 132.0)))));
}

// per-component constants, one set per block in LDS:  lp_k(r) = c + ... (see comp_logp)
struct RadTab {
  double a[RAD_MAXK];      // LogNormal: mu            Gamma: concentration
  double b[RAD_MAXK];      // LogNormal: sigma         Gamma: rate
  double c[RAD_MAXK];      // LogNormal: -log sigma - log sqrt(2 pi) + log pi_k      Gamma: a log b - lgamma(a) + log pi_k
  double pi[RAD_MAXK];     // mixture weight softmax(logits)_k (1 when there are no logits)
};

// called by the first wave of a block (all 64 lanes); others wait at the barrier that follows
__device__ __forceinline__ void build_tab(RadTab& t, int norm, int K, const float* __restrict__ par_a,
                                          const float* __restrict__ par_b, const float* __restrict__ logits, int lane) {
  const bool raw = (norm & USF_NORM_RAW_PARAMS) != 0;
  const int kind = norm & 0xff;
  const bool on = lane < K;
  double lw = 0.0;
  if (logits) {                               // log_softmax over the K logits
    const double l = on ? (double)logits[lane] : -INFINITY;
    const double mx = wave_max_d(l);
    const double s = wave_sum_d(on ? exp(l - mx) : 0.0);
    lw = l - mx - log(s);
  }
  if (on) {
    const double pa = (double)par_a[lane], pb = (double)par_b[lane];
    double a, b, c;
    if (kind == USF_NORM_LOGNORMAL) {
      a = pa;
      b = raw ? pb : softplus_d(pb);
      c = -log(b) - 0.91893853320467274178 + lw;
    } else {
      a = raw ? pa : softplus_d(pa);
      b = raw ? pb : softplus_d(pb);
      c = a * log(b) - lgamma(a) + lw;
    }
    t.a[lane] = a; t.b[lane] = b; t.c[lane] = c; t.pi[lane] = exp(lw);
  }
}

// log of (mixture weight x component density) at radius r (lr = log r):
//   torch LogNormal.log_prob(r) = Normal(mu, sigma).log_prob(log r) - log r      (TransformedDistribution + ExpTransform)
//   torch Gamma.log_prob(r)     = xlogy(a, b) + xlogy(a - 1, r) - b r - lgamma(a)
__device__ __forceinline__ double comp_logp(const RadTab& t, int kind, int k, double r, double lr) {
  if (kind == USF_NORM_LOGNORMAL) {
    const double d = lr - t.a[k];
    return t.c[k] - d * d / (2.0 * t.b[k] * t.b[k]) - lr;
  }
  return t.c[k] + (t.a[k] - 1.0) * lr - t.b[k] * r;
}

template <int P_ID>
__device__ __forceinline__ float row_radius(const float* __restrict__ zr, const float* __restrict__ loc, int D, int D4, int lane) {
  float acc = 0.f;
  for (int d = lane * 4; d < D4; d += 256) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(zr + d);
    const f32x4 l = *reinterpret_cast<const f32x4*>(loc + d);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float t = v[j] - l[j];
      if (P_ID == USF_BASE_LPNORM2) acc += t * t;
      else if (P_ID == USF_BASE_LPNORM1) acc += fabsf(t);
      else acc = fmaxf(acc, fabsf(t));
    }
  }
  for (int d = D4 + lane; d < D; d += 64) {
    const float t = zr[d] - loc[d];
    if (P_ID == USF_BASE_LPNORM2) acc += t * t;
    else if (P_ID == USF_BASE_LPNORM1) acc += fabsf(t);
    else acc = fmaxf(acc, fabsf(t));
  }
  acc = (P_ID == USF_BASE_LPNORMINF) ? wave_max(acc) : wave_sum(acc);
  return (P_ID == USF_BASE_LPNORM2) ? sqrtf(acc) : acc;
}

template <int P_ID>
__global__ __launch_bounds__(64 * RAD_WPB) void radial_logprob_kernel(
    const float* __restrict__ z, int64_t ldz, int M, int D, const float* __restrict__ loc, int norm, int K,
    const float* __restrict__ par_a, const float* __restrict__ par_b, const float* __restrict__ logits, double logdv_const,
    float logdet_const, const double* __restrict__ logdet_dev, float* __restrict__ logp, float* __restrict__ r_out,
    double* __restrict__ sum_out) {
  __shared__ RadTab tab;
  __shared__ double part[RAD_WPB];
  const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
  if (wib == 0) build_tab(tab, norm, K, par_a, par_b, logits, lane);
  __syncthreads();
  const int kind = norm & 0xff;
  const double logdet = (double)logdet_const + (logdet_dev ? *logdet_dev : 0.0);
  const bool vec = ((ldz & 3) == 0) && ((reinterpret_cast<uintptr_t>(z) & 15u) == 0) && ((reinterpret_cast<uintptr_t>(loc) & 15u) == 0);
  const int D4 = vec ? (D & ~3) : 0;
  double block_sum = 0.0;
  for (int64_t row = (int64_t)blockIdx.x * RAD_WPB + wib; row < M; row += (int64_t)gridDim.x * RAD_WPB) {
    const float rf = z ? row_radius<P_ID>(z + row * ldz, loc, D, D4, lane) : r_out[row];      // (z == NULL: the radii are given)
    const double r = (double)rf, lr = log(r);
    double lp;
    if (K == 1) {
      lp = comp_logp(tab, kind, 0, r, lr);
    } else {
      const double v = lane < K ? comp_logp(tab, kind, lane, r, lr) : -INFINITY;
      const double mx = wave_max_d(v);
      lp = mx + log(wave_sum_d(lane < K ? exp(v - mx) : 0.0));
    }
    const float out = (float)(lp - (logdv_const + (double)(D - 1) * lr) + logdet);
    if (lane == 0) {
      logp[row] = out;
      if (r_out && z) r_out[row] = rf;
      block_sum += (double)out;
    }
  }
  if (sum_out != nullptr) {
    if (lane == 0) part[wib] = block_sum;
    __syncthreads();
    if (threadIdx.x == 0) {
      double t = 0.0;
      for (int w = 0; w < RAD_WPB; ++w) t += part[w];
      if (t != 0.0) atomicAdd(&sum_out[0], t);
    }
  }
}

__global__ void radial_add_count_kernel(double* sum_out, double n) { sum_out[1] += n; }

static bool radial_args_ok(const char* what, int64_t M, int64_t D, int64_t ldz, int32_t p_id, int32_t norm, int32_t K, bool has_z) {
  if (M < 0 || D <= 0 || M > 0x7fffffff || D > 0x7fffffff || (has_z && ldz < D)) { set_error("%s: bad sizes", what); return false; }
  if (p_id != USF_BASE_LPNORM1 && p_id != USF_BASE_LPNORM2 && p_id != USF_BASE_LPNORMINF) {
    set_error("%s: p id %d is not an Lp-radial id", what, p_id);
    return false;
  }
  const int kind = norm & 0xff;
  if ((kind != USF_NORM_LOGNORMAL && kind != USF_NORM_GAMMA) || (norm & ~(0xff | USF_NORM_RAW_PARAMS))) {
    set_error("%s: unknown norm-distribution id %d", what, norm);
    return false;
  }
  if (K < 1 || K > RAD_MAXK) { set_error("%s: K = %d components (1..%d served)", what, K, RAD_MAXK); return false; }
  return true;
}

int radial_logprob(const float* z, int64_t ldz, int64_t M, int64_t D, int32_t p_id, const float* loc, int32_t norm, int32_t K,
                   const float* par_a, const float* par_b, const float* logits, double logdv_const, float logdet_const,
                   const double* logdet_dev, float* logp, float* r_out, double* sum_out, hipStream_t stream) {
  if (!radial_args_ok("usf_radial_logprob_f32", M, D, ldz, p_id, norm, K, z != nullptr)) return -2;
  if (M == 0) return 0;
  if ((z && !loc) || (!z && !r_out) || !par_a || !par_b || !logp || (K > 1 && !logits)) { set_error("usf_radial_logprob_f32: null pointer"); return -1; }
  int64_t blocks = (M + RAD_WPB - 1) / RAD_WPB;
  if (blocks > 256 * 8) blocks = 256 * 8;
  dim3 g((unsigned)blocks), b(64 * RAD_WPB);
#define USF_LAUNCH_RAD(P)                                                                                                  \
  hipLaunchKernelGGL((radial_logprob_kernel<P>), g, b, 0, stream, z, ldz, (int)M, (int)D, loc, (int)norm, (int)K, par_a, par_b, \
                     logits, logdv_const, logdet_const, logdet_dev, logp, r_out, sum_out)
  if (p_id == USF_BASE_LPNORM1) USF_LAUNCH_RAD(USF_BASE_LPNORM1);
  else if (p_id == USF_BASE_LPNORM2) USF_LAUNCH_RAD(USF_BASE_LPNORM2);
  else USF_LAUNCH_RAD(USF_BASE_LPNORMINF);
#undef USF_LAUNCH_RAD
  int rc = check_launch("usf_radial_logprob_f32");
  if (rc) return rc;
  if (sum_out) {
    hipLaunchKernelGGL(radial_add_count_kernel, dim3(1), dim3(1), 0, stream, sum_out, (double)M);
    rc = check_launch("usf_radial_logprob_f32(count)");
  }
  return rc;
}

// ---- backward ---------------------------------------------------------------------------------------------------------------
// per row, with g = g_lp[m], w_k = the posterior weight of component k at r (softmax_k of comp_logp):
//   dlogp/dr      = sum_k w_k dlp_k/dr - (D - 1) / r
//   dlogp/dtheta_k = w_k dlp_k/dtheta_k,      dlogp/dlogit_k = w_k - pi_k
//   dr/dz_d: p = 1: sign(t)   p = 2: t / r   p = inf: sign(t) where |t| == r        (t = z_d - loc_d; ATen's norm backward)
// part [blocks][3][RAD_MAXK] doubles: the block's sums of g * dlogp/d(a_k, b_k, logit_k) over its rows.
template <int P_ID>
__global__ __launch_bounds__(64 * RAD_WPB) void radial_grad_kernel(
    const float* __restrict__ z, int64_t ldz, const float* __restrict__ r_in, const float* __restrict__ g_lp, int M, int D,
    const float* __restrict__ loc, int norm, int K, const float* __restrict__ par_a, const float* __restrict__ par_b,
    const float* __restrict__ logits, float* __restrict__ g, int64_t ldg, double* __restrict__ part) {
  __shared__ RadTab tab;
  __shared__ double dig[RAD_MAXK];
  __shared__ double red[RAD_WPB][3][RAD_MAXK];
  const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
  const int kind = norm & 0xff;
  if (wib == 0) {
    build_tab(tab, norm, K, par_a, par_b, logits, lane);
    if (lane < K && kind == USF_NORM_GAMMA) dig[lane] = digamma_d(tab.a[lane]);
  }
  __syncthreads();
  const bool vec = ((ldz & 3) == 0) && ((ldg & 3) == 0) && ((reinterpret_cast<uintptr_t>(z) & 15u) == 0) &&
                   ((reinterpret_cast<uintptr_t>(loc) & 15u) == 0) && ((reinterpret_cast<uintptr_t>(g) & 15u) == 0);
  const int D4 = vec ? (D & ~3) : 0;
  double acc_a = 0.0, acc_b = 0.0, acc_l = 0.0;
  const bool on = lane < K;
  for (int64_t row = (int64_t)blockIdx.x * RAD_WPB + wib; row < M; row += (int64_t)gridDim.x * RAD_WPB) {
    const float rf = r_in[row];
    const double r = (double)rf, lr = log(r), gm = (double)g_lp[row];
    double w = 1.0;
    if (K > 1) {
      const double v = on ? comp_logp(tab, kind, lane, r, lr) : -INFINITY;
      const double mx = wave_max_d(v);
      const double e = on ? exp(v - mx) : 0.0;
      w = e / wave_sum_d(e);
    }
    double dr_k = 0.0;
    if (on) {
      if (kind == USF_NORM_LOGNORMAL) {
        const double s = tab.b[lane], d = lr - tab.a[lane], is2 = 1.0 / (s * s);
        dr_k = w * (-d * is2 - 1.0) / r;
        acc_a += gm * w * d * is2;                               // d/dmu
        acc_b += gm * w * (d * d * is2 / s - 1.0 / s);           // d/dsigma
      } else {
        const double a = tab.a[lane], b = tab.b[lane];
        dr_k = w * ((a - 1.0) / r - b);
        acc_a += gm * w * (log(b) + lr - dig[lane]);             // d/dconcentration
        acc_b += gm * w * (a / b - r);                           // d/drate
      }
      acc_l += gm * (w - tab.pi[lane]);
    }
    const double dlp_dr = (K > 1 ? wave_sum_d(dr_k) : __shfl(dr_k, 0, 64)) - (double)(D - 1) / r;
    const float coef = (float)(gm * dlp_dr);
    if (!z) {                                    // radii given: the gradient at r
      if (lane == 0) g[row] = coef;
      continue;
    }
    const float inv_r = rf > 0.f ? 1.0f / rf : 0.f;
    const float* zr = z + row * ldz;
    float* gr = g + row * ldg;
    float coef_inf = coef;
    if (P_ID == USF_BASE_LPNORMINF) {
      // x.norm(p=inf) (distributions.py:506) shares the gradient EVENLY among tied maxima (ATen's norm backward divides by their
      // number): one more pass over the row counts them (quantised or clamped latents; a row of distinct values counts 1)
      int cnt = 0;
      for (int d = lane; d < D; d += 64) cnt += fabsf(zr[d] - loc[d]) == rf ? 1 : 0;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
      coef_inf = coef / (float)(cnt > 0 ? cnt : 1);
    }
    for (int d = lane * 4; d < D4; d += 256) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(zr + d);
      const f32x4 l = *reinterpret_cast<const f32x4*>(loc + d);
      f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float t = v[j] - l[j];
        const float sg = (float)((t > 0.f) - (t < 0.f));
        if (P_ID == USF_BASE_LPNORM1) o[j] = coef * sg;
        else if (P_ID == USF_BASE_LPNORM2) o[j] = coef * (t * inv_r);
        else o[j] = (fabsf(t) == rf) ? coef_inf * sg : 0.f;
      }
      *reinterpret_cast<f32x4*>(gr + d) = o;
    }
    for (int d = D4 + lane; d < ldg; d += 64) {
      float o = 0.f;
      if (d < D) {
        const float t = zr[d] - loc[d];
        const float sg = (float)((t > 0.f) - (t < 0.f));
        if (P_ID == USF_BASE_LPNORM1) o = coef * sg;
        else if (P_ID == USF_BASE_LPNORM2) o = coef * (t * inv_r);
        else o = (fabsf(t) == rf) ? coef_inf * sg : 0.f;
      }
      gr[d] = o;
    }
  }
  red[wib][0][lane] = acc_a; red[wib][1][lane] = acc_b; red[wib][2][lane] = acc_l;
  __syncthreads();
  if (threadIdx.x < 3 * RAD_MAXK) {
    const int q = threadIdx.x / RAD_MAXK, k = threadIdx.x % RAD_MAXK;
    double t = 0.0;
    for (int w = 0; w < RAD_WPB; ++w) t += red[w][q][k];
    part[((int64_t)blockIdx.x * 3 + q) * RAD_MAXK + k] = t;
  }
}

// the blocks' partial sums added in a fixed order (one wave per output: lane l adds blocks l, l + 64, ..., then a shuffle tree);
// chain rule through softplus to the stored parameters.  grid = 3 * K waves.  (A single thread per output walking up to 1024
// partials one dependent load at a time took 0.25 ms at 4096 rows -- longer than the pass over the batch.)
__global__ __launch_bounds__(64) void radial_grad_finish_kernel(const double* __restrict__ part, int blocks, int norm, int K,
                                                                const float* __restrict__ par_a, const float* __restrict__ par_b,
                                                                float* __restrict__ d_a, float* __restrict__ d_b,
                                                                float* __restrict__ d_logits) {
  const int q = blockIdx.x / K, k = blockIdx.x - q * K;
  const int lane = threadIdx.x;
  double t = 0.0;
  for (int b = lane; b < blocks; b += 64) t += part[((int64_t)b * 3 + q) * RAD_MAXK + k];
  t = wave_sum_d(t);
  if (lane != 0) return;
  const bool raw = (norm & USF_NORM_RAW_PARAMS) != 0;
  const int kind = norm & 0xff;
  if (q == 0) {
    if (!raw && kind == USF_NORM_GAMMA) t *= softplus_grad_d((double)par_a[k]);
    if (d_a) d_a[k] = (float)t;
  } else if (q == 1) {
    if (!raw) t *= softplus_grad_d((double)par_b[k]);
    if (d_b) d_b[k] = (float)t;
  } else if (d_logits) {
    d_logits[k] = (float)t;
  }
}

static int64_t radial_grad_blocks(int64_t M) {
  int64_t blocks = (M + RAD_WPB - 1) / RAD_WPB;
  if (blocks > 1024) blocks = 1024;
  return blocks < 1 ? 1 : blocks;
}

// workspace layout (bytes): [blocks * 3 * RAD_MAXK doubles | colsum workspace floats]
static int64_t colsum_ws_floats(int64_t M, int64_t N) { return ((M + 255) / 256 + (M + 65535) / 65536 + 2) * N; }

int64_t radial_grad_workspace(int64_t M, int64_t D) {
  if (M < 0 || D <= 0) return 0;
  return radial_grad_blocks(M) * 3 * RAD_MAXK * (int64_t)sizeof(double) + colsum_ws_floats(M, D) * (int64_t)sizeof(float);
}

int radial_grad(const float* z, int64_t ldz, const float* r, const float* g_lp, int64_t M, int64_t D, int32_t p_id,
                const float* loc, int32_t norm, int32_t K, const float* par_a, const float* par_b, const float* logits, float* g,
                int64_t ldg, float* d_loc, float* d_a, float* d_b, float* d_logits, void* workspace, int64_t workspace_bytes,
                hipStream_t stream) {
  if (!radial_args_ok("usf_radial_logprob_grad_f32", M, D, ldz, p_id, norm, K, z != nullptr)) return -2;
  if (z && ldg < D) { set_error("usf_radial_logprob_grad_f32: ldg < D"); return -2; }
  if (M == 0) {                     // an empty batch: the parameter gradients are zeros
    if (d_loc) (void)hipMemsetAsync(d_loc, 0, (size_t)D * sizeof(float), stream);
    if (d_a) (void)hipMemsetAsync(d_a, 0, (size_t)K * sizeof(float), stream);
    if (d_b) (void)hipMemsetAsync(d_b, 0, (size_t)K * sizeof(float), stream);
    if (d_logits) (void)hipMemsetAsync(d_logits, 0, (size_t)K * sizeof(float), stream);
    return check_launch("usf_radial_logprob_grad_f32(empty)");
  }
  if ((z && !loc) || (!z && d_loc) || !r || !g_lp || !par_a || !par_b || !g || (K > 1 && !logits) || !workspace) {
    set_error("usf_radial_logprob_grad_f32: null pointer");
    return -1;
  }
  if (workspace_bytes < radial_grad_workspace(M, D) || (reinterpret_cast<uintptr_t>(workspace) & 7u)) {
    set_error("usf_radial_logprob_grad_f32: workspace too small or misaligned");
    return -4;
  }
  const int64_t blocks = radial_grad_blocks(M);
  double* part = reinterpret_cast<double*>(workspace);
  dim3 gdim((unsigned)blocks), b(64 * RAD_WPB);
#define USF_LAUNCH_RADG(P)                                                                                                     \
  hipLaunchKernelGGL((radial_grad_kernel<P>), gdim, b, 0, stream, z, ldz, r, g_lp, (int)M, (int)D, loc, (int)norm, (int)K, par_a, \
                     par_b, logits, g, ldg, part)
  if (p_id == USF_BASE_LPNORM1) USF_LAUNCH_RADG(USF_BASE_LPNORM1);
  else if (p_id == USF_BASE_LPNORM2) USF_LAUNCH_RADG(USF_BASE_LPNORM2);
  else USF_LAUNCH_RADG(USF_BASE_LPNORMINF);
#undef USF_LAUNCH_RADG
  int rc = check_launch("usf_radial_logprob_grad_f32");
  if (rc) return rc;
  if (d_a || d_b || d_logits) {
    hipLaunchKernelGGL(radial_grad_finish_kernel, dim3((unsigned)(3 * K)), dim3(64), 0, stream, part, (int)blocks, (int)norm, (int)K,
                       par_a, par_b, d_a, d_b, d_logits);
    rc = check_launch("usf_radial_logprob_grad_f32(finish)");
    if (rc) return rc;
  }
  if (d_loc) {
    float* cws = reinterpret_cast<float*>(part + blocks * 3 * RAD_MAXK);
    rc = colsum(g, ldg, M, D, d_loc, -1.0f, 0.0f, cws, colsum_ws_floats(M, D), stream);
  }
  return rc;
}

}  // namespace usf
