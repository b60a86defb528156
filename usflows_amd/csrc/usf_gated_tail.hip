// The tail of a GatedConv layer of ConvNet2D at TRAINING batches (the reference's 32 rows: 1 568 pixels of the live MNIST
// configuration) -- forward and backward each ONE launch:
//
//     a = in_act(h)                       h = the layer's 3 x 3 convolution (networks.py:108-122)
//     [val, gate] = W a + bias            the layer's 1 x 1 convolution, W [2 C, C]
//     r = x + val * sigmoid(gate)         x = the layer's input (the skip connection)
//     y = LayerNormChannels(post_act(r))  the nonlinearity and layer norm ConvNet2D puts behind the layer (networks.py:40-58,
//                                         480-493); without a layer norm y = r
//
// The step at such a batch is a chain of ~1 000 dependent launches of ~5 us: what counts is the NUMBER of launches.  The chain
// of round 4 spent 3 launches on this tail in the forward pass (pointwise convolution, gated residual, layer norm) and 5 in the
// backward pass (layer norm, its parameter sums, gated residual, a transposed copy of W, pointwise data gradient).  The backward
// kernel here recomputes the forward from (h, x) -- val / gate / r are never stored -- and writes dx (skip branch), dh
// (through W^T and in_act') and per-block partial sums of ALL parameter gradients: dW = sum over pixels of d[val, gate] (x) a
// (the block's 32 pixels meet in LDS; C a multiple of 16: 16 x 16 tiles on v_mfma_f32_16x16x4_f32, else thread t owns the
// entries e = t, t + 256, ... and adds the pixels in ascending order), dbias, dgamma, dbeta -- one slot of 2 C C + 4 C floats per block (last sum: usf_partial_sum_jobs_f32 or at once).
// d[val, gate] itself is written only on request (tests).
//
// Few pixels, so EIGHT lanes share a pixel (lane = 8 * channel group + pixel of the wave's 8): channel group cg owns channels
// cg, cg + 8, ...; each lane holds all C values of a (the same loads in the 8 lanes of a pixel hit L1), walks its 2 C / 8 rows
// of W from LDS (row stride C + 4 floats: the 8 rows of a wave instruction start 4 banks apart -- conflict-free b128 reads),
// the layer norm's sums cross the 8 lanes by 3 shuffles, and dh = W^T d[val, gate] is a reduce-scatter over them (16 + 8 + 4
// values for C = 32).  Exact fp32 (fmaf chains in ascending channel order, IEEE division, expf).
#include "usf_common.h"

namespace usf {

// the last sum of per-block slots (usf_image_grad.hip): now (job == NULL) or as jobs of usf_partial_sum_jobs_f32
int sum_slots(const float* part, int nparts, int n, float* scratch, float* out, usf_psum_job* job, hipStream_t stream, const char* what);
int64_t sum_slots_scratch(int64_t n);

struct GtArgs {
  const float* h; const float* x; const float* W; const float* bias; const float* gamma; const float* beta; const float* dy;
  float* y; float* dx; float* dh; float* dvg; float* part;
  int64_t BP, P;
  float eps, in_slope, post_slope;      // slope 1 = no nonlinearity
  int slot_floats;                      // backward: 2 C C + 2 C (+ 2 C with a layer norm) floats per block in part
};

__device__ __forceinline__ float leaky(float v, float slope) { return v > 0.f ? v : v * slope; }

template <int C, bool LN, bool BWD>
__global__ __launch_bounds__(256) void gated_tail_kernel(const GtArgs a) {
  constexpr int NC = C / 8, WS = C + 4;
  __shared__ __attribute__((aligned(16))) float wl[2 * C * WS];
  __shared__ float red[BWD && LN ? 4 * 2 * C : 1];
  __shared__ float sa[BWD ? 32 * C : 1];                       // a of the block's 32 pixels
  __shared__ float sd[BWD ? 32 * 2 * C : 1];                   // d[val, gate] of the block's 32 pixels
  // entries of dW per thread: C % 16 == 0: the 4 accumulators of each 16 x 16 tile its wave owns; else e = t, t + 256, ...
  constexpr int NE = C % 16 == 0 ? 4 * (((2 * C / 16) * (C / 16) + 3) / 4) : (2 * C * C + 255) / 256;
  float dwacc[BWD ? NE : 1], dbacc = 0.f;
#pragma unroll
  for (int i = 0; i < (BWD ? NE : 1); ++i) dwacc[i] = 0.f;
  for (int i = threadIdx.x; i < 2 * C * C; i += 256) {
    const int r = i / C, c = i - r * C;
    wl[r * WS + c] = a.W[i];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int pi = lane & 7, cg = lane >> 3;
  float gam[NC], bet[NC], bv[NC], bg[NC], dgam[NC], dbet[NC];
#pragma unroll
  for (int k = 0; k < NC; ++k) {
    const int c = cg + 8 * k;
    gam[k] = LN ? a.gamma[c] : 1.f;
    bet[k] = LN ? a.beta[c] : 0.f;
    bv[k] = a.bias ? a.bias[c] : 0.f;
    bg[k] = a.bias ? a.bias[C + c] : 0.f;
    dgam[k] = dbet[k] = 0.f;
  }
  const float inv_c = 1.f / (float)C;
  // block-uniform trip count (the backward's waves meet at barriers): a wave whose 8 pixels lie past the end works on a
  // clamped pixel with zero gradients and stores nothing
  for (int64_t blk = blockIdx.x; blk * 32 < a.BP; blk += gridDim.x) {
    const int64_t i0 = (blk * 4 + wave) * 8 + pi;
    const bool on = i0 < a.BP;
    const int64_t i = on ? i0 : a.BP - 1;
    const int64_t b = i / a.P, p = i - b * a.P;
    const float* hb = a.h + b * C * a.P + p;
    const float* xb = a.x + b * C * a.P + p;
    float av[C], ho[NC], xo[NC], d[NC];
#pragma unroll
    for (int ci = 0; ci < C; ++ci) av[ci] = hb[(int64_t)ci * a.P];
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      const int64_t off = (int64_t)(cg + 8 * k) * a.P;
      ho[k] = hb[off];
      xo[k] = xb[off];
      if (BWD) d[k] = on ? a.dy[b * C * a.P + p + off] : 0.f;
    }
#pragma unroll
    for (int ci = 0; ci < C; ++ci) av[ci] = leaky(av[ci], a.in_slope);
    float val[NC], sg[NC], r[NC], nh[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      const float* wv_ = wl + (cg + 8 * k) * WS;
      const float* wg_ = wl + (C + cg + 8 * k) * WS;
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int ci = 0; ci < C; ci += 4) {
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(wv_ + ci);
        const f32x4 w1 = *reinterpret_cast<const f32x4*>(wg_ + ci);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          s0 = fmaf(w0[u], av[ci + u], s0);
          s1 = fmaf(w1[u], av[ci + u], s1);
        }
      }
      val[k] = s0 + bv[k];
      sg[k] = 1.f / (1.f + expf(-(s1 + bg[k])));
      r[k] = xo[k] + val[k] * sg[k];
    }
    float rden = 1.f;
    if (LN) {
      float sum = 0.f;
#pragma unroll
      for (int k = 0; k < NC; ++k) {
        nh[k] = leaky(r[k], a.post_slope);
        sum += nh[k];
      }
#pragma unroll
      for (int o = 8; o < 64; o <<= 1) sum += __shfl_xor(sum, o, 64);
      const float mean = sum * inv_c;
      float sq = 0.f;
#pragma unroll
      for (int k = 0; k < NC; ++k) {
        nh[k] -= mean;
        sq += nh[k] * nh[k];
      }
#pragma unroll
      for (int o = 8; o < 64; o <<= 1) sq += __shfl_xor(sq, o, 64);
      const float den = sqrtf(sq * inv_c + a.eps);
      rden = 1.f / den;
#pragma unroll
      for (int k = 0; k < NC; ++k) nh[k] = nh[k] / den;
    }
    if (!BWD) {
      float* yb = a.y + b * C * a.P + p;
#pragma unroll
      for (int k = 0; k < NC; ++k)
        if (on) yb[(int64_t)(cg + 8 * k) * a.P] = LN ? nh[k] * gam[k] + bet[k] : r[k];
      continue;
    }
    // ---- backward ----
    float dr[NC];
    if (LN) {
      float m1 = 0.f, m2 = 0.f;
#pragma unroll
      for (int k = 0; k < NC; ++k) {
        const float g = d[k] * gam[k];
        m1 += g;
        m2 += g * nh[k];
        dgam[k] += d[k] * nh[k];
        dbet[k] += d[k];
        dr[k] = g;
      }
#pragma unroll
      for (int o = 8; o < 64; o <<= 1) {
        m1 += __shfl_xor(m1, o, 64);
        m2 += __shfl_xor(m2, o, 64);
      }
      m1 *= inv_c;
      m2 *= inv_c;
#pragma unroll
      for (int k = 0; k < NC; ++k) {
        const float t = (dr[k] - m1 - nh[k] * m2) * rden;
        dr[k] = r[k] > 0.f ? t : t * a.post_slope;
      }
    } else {
#pragma unroll
      for (int k = 0; k < NC; ++k) dr[k] = d[k];
    }
    float dv[NC], dgt[NC];
    {
      float* dxb = a.dx + b * C * a.P + p;
      float* dvb = a.dvg + b * 2 * C * a.P + p;
#pragma unroll
      for (int k = 0; k < NC; ++k) {
        dv[k] = dr[k] * sg[k];
        dgt[k] = dr[k] * val[k] * (sg[k] * (1.f - sg[k]));
        if (on) {
          const int64_t off = (int64_t)(cg + 8 * k) * a.P;
          dxb[off] = dr[k];
          if (a.dvg) {
            dvb[off] = dv[k];
            dvb[off + (int64_t)C * a.P] = dgt[k];
          }
        }
        const int px = wave * 8 + pi;
        sa[px * C + cg + 8 * k] = leaky(ho[k], a.in_slope);
        sd[px * 2 * C + cg + 8 * k] = dv[k];                   // (zero for a pixel past the end: d == 0)
        sd[px * 2 * C + C + cg + 8 * k] = dgt[k];
      }
    }
    __syncthreads();
    // dW[co][ci] += d[val, gate][px][co] * a[px][ci] over the block's 32 pixels
    if constexpr (C % 16 == 0) {
      // on the fp32 matrix instruction (exact fp32 products and sums): tile (cot, cit) = 16 rows of d[val, gate] x 16 channels
      // of a, K = the pixels four at a time; wave w owns the tiles w, w + 4, ...
      constexpr int NT = (2 * C / 16) * (C / 16);
#pragma unroll
      for (int ti = 0; ti < (NT + 3) / 4; ++ti) {
        const int t = wave + 4 * ti;
        if (t < NT) {
          const int cot = t / (C / 16), cit = t - cot * (C / 16);
          f32x4 acc = {dwacc[4 * ti], dwacc[4 * ti + 1], dwacc[4 * ti + 2], dwacc[4 * ti + 3]};
#pragma unroll
          for (int k0 = 0; k0 < 32; k0 += 4) {
            const int px = k0 + (lane >> 4);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(sd[px * 2 * C + cot * 16 + (lane & 15)], sa[px * C + cit * 16 + (lane & 15)],
                                                       acc, 0, 0, 0);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) dwacc[4 * ti + r] = acc[r];
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < NE; ++i) {
        const int e = threadIdx.x + 256 * i;
        if (e < 2 * C * C) {
          const int co = e / C, ci = e - co * C;
          for (int px = 0; px < 32; ++px) dwacc[i] = fmaf(sd[px * 2 * C + co], sa[px * C + ci], dwacc[i]);
        }
      }
    }
    if (threadIdx.x < 2 * C)
      for (int px = 0; px < 32; ++px) dbacc += sd[px * 2 * C + threadIdx.x];
    __syncthreads();                                            // (the next round's pixels overwrite sa / sd)
    // da[ci] = sum over the 2 C rows of W[row][ci] * d[val, gate][row]: this lane's 2 NC rows, then a reduce-scatter over
    // the 8 channel groups (fixed order: deterministic); av is reused for the partial sums
#pragma unroll
    for (int ci = 0; ci < C; ++ci) av[ci] = 0.f;
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      const float* wv_ = wl + (cg + 8 * k) * WS;
      const float* wg_ = wl + (C + cg + 8 * k) * WS;
#pragma unroll
      for (int ci = 0; ci < C; ci += 4) {
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(wv_ + ci);
        const f32x4 w1 = *reinterpret_cast<const f32x4*>(wg_ + ci);
#pragma unroll
        for (int u = 0; u < 4; ++u) av[ci + u] = fmaf(w1[u], dgt[k], fmaf(w0[u], dv[k], av[ci + u]));
      }
    }
    // channel ci = j + 8 k belongs to group j: stage 1 pairs the groups that differ in bit 2 (lanes 32 apart), ...
    const bool b2 = (cg & 4) != 0, b1 = (cg & 2) != 0, b0 = (cg & 1) != 0;
    float q[4 * NC], q2[2 * NC], da[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const float lo = av[jj + 8 * k], hi = av[jj + 4 + 8 * k];
        q[jj + 4 * k] = (b2 ? hi : lo) + __shfl_xor(b2 ? lo : hi, 32, 64);
      }
#pragma unroll
    for (int k = 0; k < NC; ++k)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const float lo = q[jj + 4 * k], hi = q[jj + 2 + 4 * k];
        q2[jj + 2 * k] = (b1 ? hi : lo) + __shfl_xor(b1 ? lo : hi, 16, 64);
      }
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      const float lo = q2[2 * k], hi = q2[1 + 2 * k];
      da[k] = (b0 ? hi : lo) + __shfl_xor(b0 ? lo : hi, 8, 64);
    }
    float* dhb = a.dh + b * C * a.P + p;
#pragma unroll
    for (int k = 0; k < NC; ++k)
      if (on) dhb[(int64_t)(cg + 8 * k) * a.P] = ho[k] > 0.f ? da[k] : da[k] * a.in_slope;
  }
  if (BWD) {
    float* slot = a.part + (int64_t)blockIdx.x * a.slot_floats;
    if constexpr (C % 16 == 0) {
      constexpr int NT = (2 * C / 16) * (C / 16);
#pragma unroll
      for (int ti = 0; ti < (NT + 3) / 4; ++ti) {
        const int t = wave + 4 * ti;
        if (t < NT) {
          const int cot = t / (C / 16), cit = t - cot * (C / 16);
#pragma unroll
          for (int r = 0; r < 4; ++r) slot[(cot * 16 + 4 * (lane >> 4) + r) * C + cit * 16 + (lane & 15)] = dwacc[4 * ti + r];
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < NE; ++i) {
        const int e = threadIdx.x + 256 * i;
        if (e < 2 * C * C) slot[e] = dwacc[i];
      }
    }
    if (threadIdx.x < 2 * C) slot[2 * C * C + threadIdx.x] = dbacc;
  }
  if (BWD && LN) {
    // dgamma / dbeta: over the wave's 8 pixels, then over the block's 4 waves
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      float g = dgam[k], bt = dbet[k];
#pragma unroll
      for (int o = 1; o < 8; o <<= 1) {
        g += __shfl_xor(g, o, 64);
        bt += __shfl_xor(bt, o, 64);
      }
      if (pi == 0) {
        red[wave * 2 * C + cg + 8 * k] = g;
        red[wave * 2 * C + C + cg + 8 * k] = bt;
      }
    }
    __syncthreads();
    if (threadIdx.x < 2 * C) {
      const int j = threadIdx.x;
      a.part[(int64_t)blockIdx.x * a.slot_floats + 2 * C * C + 2 * C + j] = ((red[j] + red[2 * C + j]) + red[4 * C + j]) + red[6 * C + j];
    }
  }
}

// (C = 48, 64 would hold 2 x 48 / 64 live values per lane next to the weight rows in flight: spills; those widths keep the chain)
static bool gated_tail_c_ok(int64_t C) { return C == 8 || C == 16 || C == 24 || C == 32; }

static int gated_tail_blocks(int64_t BP) {
  int64_t blocks = (BP + 31) / 32;                 // 4 waves x 8 pixels
  return (int)(blocks > 2048 ? 2048 : blocks);
}

int gated_tail_supported(int64_t C) { return gated_tail_c_ok(C) ? 1 : 0; }

// floats of the parameter gradients [dW (2 C C) | dbias (2 C) | dgamma (C) | dbeta (C)] (the last two with a layer norm)
static int64_t gated_tail_nparams(int64_t C, bool ln) { return 2 * C * C + 2 * C + (ln ? 2 * C : 0); }

int64_t gated_tail_workspace(int64_t B, int64_t C, int64_t P) {
  if (B <= 0 || P <= 0 || !gated_tail_c_ok(C)) return 0;
  const int64_t n = gated_tail_nparams(C, true);
  return (int64_t)gated_tail_blocks(B * P) * n + sum_slots_scratch(n);
}

static int gated_tail_check(const char* what, int64_t B, int64_t C, int64_t P, const float* h, const float* x, const float* W,
                            int32_t in_act, int32_t post_act, const float* gamma, const float* beta) {
  if (B < 0 || P <= 0 || !gated_tail_c_ok(C)) { set_error("%s: unsupported sizes (C in {8, 16, 24, 32})", what); return -2; }
  if (B == 0) return 1;
  if (!h || !x || !W) { set_error("%s: null pointer", what); return -1; }
  if ((in_act != USF_ACT_NONE && in_act != USF_ACT_LEAKY_RELU) || (post_act != USF_ACT_NONE && post_act != USF_ACT_LEAKY_RELU)) {
    set_error("%s: bad act", what);
    return -2;
  }
  if ((gamma == nullptr) != (beta == nullptr) || (!gamma && post_act != USF_ACT_NONE)) {
    set_error("%s: the layer norm needs gamma and beta; post_act belongs to the layer norm", what);
    return -2;
  }
  if (B * P > 0x7fffffffLL * 8) { set_error("%s: too many pixels", what); return -3; }
  return 0;
}

#define USF_GT_DISPATCH(LN_, BWD_)                                                                                         \
  switch ((int)C) {                                                                                                        \
    case 8: hipLaunchKernelGGL((gated_tail_kernel<8, LN_, BWD_>), g, bl, 0, stream, a); break;                             \
    case 16: hipLaunchKernelGGL((gated_tail_kernel<16, LN_, BWD_>), g, bl, 0, stream, a); break;                           \
    case 24: hipLaunchKernelGGL((gated_tail_kernel<24, LN_, BWD_>), g, bl, 0, stream, a); break;                           \
    default: hipLaunchKernelGGL((gated_tail_kernel<32, LN_, BWD_>), g, bl, 0, stream, a); break;                           \
  }

int gated_tail_fwd(const float* h, const float* x, float* y, int64_t B, int64_t C, int64_t P, const float* W, const float* bias,
                   int32_t in_act, float in_slope, int32_t post_act, float post_slope, const float* gamma, const float* beta,
                   float eps, hipStream_t stream) {
  const int rc = gated_tail_check("usf_gated_tail_f32", B, C, P, h, x, W, in_act, post_act, gamma, beta);
  if (rc) return rc < 0 ? rc : 0;
  if (!y) { set_error("usf_gated_tail_f32: null pointer"); return -1; }
  if (y == h || y == x) { set_error("usf_gated_tail_f32: in-place operation is not supported"); return -2; }
  GtArgs a{};
  a.h = h; a.x = x; a.W = W; a.bias = bias; a.gamma = gamma; a.beta = beta; a.y = y; a.BP = B * P; a.P = P; a.eps = eps;
  a.in_slope = in_act == USF_ACT_LEAKY_RELU ? in_slope : 1.f;
  a.post_slope = post_act == USF_ACT_LEAKY_RELU ? post_slope : 1.f;
  const dim3 g((unsigned)gated_tail_blocks(a.BP)), bl(256);
  if (gamma) { USF_GT_DISPATCH(true, false) } else { USF_GT_DISPATCH(false, false) }
  return check_launch("usf_gated_tail_f32");
}

int gated_tail_bwd(const float* h, const float* x, const float* dy, float* dx, float* dh, float* dvg, int64_t B, int64_t C, int64_t P,
                   const float* W, const float* bias, int32_t in_act, float in_slope, int32_t post_act, float post_slope,
                   const float* gamma, const float* beta, float eps, float* dparams, float* workspace, int64_t workspace_floats,
                   usf_psum_job* job, hipStream_t stream) {
  const int rc = gated_tail_check("usf_gated_tail_bwd_f32", B, C, P, h, x, W, in_act, post_act, gamma, beta);
  if (job) job[0].nparts = job[1].nparts = 0;
  if (rc) return rc < 0 ? rc : 0;
  if (!dy || !dx || !dh) { set_error("usf_gated_tail_bwd_f32: null pointer"); return -1; }
  if (!dparams || !workspace || workspace_floats < gated_tail_workspace(B, C, P)) {
    set_error("usf_gated_tail_bwd_f32: dparams [2 C C + 2 C (+ 2 C)] and a workspace of usf_gated_tail_workspace floats are needed");
    return -2;
  }
  GtArgs a{};
  a.h = h; a.x = x; a.W = W; a.bias = bias; a.gamma = gamma; a.beta = beta; a.dy = dy; a.dx = dx; a.dh = dh; a.dvg = dvg;
  a.part = workspace; a.BP = B * P; a.P = P; a.eps = eps;
  a.in_slope = in_act == USF_ACT_LEAKY_RELU ? in_slope : 1.f;
  a.post_slope = post_act == USF_ACT_LEAKY_RELU ? post_slope : 1.f;
  const int n = (int)gated_tail_nparams(C, gamma != nullptr);
  a.slot_floats = n;
  const int blocks = gated_tail_blocks(a.BP);
  const dim3 g((unsigned)blocks), bl(256);
  if (gamma) { USF_GT_DISPATCH(true, true) } else { USF_GT_DISPATCH(false, true) }
  const int rc2 = check_launch("usf_gated_tail_bwd_f32");
  if (rc2) return rc2;
  return sum_slots(workspace, blocks, n, workspace + (int64_t)blocks * n, dparams, job, stream, "usf_gated_tail_bwd_f32 (sums)");
}
#undef USF_GT_DISPATCH

}  // namespace usf
