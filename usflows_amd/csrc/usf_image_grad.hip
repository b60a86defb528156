// Gradients of the image-shaped coupling layer's pieces (SURVEY rows N2 x N4: what torch autograd computes for
// networks.py:40-122, 405-510 and the 1 x 1 convolution of transforms.py:904-962 when Flow.fit trains an image flow).
//
//   conv_wgrad         dW[co, ci, tap] = sum_{b, p} dy[b, co, p] * xin[b, ci, p + tap],  db[co] = sum_{b, p} dy[b, co, p]
//                      for a stride-1 "same" convolution with kernel 1 or 3 (xin = in_act(x - pre_sub) * in_mul: the
//                      staging transforms of usf_conv2d_same_f32 / usf_channel_affine_f32), exact fp32 on
//                      v_mfma_f32_16x16x4_f32.  The data gradient of the same layers is the forward kernel on the flipped,
//                      transposed weight -- no kernel of its own.
//   layernorm_channels_bwd   dx, dgamma, dbeta of usf_layernorm_channels_f32 (with its folded (Leaky)ReLU)
//   gated_residual_bwd       d(vg) of usf_gated_residual_f32 (dx is dy itself)
//
// Reductions over the batch are deterministic: every wave writes its partial sums to a workspace slot of its own and one
// more launch adds the slots in a fixed order.
#include "usf_common.h"

namespace usf {

// ------------------------------------------------------------------------------------------
// Weight gradient.  One wave owns one sample at a time and ALL (cout tile, cin tile, tap) accumulator tiles, so a block
// needs no barrier: the wave stages its sample's x and dy into a private LDS image [channel][padded position] (fp32; row
// stride S = W + 1 and one zero row above / below for kernel 3, so a tap is a constant offset and the zero padding is
// real zeros), then walks the positions four at a time: A = dy[16 co][4 q], B = x[16 ci][4 (q + tap)] -> D[co][ci].
// The channel stride CS is 2 * odd: the 32 lanes of a ds_read_b32 group (16 channels x 2 positions) hit 32 banks.
// The next sample's 16-byte global loads are in flight during the matrix work (registers), 1 wave per SIMD on the
// 512-register budget.  MFMA-issue bound: 2 * B * HW * cin * cout * taps flops at the exact-fp32 matrix rate
// (157 TFLOP/s), padded positions included: (H - 1) * S + W of H * W (MNIST 7 x 7: 56 / 49).
// ------------------------------------------------------------------------------------------
struct WgArgs {
  const float* x;
  const float* dy;
  float* part;             // [waves][nacc]
  const float* in_mul;     // [cin * HW] or NULL
  const float* pre_sub;    // [cin] or NULL
  int B, cin, cout, HW, W, S, base, CS, nch, q0, nvx, nvy, nacc;
  unsigned m_hw, m_w;      // ceil(2^32 / HW), ceil(2^32 / W)
  int in_act;
  float in_slope;
  int toff[9];
};

template <int CIT, int COT, int T>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgArgs a) {
  extern __shared__ __attribute__((aligned(16))) float wg_lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int gw = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
  const int wsz = (a.cin + a.cout) * a.CS;
  float* xl = wg_lds + wave * wsz;
  float* dl = xl + a.cin * a.CS;
  for (int i = lane; i < wsz; i += 64) xl[i] = 0.f;           // padding positions stay zero for the whole launch

  f32x4 acc[COT][CIT][T];
#pragma unroll
  for (int i = 0; i < COT; ++i)
#pragma unroll
    for (int j = 0; j < CIT; ++j)
#pragma unroll
      for (int t = 0; t < T; ++t) acc[i][j][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bs[COT];
#pragma unroll
  for (int i = 0; i < COT; ++i) bs[i] = 0.f;

  constexpr int VX = CIT * 4, VY = COT * 4;                   // 16-byte vectors per lane (HW <= 64)
  constexpr int UNR = (COT * CIT * T > 36) ? 1 : 2;           // position chunks in flight (register budget)
  f32x4 px[VX], py[VY];
  const f32x4* x4 = reinterpret_cast<const f32x4*>(a.x);
  const f32x4* y4 = reinterpret_cast<const f32x4*>(a.dy);
  const f32x4* m4 = reinterpret_cast<const f32x4*>(a.in_mul);

  auto gload = [&](int s) {
    const f32x4* xs = x4 + (int64_t)s * a.nvx;
    const f32x4* ys = y4 + (int64_t)s * a.nvy;
#pragma unroll
    for (int j = 0; j < VX; ++j) {
      const int v = lane + 64 * j;
      if (v < a.nvx) px[j] = xs[v];
    }
#pragma unroll
    for (int j = 0; j < VY; ++j) {
      const int v = lane + 64 * j;
      if (v < a.nvy) py[j] = ys[v];
    }
  };
  // element e of a sample -> (channel, LDS index)
  auto lidx = [&](int e, int& c) {
    c = a.m_hw ? (int)__umulhi((unsigned)e, a.m_hw) : e;       // (magic 0: divisor 1)
    const int p = e - c * a.HW;
    const int r = a.m_w ? (int)__umulhi((unsigned)p, a.m_w) : p;
    return c * a.CS + a.base + p + r * (a.S - a.W);
  };
  auto stage = [&]() {
#pragma unroll
    for (int j = 0; j < VX; ++j) {
      const int v = lane + 64 * j;
      if (v < a.nvx) {
        f32x4 m = f32x4{1.f, 1.f, 1.f, 1.f};
        if (m4) m = m4[v];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          int c;
          const int idx = lidx(4 * v + i, c);
          float val = px[j][i];
          if (a.pre_sub) val -= a.pre_sub[c];
          val = act_apply(val, a.in_act, a.in_slope);
          xl[idx] = val * m[i];
        }
      }
    }
#pragma unroll
    for (int j = 0; j < VY; ++j) {
      const int v = lane + 64 * j;
      if (v < a.nvy) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          int c;
          const int idx = lidx(4 * v + i, c);
          dl[idx] = py[j][i];
        }
      }
    }
  };

  int s = gw;
  if (s < a.B) gload(s);
  const int loff = (lane & 15) * a.CS + a.q0 + (lane >> 4);
  for (; s < a.B; s += nw) {
    stage();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (s + nw < a.B) gload(s + nw);
#pragma unroll UNR
    for (int ch = 0; ch < a.nch; ++ch) {
      const int o = loff + 4 * ch;
      float av[COT], bv[CIT][T];
#pragma unroll
      for (int i = 0; i < COT; ++i) av[i] = dl[i * 16 * a.CS + o];
#pragma unroll
      for (int j = 0; j < CIT; ++j)
#pragma unroll
        for (int t = 0; t < T; ++t) bv[j][t] = xl[j * 16 * a.CS + o + a.toff[t]];
#pragma unroll
      for (int i = 0; i < COT; ++i) {
        bs[i] += av[i];
#pragma unroll
        for (int j = 0; j < CIT; ++j)
#pragma unroll
          for (int t = 0; t < T; ++t) acc[i][j][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j][t], acc[i][j][t], 0, 0, 0);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }

  // this wave's partial sums: tiles in register order (coalesced), the bias sums behind them
  float* pw = a.part + (int64_t)gw * a.nacc;
#pragma unroll
  for (int i = 0; i < COT; ++i)
#pragma unroll
    for (int j = 0; j < CIT; ++j)
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const int tile = (i * CIT + j) * T + t;
#pragma unroll
        for (int r = 0; r < 4; ++r) pw[tile * 256 + r * 64 + lane] = acc[i][j][t][r];
      }
#pragma unroll
  for (int i = 0; i < COT; ++i) {
    float v = bs[i];
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    if (lane < 16) pw[COT * CIT * T * 256 + i * 16 + lane] = v;
  }
}

// out[j] = sum over the partial slots in a fixed order (four interleaved chains, then ((0 + 1) + 2) + 3).
// mode 0: out[j] for j < n.  mode 1 (weight gradient): slot layout of conv_wgrad_kernel -> dW [cout][cin][T], db [cout].
__global__ __launch_bounds__(256) void partial_sum_kernel(const float* __restrict__ part, int nparts, int n, float* __restrict__ out,
                                                          float* __restrict__ out2, int mode, int cin, int cout, int CIT, int T,
                                                          int ntile) {
  __shared__ float red[4][64];
  const int jj = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + jj;
  float s = 0.f;
  if (j < n)
    for (int p = g; p < nparts; p += 4) s += part[(int64_t)p * n + j];
  red[g][jj] = s;
  __syncthreads();
  if (g != 0 || j >= n) return;
  const float total = ((red[0][jj] + red[1][jj]) + red[2][jj]) + red[3][jj];
  if (mode == 0) {
    out[j] = total;
    return;
  }
  if (j >= ntile * 256) {
    const int co = j - ntile * 256;
    if (out2 && co < cout) out2[co] = total;
    return;
  }
  const int tile = j >> 8, r = (j >> 6) & 3, ln = j & 63;
  const int cot = tile / (CIT * T), cit = (tile / T) % CIT, t = tile % T;
  const int co = cot * 16 + 4 * (ln >> 4) + r, ci = cit * 16 + (ln & 15);
  if (co < cout && ci < cin) out[((int64_t)co * cin + ci) * T + t] = total;
}

// ceil(2^32 / d): floor(n / d) == umulhi(n, magic) for n * d < 2^32; 0 stands for d == 1
static unsigned magic_div(int d) { return d <= 1 ? 0u : (unsigned)((0x100000000ULL + (unsigned)d - 1) / (unsigned)d); }

// workspace floats needed by conv_wgrad for this shape / batch (0: shape not served)
int64_t conv_wgrad_workspace(int64_t B, int64_t cin, int64_t cout, int64_t H, int64_t W, int64_t ks);

namespace {
struct WgPlan {
  int CIT, COT, T, S, base, CS, nch, q0, nacc, lds_bytes, blocks;
};
bool wgrad_plan(int64_t B, int64_t cin, int64_t cout, int64_t H, int64_t W, int64_t ks, WgPlan& pl) {
  if (B <= 0 || cin <= 0 || cout <= 0 || (cin & 15) || (cout & 15) || H <= 0 || W <= 0 || H * W > 64 || (ks != 1 && ks != 3))
    return false;
  pl.CIT = (int)(cin / 16);
  pl.COT = (int)(cout / 16);
  pl.T = (int)(ks * ks);
  if (pl.CIT > 4 || pl.COT > 4) return false;
  if (pl.T == 9 && pl.CIT * pl.COT > 6) return false;           // accumulator tiles on the register budget
  int need;
  if (ks == 3) {
    pl.S = (int)W + 1;
    pl.base = pl.S + 1;
    pl.q0 = pl.S + 1;
    const int nk = (int)((H - 1) * pl.S + W);                    // padded positions from the first to the last pixel
    pl.nch = (nk + 3) / 4;
    need = pl.q0 + 4 * pl.nch + pl.S + 1;                         // last position read + 1
  } else {
    pl.S = (int)W;
    pl.base = 0;
    pl.q0 = 0;
    pl.nch = (int)((H * W + 3) / 4);
    need = 4 * pl.nch;
  }
  const int full = (ks == 3) ? (int)((H + 2) * pl.S + 1) : (int)(H * W);
  if (need < full) need = full;
  int cs = (need + 1) / 2;                                        // CS = 2 * odd >= need
  if (!(cs & 1)) ++cs;
  pl.CS = 2 * cs;
  pl.nacc = pl.COT * pl.CIT * pl.T * 256 + pl.COT * 16;
  pl.lds_bytes = 4 * (int)(cin + cout) * pl.CS * 4;
  if (pl.lds_bytes > 160 * 1024) return false;
  const int per_cu = (2 * pl.lds_bytes <= 160 * 1024 && pl.CIT * pl.COT * pl.T * 4 <= 96) ? 2 : 1;
  int64_t blocks = (B + 3) / 4;
  const int64_t cap = (int64_t)device_cu_count() * per_cu;
  if (blocks > cap) blocks = cap;
  pl.blocks = (int)blocks;
  return true;
}
}  // namespace

int64_t conv_wgrad_workspace(int64_t B, int64_t cin, int64_t cout, int64_t H, int64_t W, int64_t ks) {
  WgPlan pl;
  if (!wgrad_plan(B, cin, cout, H, W, ks, pl)) return 0;
  return (int64_t)pl.blocks * 4 * pl.nacc;
}

int conv_wgrad(const float* x, const float* dy, int64_t B, int64_t cin, int64_t cout, int64_t H, int64_t W, int64_t ks,
               const float* in_mul, const float* pre_sub, int32_t in_act, float in_slope, float* dW, float* db, float* workspace,
               int64_t workspace_floats, hipStream_t stream) {
  WgPlan pl;
  if (B < 0) { set_error("usf_conv_wgrad_f32: bad sizes"); return -2; }
  if (B == 0 || !wgrad_plan(B, cin, cout, H, W, ks, pl)) {
    if (B == 0) { set_error("usf_conv_wgrad_f32: empty batch"); return -2; }
    return 1;                                                     // shape not served
  }
  if (!x || !dy || !dW || !workspace) { set_error("usf_conv_wgrad_f32: null pointer"); return -1; }
  if (!aligned16(x) || !aligned16(dy) || (in_mul && !aligned16(in_mul))) { set_error("usf_conv_wgrad_f32: tensors must be 16-byte aligned"); return -2; }
  if (in_act != USF_ACT_NONE && in_act != USF_ACT_LEAKY_RELU) { set_error("usf_conv_wgrad_f32: bad act"); return -2; }
  if (workspace_floats < (int64_t)pl.blocks * 4 * pl.nacc) { set_error("usf_conv_wgrad_f32: workspace too small"); return -2; }
  WgArgs a;
  a.x = x; a.dy = dy; a.part = workspace; a.in_mul = in_mul; a.pre_sub = pre_sub;
  a.B = (int)B; a.cin = (int)cin; a.cout = (int)cout; a.HW = (int)(H * W); a.W = (int)W; a.S = pl.S; a.base = pl.base;
  a.CS = pl.CS; a.nch = pl.nch; a.q0 = pl.q0; a.nvx = (int)(cin * H * W / 4); a.nvy = (int)(cout * H * W / 4); a.nacc = pl.nacc;
  a.m_hw = magic_div(a.HW); a.m_w = magic_div(a.W); a.in_act = in_act; a.in_slope = in_slope;
  for (int t = 0; t < 9; ++t) a.toff[t] = (ks == 3) ? ((t / 3) - 1) * pl.S + ((t % 3) - 1) : 0;
  const dim3 g((unsigned)pl.blocks), b(256);
  static bool attr_done[USF_MAX_DEVICES][5][5][2];
  const int dev = current_device_slot();
#define USF_WG(CIT_, COT_, T_)                                                                                             \
  if (pl.CIT == CIT_ && pl.COT == COT_ && pl.T == T_) {                                                                     \
    if (!attr_done[dev][CIT_][COT_][T_ == 9]) {                                                                            \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<CIT_, COT_, T_>),                           \
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {                     \
        set_error("usf_conv_wgrad_f32: cannot raise the dynamic LDS limit"); return -4; }                                  \
      attr_done[dev][CIT_][COT_][T_ == 9] = true;                                                                          \
    }                                                                                                                      \
    hipLaunchKernelGGL((conv_wgrad_kernel<CIT_, COT_, T_>), g, b, (size_t)pl.lds_bytes, stream, a);                        \
    launched = true;                                                                                                       \
  }
  bool launched = false;
  USF_WG(1, 1, 9) USF_WG(1, 2, 9) USF_WG(2, 1, 9) USF_WG(2, 2, 9) USF_WG(3, 2, 9) USF_WG(2, 3, 9) USF_WG(1, 3, 9) USF_WG(3, 1, 9)
  USF_WG(1, 1, 1) USF_WG(1, 2, 1) USF_WG(2, 1, 1) USF_WG(2, 2, 1) USF_WG(2, 4, 1) USF_WG(4, 2, 1) USF_WG(3, 3, 1) USF_WG(4, 4, 1)
  USF_WG(1, 4, 1) USF_WG(4, 1, 1) USF_WG(3, 2, 1) USF_WG(2, 3, 1) USF_WG(1, 3, 1) USF_WG(3, 1, 1) USF_WG(3, 4, 1) USF_WG(4, 3, 1)
#undef USF_WG
  if (!launched) return 1;
  int rc = check_launch("usf_conv_wgrad_f32");
  if (rc) return rc;
  const int ntile = pl.COT * pl.CIT * pl.T;
  hipLaunchKernelGGL(partial_sum_kernel, dim3((unsigned)((pl.nacc + 63) / 64)), dim3(256), 0, stream, workspace, pl.blocks * 4,
                     pl.nacc, dW, db, 1, (int)cin, (int)cout, pl.CIT, pl.T, ntile);
  return check_launch("usf_conv_wgrad_f32 (reduce)");
}

// ------------------------------------------------------------------------------------------
// LayerNormChannels backward (with the folded (Leaky)ReLU): per pixel over the channel axis
//   a = act(x), xh = (a - mean) / den, y = xh * gamma + beta
//   g = dy * gamma, da = (g - mean_c g - xh * mean_c (g * xh)) / den, dx = da * act'(x)
//   dgamma[c] = sum_{b, p} dy * xh, dbeta[c] = sum_{b, p} dy
// One thread per pixel, grid-stride; the parameter sums stay in registers and leave once per wave.
// HBM-bound: 12 bytes per element.
// ------------------------------------------------------------------------------------------
template <int CMAX>
__global__ __launch_bounds__(256) void layernorm_channels_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                     float* __restrict__ dx, int64_t BP, int C, int64_t P,
                                                                     const float* __restrict__ gamma, float eps, int act,
                                                                     float slope, float* __restrict__ part) {
  float dg[CMAX], dbt[CMAX];
#pragma unroll
  for (int c = 0; c < CMAX; ++c) dg[c] = dbt[c] = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < BP; i += (int64_t)gridDim.x * 256) {
    const int64_t b = i / P, p = i - b * P;
    const float* xb = x + b * C * P + p;
    const float* gb = dy + b * C * P + p;
    float v[CMAX], xr[CMAX];
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
      xr[c] = (c < C) ? xb[(int64_t)c * P] : 0.f;
      v[c] = (c < C) ? act_apply(xr[c], act, slope) : 0.f;
      sum += v[c];
    }
    const float mean = sum / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
      const float d = (c < C) ? v[c] - mean : 0.f;
      sq += d * d;
    }
    const float den = sqrtf(sq / (float)C + eps);
    float m1 = 0.f, m2 = 0.f;
    float g[CMAX];
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
      const float d = (c < C) ? gb[(int64_t)c * P] : 0.f;
      v[c] = (c < C) ? (v[c] - mean) / den : 0.f;            // xh
      g[c] = (c < C) ? d * gamma[c] : 0.f;
      m1 += g[c];
      m2 += g[c] * v[c];
      dg[c] += d * v[c];
      dbt[c] += d;
    }
    m1 /= (float)C;
    m2 /= (float)C;
    float* ob = dx + b * C * P + p;
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
      if (c < C) {
        float da = (g[c] - m1 - v[c] * m2) / den;
        if (act == USF_ACT_LEAKY_RELU) da = gate_apply(da, xr[c], slope);
        ob[(int64_t)c * P] = da;
      }
  }
  const int lane = threadIdx.x & 63;
  float* pw = part + ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * (2 * C);
#pragma unroll
  for (int c = 0; c < CMAX; ++c)
    if (c < C) {
      const float a = wave_sum(dg[c]), bsum = wave_sum(dbt[c]);
      if (lane == 0) { pw[c] = a; pw[C + c] = bsum; }
    }
}

int64_t layernorm_channels_bwd_workspace(int64_t B, int64_t C, int64_t P) {
  if (B <= 0 || C <= 0 || C > 64 || P <= 0) return 0;
  int64_t blocks = (B * P + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  return blocks * 4 * 2 * C;
}

int layernorm_channels_bwd(const float* x, const float* dy, float* dx, int64_t B, int64_t C, int64_t P, const float* gamma, float eps,
                           int32_t act, float slope, float* dgamma, float* dbeta, float* workspace, int64_t workspace_floats,
                           hipStream_t stream) {
  if (B <= 0 || C <= 0 || P <= 0 || C > 64) { set_error("usf_layernorm_channels_bwd_f32: bad sizes (C must be 1..64, B > 0)"); return -2; }
  if (!x || !dy || !dx || !gamma || !dgamma || !dbeta || !workspace) { set_error("usf_layernorm_channels_bwd_f32: null pointer"); return -1; }
  if (dgamma + C != dbeta) { set_error("usf_layernorm_channels_bwd_f32: dbeta must follow dgamma (one [2 C] buffer)"); return -2; }
  if (act != USF_ACT_NONE && act != USF_ACT_LEAKY_RELU) { set_error("usf_layernorm_channels_bwd_f32: bad act"); return -2; }
  const int64_t need = layernorm_channels_bwd_workspace(B, C, P);
  if (workspace_floats < need) { set_error("usf_layernorm_channels_bwd_f32: workspace too small"); return -2; }
  const int64_t BP = B * P;
  const int blocks = (int)(need / (4 * 2 * C));
  const dim3 g((unsigned)blocks), b(256);
  if (C <= 16) hipLaunchKernelGGL(layernorm_channels_bwd_kernel<16>, g, b, 0, stream, x, dy, dx, BP, (int)C, P, gamma, eps, act, slope, workspace);
  else if (C <= 32) hipLaunchKernelGGL(layernorm_channels_bwd_kernel<32>, g, b, 0, stream, x, dy, dx, BP, (int)C, P, gamma, eps, act, slope, workspace);
  else hipLaunchKernelGGL(layernorm_channels_bwd_kernel<64>, g, b, 0, stream, x, dy, dx, BP, (int)C, P, gamma, eps, act, slope, workspace);
  int rc = check_launch("usf_layernorm_channels_bwd_f32");
  if (rc) return rc;
  const int n = (int)(2 * C);
  hipLaunchKernelGGL(partial_sum_kernel, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, stream, workspace, blocks * 4, n, dgamma,
                     (float*)nullptr, 0, 0, 0, 0, 0, 0);
  return check_launch("usf_layernorm_channels_bwd_f32 (reduce)");
}

// ------------------------------------------------------------------------------------------
// Weight planes of usf_conv2d_same_f32 from an fp32 nn.Conv2d weight [cout, cin, ks, ks] in one launch (the training step
// needs them twice per convolution and step: as they are, and flipped / transposed for the data gradient):
//   planes[q][co][tap * cp + ci] = q-th bf16 term of W[co, ci, tap]           (transposed == 0)
//                                = q-th bf16 term of W[ci, co, ks*ks-1 - tap] (transposed != 0: rows = the forward's INPUT channels)
// round-to-nearest-even residual split h = bf16(w), m = bf16(w - h), l = bf16(w - h - m); zeros in all padding.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned short bf16_rne(float f) {
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);   // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_to_f32(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

__global__ __launch_bounds__(256) void conv_weight_planes_kernel(const float* __restrict__ w, unsigned short* __restrict__ planes, int rows,
                                                                 int cols, int ks2, int cp, int kp, int coutp, int transposed) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= coutp * kp) return;
  const int r = i / kp, k = i - r * kp;
  const int tap = k / cp, c = k - tap * cp;
  float v = 0.f;
  if (r < rows && tap < ks2 && c < cols)
    v = transposed ? w[((int64_t)c * rows + r) * ks2 + (ks2 - 1 - tap)] : w[((int64_t)r * cols + c) * ks2 + tap];
  const unsigned short h = bf16_rne(v);
  const float r1 = v - bf16_to_f32(h);
  const unsigned short m = bf16_rne(r1);
  const unsigned short l = bf16_rne(r1 - bf16_to_f32(m));
  planes[i] = h;
  planes[(int64_t)coutp * kp + i] = m;
  planes[2 * (int64_t)coutp * kp + i] = l;
}

int conv2d_weight_planes(const float* w, void* planes, int64_t cin, int64_t cout, int64_t ks, int32_t transposed, hipStream_t stream) {
  if (cin <= 0 || cout <= 0 || cin > 4096 || cout > 4096 || (ks != 1 && ks != 3)) { set_error("usf_conv2d_weight_planes_f32: bad sizes"); return -2; }
  if (!w || !planes) { set_error("usf_conv2d_weight_planes_f32: null pointer"); return -1; }
  // rows / cols of the packed matrix: the convolution the planes are FOR maps `cols` channels to `rows` channels
  const int rows = (int)(transposed ? cin : cout), cols = (int)(transposed ? cout : cin);
  const int cp = (cols + 7) / 8 * 8, kp = (int)((ks * ks * cp + 31) / 32 * 32), coutp = (rows + 15) / 16 * 16;
  const int n = coutp * kp;
  hipLaunchKernelGGL(conv_weight_planes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, w,
                     reinterpret_cast<unsigned short*>(planes), rows, cols, (int)(ks * ks), cp, kp, coutp, (int)transposed);
  return check_launch("usf_conv2d_weight_planes_f32");
}

// d(vg) of y = x + val * sigmoid(gate): d val = dy * s, d gate = dy * val * s * (1 - s); dx = dy (no kernel)
__global__ __launch_bounds__(256) void gated_residual_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ vg,
                                                                 float* __restrict__ dvg, int64_t total, int64_t CP) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t b = e / CP, r = e - b * CP;
    const float val = vg[b * 2 * CP + r], gate = vg[b * 2 * CP + CP + r];
    const float s = 1.f / (1.f + expf(-gate));
    const float d = dy[e];
    dvg[b * 2 * CP + r] = d * s;
    dvg[b * 2 * CP + CP + r] = d * val * (s * (1.f - s));
  }
}

int gated_residual_bwd(const float* dy, const float* vg, float* dvg, int64_t B, int64_t CP, hipStream_t stream) {
  if (B < 0 || CP <= 0) { set_error("usf_gated_residual_bwd_f32: bad sizes"); return -2; }
  if (B == 0) return 0;
  if (!dy || !vg || !dvg) { set_error("usf_gated_residual_bwd_f32: null pointer"); return -1; }
  int64_t blocks = (B * CP + 255) / 256;
  if (blocks > 256 * 64) blocks = 256 * 64;
  hipLaunchKernelGGL(gated_residual_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, dy, vg, dvg, B * CP, CP);
  return check_launch("usf_gated_residual_bwd_f32");
}

}  // namespace usf
