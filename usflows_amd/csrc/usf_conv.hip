// Convolution conditioner of the image-shaped flows: Conv2d with stride 1, "same" zero padding,
// kernel 1 x 1 or 3 x 3 (networks.py:405-510 ConvNet2D, :61-122 GatedConv) as an implicit GEMM on the bf16 matrix cores
// with the fp32-equivalent bf16x3 arithmetic of the rest of the path (three bf16 planes per operand, six
// v_mfma_f32_16x16x32_bf16 per product, fp32 accumulation).
//
//   y[b, co, p] = out_act( bias[co] + sum_{tap, ci} W[co, ci, tap] * a[b, ci, p + tap] ),   a = in_act(x) * in_mul
//
// in_act: the (Leaky)ReLU GatedConv / ConvNet2D put in front of a convolution; in_mul [Cin * H * W]: the coupling mask
// of MaskedCoupling (x * mask feeds the first convolution); both fold into the staging pass.
//
// Data flow of a 512-thread block (persistent over groups of S samples):
//   * weights: pre-split planes [3][CoutP][KP] bf16 (K order: tap-major, channel-minor, channels padded to Cp = ceil8(Cin),
//     KP = ceil32(taps * Cp)) copied into LDS once per block;
//   * inputs: the S samples of a group are read from HBM once (coalesced fp32), activated / masked, split into three bf16
//     planes and written into a zero-bordered LDS image [S x (H+2)(W+2) positions][Cp channels] -- channel-minor, so a
//     lane's MFMA operand (8 consecutive k = 8 channels of one tap) is one 16-byte LDS read at position (p + tap);
//   * GEMM: M = output channels (A operand = weights), N = rows (sample, pixel) of the group (B operand = input image),
//     K = taps * Cp; a wave owns 32 x 32 output patches (2 x 2 MFMA tiles: every fragment read feeds 12 MFMAs);
//   * epilogue: bias, activation, 16 consecutive pixels per store instruction.
// Row strides of both LDS images are an odd number of 16-byte units: the 16 lanes of a fragment read hit 16 different
// bank groups.
#include "usf_common.h"
#include <type_traits>

namespace usf {

typedef __bf16 cv_bf16x8 __attribute__((ext_vector_type(8)));

struct ConvArgs {
  const float* x; float* y;
  const __bf16* wp;            // [3][coutp][kp]
  const float* bias;           // [cout] or null
  const float* in_mul;         // [cin * hw] or null
  int B, cin, cout, H, W, ks;
  int cp, kp, coutp;           // padded channel count, padded K, padded Cout (multiple of 16)
  int S;                       // samples per group
  int xs16, ws16;              // LDS row strides in 16-byte units (odd)
  int in_act, out_act; float in_slope, out_slope;
  const float* gate_x; int gateC;   // gated mode (GatedConv's second convolution + its gate): see usf_conv2d_same_f32
  int dbg;                     // tuning aid (USF_CONV_DBG): 1 no staging, 2 no k loop, 4 no output stores, 8 no input loads
};

__device__ __forceinline__ void cv_split(float x, __bf16& h, __bf16& m, __bf16& l) {
  h = (__bf16)x;
  const float r = x - (float)h;
  m = (__bf16)r;
  l = (__bf16)(r - (float)m);
}

__global__ __launch_bounds__(512, 2) void conv2d_same_bf16x3_kernel(const ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int HW = a.H * a.W;
  const int pad = a.ks >> 1;
  const int PW = a.W + 2 * pad, PP = (a.H + 2 * pad) * PW;         // padded image
  const int taps = a.ks * a.ks;
  const int wrow = a.ws16 * 16, xrow = a.xs16 * 16;                // bytes per LDS row
  const int wplane = a.coutp * wrow, xplane = a.S * PP * xrow;      // bytes per plane
  unsigned char* const Wl = smem;
  unsigned char* const Xl = smem + 3 * wplane;

  // ---- once per block: weights into LDS, the input image zeroed (borders and padding channels stay zero) ----
  {
    const int units = a.kp / 8;                                      // 16-byte units per weight row
    for (int i = tid; i < 3 * a.coutp * units; i += 512) {
      const int pl = i / (a.coutp * units), rem = i % (a.coutp * units);
      const int r = rem / units, u = rem % units;
      *reinterpret_cast<cv_bf16x8*>(Wl + pl * wplane + r * wrow + u * 16) =
          *reinterpret_cast<const cv_bf16x8*>(a.wp + ((size_t)(pl * a.coutp + r) * a.kp + 8 * u));
    }
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (int i = tid; i < 3 * xplane / 16; i += 512) *reinterpret_cast<f32x4*>(Xl + i * 16) = z;
  }
  __syncthreads();

  const int ngroups = (a.B + a.S - 1) / a.S;
  const int nblk = a.kp / 32;
  const int cpairs = (a.cin + 1) >> 1;
  const int ppass = (HW + 63) >> 6;                                  // 64-pixel passes over a channel plane
  // Staging: a wave takes (sample, channel pair) planes, a lane one pixel of both channels -- two coalesced loads, two
  // splits, three 4-byte LDS stores (the pair is adjacent in the channel-minor image).  The loads of group g + 1 are
  // issued before the MFMA phase of group g and consumed after it: HBM latency hides under the matrix work.
  constexpr int MAXIT = 16;                                          // (host: S is chosen so that a group needs <= MAXIT)
  float pre[MAXIT][2];
  // per 64-pixel pass q: this lane's pixel and the LDS offset of its (padded) position -- no divisions in the loops below
  int poff[4], pix[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int p = min(lane + 64 * q, HW - 1);
    const int py = p / a.W, px = p - py * a.W;
    pix[q] = (lane + 64 * q < HW) ? p : -1;
    poff[q] = ((py + pad) * PW + (px + pad)) * xrow;
  }
  auto fetch = [&](int gidx) {
    const int s0 = gidx * a.S;
    const int npl = min(a.S, a.B - s0) * cpairs;                      // (sample, channel pair) planes of the group
    const float* xg = a.x + (size_t)s0 * a.cin * HW;
    int j = 0, q = 0;                                                 // plane round / pixel pass of iteration `it` (wave-uniform)
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      const int pp = wave + 8 * j;
      const int p = q == 0 ? pix[0] : q == 1 ? pix[1] : q == 2 ? pix[2] : pix[3];
      pre[it][0] = pre[it][1] = 0.f;
      if (pp < npl && p >= 0) {
        const int sl = pp / cpairs, c = 2 * (pp - sl * cpairs);
        const float* x0 = xg + ((size_t)sl * a.cin + c) * HW + p;
        pre[it][0] = x0[0];
        if (c + 1 < a.cin) pre[it][1] = x0[HW];
      }
      if (++q == ppass) { q = 0; ++j; }
    }
  };
  auto stage = [&](int gidx) {
    const int s0 = gidx * a.S;
    const int npl = min(a.S, a.B - s0) * cpairs;
    int j = 0, q = 0;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      const int pp = wave + 8 * j;
      const int p = q == 0 ? pix[0] : q == 1 ? pix[1] : q == 2 ? pix[2] : pix[3];
      const int po = q == 0 ? poff[0] : q == 1 ? poff[1] : q == 2 ? poff[2] : poff[3];
      if (pp < npl && p >= 0) {
        const int sl = pp / cpairs, c = 2 * (pp - sl * cpairs);
        float v0 = act_apply(pre[it][0], a.in_act, a.in_slope), v1 = act_apply(pre[it][1], a.in_act, a.in_slope);
        if (a.in_mul) {
          v0 *= a.in_mul[c * HW + p];
          if (c + 1 < a.cin) v1 *= a.in_mul[(c + 1) * HW + p];
        }
        __bf16 h0, m0, l0, h1, m1, l1;
        cv_split(v0, h0, m0, l0);
        cv_split(v1, h1, m1, l1);
        unsigned char* dst = Xl + sl * PP * xrow + po + 2 * c;
        typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
        *reinterpret_cast<bf2*>(dst) = (bf2){h0, h1};
        *reinterpret_cast<bf2*>(dst + xplane) = (bf2){m0, m1};
        *reinterpret_cast<bf2*>(dst + 2 * xplane) = (bf2){l0, l1};
      }
      if (++q == ppass) { q = 0; ++j; }
    }
  };
  if ((int)blockIdx.x < ngroups) fetch(blockIdx.x);
  for (int gidx = blockIdx.x; gidx < ngroups; gidx += gridDim.x) {
    const int s0 = gidx * a.S;
    const int ns = min(a.S, a.B - s0);
    const int R = ns * HW;                                           // live rows of this group
    if (!(a.dbg & 1)) stage(gidx);
    __syncthreads();
    if (gidx + (int)gridDim.x < ngroups && !(a.dbg & 8)) fetch(gidx + gridDim.x);
    // ---- implicit GEMM: 32 (co) x 32 (rows) patches dealt over the waves ----
    // Patch shape by the amount of work in the group: 32 (co) x 32 (rows) when that gives every wave a patch, otherwise
    // 16 x 32 or 16 x 16 (more patches, fewer MFMAs per fragment read)
    auto patches = [&](auto pc_, auto pr_) {
      constexpr int PC = decltype(pc_)::value, PR = decltype(pr_)::value;      // 16-row tiles per patch in co / rows
      const int ctn = (a.coutp / 16 + PC - 1) / PC, rtn = (R + 16 * PR - 1) / (16 * PR);
      for (int t = wave; t < ctn * rtn; t += 8) {
        const int ct = t / rtn, rt = t - ct * rtn;
        // this lane's row positions (B operand columns): row r -> (sample, pixel) -> top-left of its window
        int xbase[PR];
#pragma unroll
        for (int b = 0; b < PR; ++b) {
          const int r = min(rt * 16 * PR + b * 16 + li, R - 1);
          const int sl = r / HW, p = r - sl * HW;
          const int py = p / a.W, px = p - py * a.W;
          xbase[b] = (sl * PP + py * PW + px) * xrow;
        }
        const int co0 = ct * 16 * PC;
        const bool two_co = PC == 2 && co0 + 16 < a.coutp;              // wave-uniform
        int wbase[PC];
        wbase[0] = (co0 + li) * wrow;
        if (PC == 2) wbase[PC - 1] = (min(co0 + 16, a.coutp - 16) + li) * wrow;
        f32x4 acc[PC][PR];
#pragma unroll
        for (int i = 0; i < PC; ++i)
#pragma unroll
          for (int b = 0; b < PR; ++b) acc[i][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
        cv_bf16x8 wf[2][PC][3], xf[2][PR][3];                            // [buffer][tile][plane]: block k + 1 is read under block k
        auto read_blk = [&](int blk, cv_bf16x8 (&w)[PC][3], cv_bf16x8 (&xv)[PR][3]) {
          const int kflat = blk * 32 + 8 * lg;
          int tap = kflat / a.cp;
          const int c = kflat - tap * a.cp;
          tap = min(tap, taps - 1);                                      // K padding: weights are zero there
          const int dy = tap / a.ks, dx = tap - dy * a.ks;
          const int xoff = (dy * PW + dx) * xrow + 2 * c;
          const int woff = 2 * kflat;
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
            for (int i = 0; i < PC; ++i) w[i][pl] = *reinterpret_cast<const cv_bf16x8*>(Wl + pl * wplane + wbase[i] + woff);
#pragma unroll
            for (int b = 0; b < PR; ++b) xv[b][pl] = *reinterpret_cast<const cv_bf16x8*>(Xl + pl * xplane + xbase[b] + xoff);
          }
        };
        auto mm_blk = [&](const cv_bf16x8 (&w)[PC][3], const cv_bf16x8 (&xv)[PR][3]) {
#define USF_CV(I, B_, P, Q) acc[I][B_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[I][P], xv[B_][Q], acc[I][B_], 0, 0, 0)
#pragma unroll
          for (int b = 0; b < PR; ++b) {
            USF_CV(0, b, 2, 0); USF_CV(0, b, 1, 1); USF_CV(0, b, 0, 2); USF_CV(0, b, 1, 0); USF_CV(0, b, 0, 1); USF_CV(0, b, 0, 0);
            if (PC == 2 && two_co) {
              USF_CV(PC - 1, b, 2, 0); USF_CV(PC - 1, b, 1, 1); USF_CV(PC - 1, b, 0, 2); USF_CV(PC - 1, b, 1, 0); USF_CV(PC - 1, b, 0, 1);
              USF_CV(PC - 1, b, 0, 0);
            }
          }
#undef USF_CV
        };
        read_blk(0, wf[0], xf[0]);
        int blk = (a.dbg & 2) ? nblk : 0;
        for (; blk + 2 <= nblk; blk += 2) {
          read_blk(blk + 1, wf[1], xf[1]);
          mm_blk(wf[0], xf[0]);
          read_blk(min(blk + 2, nblk - 1), wf[0], xf[0]);
          mm_blk(wf[1], xf[1]);
        }
        if (blk < nblk) mm_blk(wf[0], xf[0]);
        // ---- epilogue: lane (col = li, g = lg) of tile (i, b) holds output channels co0 + 16 i + 4 g + (0..3) of row b*16 + li
#pragma unroll
        for (int b = 0; b < PR; ++b) {
          const int r = rt * 16 * PR + b * 16 + li;
          if (r >= R) continue;
          const int sl = r / HW, p = r - sl * HW;
          float* yb = a.y + ((size_t)(s0 + sl) * a.cout) * HW + p;
          if (PC == 2 && a.gateC > 0) {
            // gated mode: the patch's two tiles are (value, gate) of the same 16 channels (rows interleaved at pack time):
            // y[c] = x[c] + value * sigmoid(gate) -- GatedConv.forward's tail (networks.py:108-122) without the [B, 2C] tensor
            const size_t gb = ((size_t)(s0 + sl) * a.gateC) * HW + p;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int c = (co0 >> 1) + 4 * lg + j;
              if (c < a.gateC && !(a.dbg & 4)) {
                const float val = acc[0][b][j] + (a.bias ? a.bias[co0 + 4 * lg + j] : 0.f);
                const float gt = acc[PC - 1][b][j] + (a.bias ? a.bias[co0 + 16 + 4 * lg + j] : 0.f);
                a.y[gb + (size_t)c * HW] = a.gate_x[gb + (size_t)c * HW] + val * (1.f / (1.f + expf(-gt)));
              }
            }
            continue;
          }
#pragma unroll
          for (int i = 0; i < PC; ++i) {
            if (i == 1 && !two_co) break;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int co = co0 + 16 * i + 4 * lg + j;
              if (co < a.cout && !(a.dbg & 4)) {
                float v = acc[i][b][j] + (a.bias ? a.bias[co] : 0.f);
                yb[(size_t)co * HW] = act_apply(v, a.out_act, a.out_slope);
              }
            }
          }
        }
      }
    };
    typedef std::integral_constant<int, 1> I1;
    typedef std::integral_constant<int, 2> I2;
    const int c16 = a.coutp / 16, r32 = (R + 31) / 32;
    if (((c16 + 1) / 2) * r32 >= 7 || (a.gateC > 0 && (c16 / 2) * r32 >= 4)) patches(I2(), I2());
    else if (a.gateC > 0) patches(I2(), I1());                       // (value, gate) tile pairs stay together
    else if (c16 * r32 >= 7) patches(I1(), I2());
    else patches(I1(), I1());
    __syncthreads();                                                 // the image is rewritten by the next group
  }
}

// second kernel (usf_conv_wreg.hip): 1 = launched, 0 = shape not served there, < 0 = error
int conv2d_same_wreg(const float* x, float* y, int64_t B, int64_t cin, int64_t cout, int64_t H, int64_t W, const void* wplanes,
                     const float* bias, const float* in_mul, int32_t in_act, float in_slope, int32_t out_act, float out_slope,
                     const float* res_x, const float* res_mul, float res_sign, int res_mode, hipStream_t stream);

static int odd16(int units) { return units | 1; }
// register-staged iterations per thread for a group of S samples (the kernel holds at most 16 pixel pairs per thread)
static int conv_stage_iters(int cin, int HW, int S) { return ((S * ((cin + 1) / 2) + 7) / 8) * ((HW + 63) / 64); }

// LDS bytes of a launch with S samples per group; the layout the kernel derives from the same numbers
static int64_t conv_lds_bytes(int cin, int cout, int H, int W, int ks, int S, int* xs16, int* ws16, int* cp_, int* kp_, int* coutp_) {
  const int cp = (cin + 7) / 8 * 8, taps = ks * ks, kp = (taps * cp + 31) / 32 * 32, coutp = (cout + 15) / 16 * 16;
  const int pad = ks / 2, PP = (H + 2 * pad) * (W + 2 * pad);
  *xs16 = odd16(cp / 8); *ws16 = odd16(kp / 8); *cp_ = cp; *kp_ = kp; *coutp_ = coutp;
  return 3LL * coutp * (*ws16) * 16 + 3LL * S * PP * (*xs16) * 16;
}

// elements ([3][coutp][kp] bf16) of the weight planes usf_conv2d_same_f32 expects for these sizes
int64_t conv2d_weight_elems(int64_t cin, int64_t cout, int64_t ks) {
  if (cin <= 0 || cout <= 0 || (ks != 1 && ks != 3)) return -1;
  const int64_t cp = (cin + 7) / 8 * 8, kp = (ks * ks * cp + 31) / 32 * 32, coutp = (cout + 15) / 16 * 16;
  return 3 * coutp * kp;
}

// samples per group the kernel would use for these sizes (the LDS decides); 0: the shape is not served
int conv2d_same_fits(int64_t cin, int64_t cout, int64_t H, int64_t W, int64_t ks) {
  if (cin <= 0 || cout <= 0 || H <= 0 || W <= 0 || cin > 64 || cout > 64 || H * W > 256 || (ks != 1 && ks != 3)) return 0;
  int xs, ws, cp, kp, coutp;
  for (int S = 8; S >= 1; --S)
    if (conv_lds_bytes((int)cin, (int)cout, (int)H, (int)W, (int)ks, S, &xs, &ws, &cp, &kp, &coutp) <= 158 * 1024 &&
        conv_stage_iters((int)cin, (int)(H * W), S) <= 16) return S;
  return 0;
}

int conv2d_same(const float* x, float* y, int64_t B, int64_t cin, int64_t cout, int64_t H, int64_t W, int64_t ks,
                const void* wplanes, const float* bias, const float* in_mul, int32_t in_act, float in_slope,
                int32_t out_act, float out_slope, const float* gate_x, int64_t gate_channels, hipStream_t stream) {
  if (B < 0 || cin <= 0 || cout <= 0 || H <= 0 || W <= 0 || cin > 64 || cout > 64 || H * W > 256 || (ks != 1 && ks != 3) ||
      B > 0x7fffffff) {
    set_error("usf_conv2d_same_f32: unsupported sizes (channels 1..64, H * W <= 256, kernel 1 or 3)");
    return -2;
  }
  if (B == 0) return 0;
  if (!x || !y || !wplanes) { set_error("usf_conv2d_same_f32: null pointer"); return -1; }
  if (x == y) { set_error("usf_conv2d_same_f32: in-place operation is not supported"); return -2; }
  if ((gate_x != nullptr) != (gate_channels > 0) || (gate_x && (cout != 32 * ((gate_channels + 15) / 16) || out_act != USF_ACT_NONE))) {
    set_error("usf_conv2d_same_f32: gated mode wants gate_x, gate_channels C > 0, cout = 32 * ceil(C / 16) interleaved rows, no output activation");
    return -2;
  }
  if (!aligned16(wplanes)) { set_error("usf_conv2d_same_f32: weight planes must be 16-byte aligned"); return -2; }
  for (int32_t act : {in_act, out_act})
    if (act != USF_ACT_NONE && act != USF_ACT_LEAKY_RELU) { set_error("usf_conv2d_same_f32: bad act"); return -2; }
  if (ks == 3 && !gate_x) {
    // the register-weight kernel where it serves the shape (the conditioner layers of the reference's image configurations)
    const int rc = conv2d_same_wreg(x, y, B, cin, cout, H, W, wplanes, bias, in_mul, in_act, in_slope, out_act, out_slope, nullptr,
                                    nullptr, 0.f, 0, stream);
    if (rc != 0) return rc < 0 ? rc : 0;
  }
  ConvArgs a;
  a.x = x; a.y = y; a.wp = reinterpret_cast<const __bf16*>(wplanes); a.bias = bias; a.in_mul = in_mul;
  a.B = (int)B; a.cin = (int)cin; a.cout = (int)cout; a.H = (int)H; a.W = (int)W; a.ks = (int)ks;
  a.in_act = in_act; a.out_act = out_act; a.in_slope = in_slope; a.out_slope = out_slope;
  a.dbg = (int)tuning("conv_dbg", 0);
  a.gate_x = gate_x; a.gateC = (int)gate_channels;
  // samples per group: as many as fit 158 KB of LDS (at most 8; at least one has to fit)
  int S = 8;
  int64_t lds = 0;
  for (; S >= 1; --S) {
    lds = conv_lds_bytes(a.cin, a.cout, a.H, a.W, a.ks, S, &a.xs16, &a.ws16, &a.cp, &a.kp, &a.coutp);
    if (lds <= 158 * 1024 && conv_stage_iters(a.cin, a.H * a.W, S) <= 16) break;
  }
  if (S < 1) { set_error("usf_conv2d_same_f32: one sample does not fit the LDS / staging registers (%lld bytes)", (long long)lds); return -3; }
  if (S > B) {
    S = (int)B;
    lds = conv_lds_bytes(a.cin, a.cout, a.H, a.W, a.ks, S, &a.xs16, &a.ws16, &a.cp, &a.kp, &a.coutp);
  }
  a.S = S;
  static bool attr_done_dev[USF_MAX_DEVICES] = {false};      // (the attribute belongs to the device)
  bool& attr_done = attr_done_dev[current_device_slot()];
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2d_same_bf16x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess) {
      set_error("usf_conv2d_same_f32: cannot raise the LDS limit");
      return -4;
    }
    attr_done = true;
  }
  const int cus = device_cu_count();
  const int64_t ngroups = (B + S - 1) / S;
  const unsigned grid = (unsigned)(ngroups < cus ? ngroups : cus);
  hipLaunchKernelGGL(conv2d_same_bf16x3_kernel, dim3(grid), dim3(512), (size_t)lds, stream, a);
  return check_launch("usf_conv2d_same_f32");
}

// y = res_x + res_sign * (res_mul * conv(...)): the last convolution of a coupling's conditioner with MaskedCoupling's
// residual in its output stream.  Only where the register-weight kernel serves the shape: 0 = done, 1 = not served here (the
// caller runs the convolution and usf_masked_residual_f32 as two passes), < 0 = error.
int conv2d_same_res(const float* x, float* y, int64_t B, int64_t cin, int64_t cout, int64_t H, int64_t W, int64_t ks,
                    const void* wplanes, const float* bias, const float* in_mul, int32_t in_act, float in_slope,
                    const float* res_x, const float* res_mul, float res_sign, hipStream_t stream) {
  if (B < 0 || cin <= 0 || cout <= 0 || H <= 0 || W <= 0 || B > 0x7fffffff) { set_error("usf_conv2d_same_res_f32: bad sizes"); return -2; }
  if (B == 0) return 0;
  if (!x || !y || !wplanes || !res_x || !res_mul) { set_error("usf_conv2d_same_res_f32: null pointer"); return -1; }
  if (x == y || res_x == y) { set_error("usf_conv2d_same_res_f32: in-place operation is not supported"); return -2; }
  if (in_act != USF_ACT_NONE && in_act != USF_ACT_LEAKY_RELU) { set_error("usf_conv2d_same_res_f32: bad act"); return -2; }
  if (ks != 3) return 1;
  const int rc = conv2d_same_wreg(x, y, B, cin, cout, H, W, wplanes, bias, in_mul, in_act, in_slope, USF_ACT_NONE, 0.f, res_x, res_mul,
                                  res_sign, 0, stream);
  return rc < 0 ? rc : (rc == 1 ? 0 : 1);
}

// y = conv(x) * (gate_h > 0 ? 1 : gate_slope) * gate_mul: a data-gradient convolution with the (Leaky)ReLU and mask factors
// of the layer's INPUT in its output stream (what usf_act_grad_f32 and a mask product would do in two more passes).
// 0 = done, 1 = not served here, < 0 = error.
int conv2d_same_gate(const float* x, float* y, int64_t B, int64_t cin, int64_t cout, int64_t H, int64_t W, int64_t ks,
                     const void* wplanes, const float* gate_h, float gate_slope, const float* gate_mul, const float* gate_add,
                     hipStream_t stream) {
  if (B < 0 || cin <= 0 || cout <= 0 || H <= 0 || W <= 0 || B > 0x7fffffff) { set_error("usf_conv2d_same_gate_f32: bad sizes"); return -2; }
  if (B == 0) return 0;
  if (!x || !y || !wplanes || !gate_h) { set_error("usf_conv2d_same_gate_f32: null pointer"); return -1; }
  if (x == y || gate_h == y) { set_error("usf_conv2d_same_gate_f32: in-place operation is not supported"); return -2; }
  if (ks != 3) return 1;
  if (gate_add && gate_mul) { set_error("usf_conv2d_same_gate_f32: gate_mul and gate_add exclude each other"); return -2; }
  if (gate_add == y) { set_error("usf_conv2d_same_gate_f32: in-place operation is not supported"); return -2; }
  const int rc = conv2d_same_wreg(x, y, B, cin, cout, H, W, wplanes, nullptr, nullptr, USF_ACT_NONE, 0.f, USF_ACT_NONE, 0.f, gate_h,
                                  gate_add ? gate_add : gate_mul, gate_slope, gate_add ? 2 : 1, stream);
  return rc < 0 ? rc : (rc == 1 ? 0 : 1);
}

}  // namespace usf
