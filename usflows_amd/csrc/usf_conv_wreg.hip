// 3 x 3 "same" convolution of the CNN conditioner (networks.py:405-510 ConvNet2D, :61-122 GatedConv), second kernel of
// usf_conv2d_same_f32: WEIGHTS IN REGISTERS, the three phases of a sample group overlapped.
//
// conv2d_same_bf16x3_kernel (usf_conv.hip) runs a group of samples as stage -> barrier -> matrix phase -> stores ->
// barrier: its phases add up (profiles/r02_tuning_experiments.md section 5: the matrix phase is 0.23 of 0.67 ms at the
// MNIST configuration's 32 -> 32 layer) and 57 of its 158 KB of LDS hold the weight planes.  Here
//   * a wave owns ONE 16-channel output tile for the whole launch and keeps that tile's weight fragments (all k-blocks,
//     three bf16 planes: 60 / 108 registers at 16 / 32 input channels) in registers: no weight image in LDS,
//     no weight fragment reads in the k loop -- the LDS read traffic of the matrix phase halves;
//   * the freed LDS double-buffers the input image (bf16x3 planes, groups of 8 channels x positions, NO zero border: a
//     tap that leaves the picture reads a shared all-zero position instead) and holds an fp32 output staging area in the tensor's own flat
//     [sample][channel][pixel] order, so both HBM streams are plain 16-byte-per-lane runs over contiguous memory
//     (the old kernel: 4-byte loads, 4-byte stores in 64-byte segments);
//   * per group every wave: issues the next group's loads, multiplies its row tiles (16 rows of (sample, pixel) x its 16
//     channels: 6 MFMAs per k-block against a B operand read from the image), streams the previous group's staged
//     outputs to HBM, splits and stores the next group's image -- two barriers per group; the two waves of a SIMD run
//     the same program, the hardware interleaves one wave's vector / memory work with the other's MFMAs.
// Arithmetic: unchanged (six v_mfma_f32_16x16x32_bf16 per fp32-equivalent product, fp32 accumulation, K order
// tap-major / channel-minor, smallest terms first).  Served: 3 x 3 kernels, 16 / 32 input channels (k-block counts
// 5 / 9; 48 channels = 14 k-blocks = 168 weight registers do not fit beside the rest), 16 / 32 / 64 output channels
// (1 / 2 / 4 tiles: 8 / 4 / 2 waves each), H W <= 64 with (channels x H W) a multiple of 4, no gate; everything else
// stays on the first kernel.
#include <stdlib.h>

#include "usf_common.h"

namespace usf {

typedef __bf16 cw_bf16x8 __attribute__((ext_vector_type(8)));

struct ConvWArgs {
  const float* x; float* y;
  const __bf16* wp;            // [3][coutp][kp]
  const float* bias;           // [cout] or null
  const float* in_mul;         // [cin * hw] or null
  int B, cin, cout, H, W;
  int kp, coutp;
  int S;                       // samples per group
  int cgs;                     // bytes of one channel group (8 channels) of an image plane: (S * HW positions + the zero one) * 16, rounded up to 256
  int img_bytes;               // bytes of one plane of one image buffer: (cin / 8) * cgs
  int in_act, out_act; float in_slope, out_slope;
  const float* res_x; const float* res_mul; float res_sign;   // y = res_x + res_sign * (res_mul * conv): MaskedCoupling's residual, or null
  int res_mode;                // 1: gate instead -- y = conv * (res_x > 0 ? 1 : res_sign) * res_mul (res_mul may be null): the data gradient's
                               // (Leaky)ReLU / mask factors, see usf_conv2d_same_gate_f32; 2: y = res_mul + conv * (res_x > 0 ? 1 : res_sign) with
                               // res_mul a FULL [B, cout, H, W] tensor (the other branch's gradient where the forward input forks)
  unsigned mSO;                // magic of cout * H W
  int dbg;                     // tuning aid (USF_CONVW_DBG; wrong results): 1 no staging, 2 no k loop, 4 no output flush, 8 no input loads,
                               // 16 no staging-area writes, 32 no barriers
  unsigned mHW, mW, mSE;       // floor(2^32 / d) + 1 for d = H W, W, cin H W: n / d == umulhi(n, m) for n < 2^16
};

__device__ __forceinline__ int cw_div(int n, unsigned m) { return (int)__umulhi((unsigned)n, m); }

__device__ __forceinline__ void cw_split(float x, __bf16& h, __bf16& m, __bf16& l) {
  h = (__bf16)x;
  const float r = x - (float)h;
  m = (__bf16)r;
  l = (__bf16)(r - (float)m);
}

// NB: k-blocks of 32 (= kp / 32), CP: input channels padded to a multiple of 8, NCT: 16-channel output tiles
template <int NB, int CP, int NCT>
__global__ __launch_bounds__(512, 2) void conv2d_same_wreg_kernel(const ConvWArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int WPC = 8 / NCT;                        // waves per output tile
  constexpr int MAXRT = 4;                            // row tiles per wave and group (host: ceil(S HW / 16) <= MAXRT * WPC)
  constexpr int MAXIT = 2;                            // staging units (8 channels of one position) per thread and group (host checks)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int ct = wave / WPC, wi = wave % WPC;
  const int HW = a.H * a.W;
  const int cgs = a.cgs;
  const int plane = a.img_bytes;                       // one plane of one buffer
  // image: [2 buffers][3 planes][channel group of 8][S HW + 1 positions] x 16 bytes.  Channel-group-major with a group
  // stride that is a multiple of 256 bytes: a ds_read_b128 is served in lane groups {0-3, 12-15, 20-27}, {4-11, 16-19,
  // 28-31}, ... (MI355X_MICROARCH.md, LDS) -- rows li of one channel group next to rows li of the next one -- and 16
  // consecutive positions of any channel groups are 16 different 16-byte bank groups (position-major rows of 80 bytes
  // collide 2-way under that grouping)
  unsigned char* const img = smem;
  float* const ostage = reinterpret_cast<float*>(smem + 6 * plane);     // [S][cout][HW] fp32
  const int zpos = a.S * HW * 16;                      // byte offset of the all-zero position inside a channel group

  // ---- once per block: this wave's weight fragments, zeroed images (padding channels and the zero rows stay zero) ----
  cw_bf16x8 wreg[NB][3];
  {
    const int co = min(ct * 16 + li, a.coutp - 1);
#pragma unroll
    for (int blk = 0; blk < NB; ++blk)
#pragma unroll
      for (int q = 0; q < 3; ++q)
        wreg[blk][q] = *reinterpret_cast<const cw_bf16x8*>(a.wp + ((size_t)(q * a.coutp + co) * a.kp + 32 * blk + 8 * lg));
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (int i = tid; i < 6 * plane / 16; i += 512) *reinterpret_cast<f32x4*>(img + i * 16) = z;
  }
  // per k-block (the same for every row tile): which tap this lane's 8 k belong to and the byte offset that tap and the
  // lane's channel group add to a row's image position; packed: bits 0..3 tap (9: K padding -> always the zero position;
  // the weights are zero there too), bits 4.. signed offset (dy W + dx) 16 + channel group * cgs
  int tapinfo[NB];
#pragma unroll
  for (int blk = 0; blk < NB; ++blk) {
    const int kflat = 32 * blk + 8 * lg;
    const int tap = min(kflat / CP, 9);
    const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
    tapinfo[blk] = tap | (((dy * a.W + dx) * 16 + ((kflat - (kflat / CP) * CP) >> 3) * cgs) << 4);
  }
  __syncthreads();

  const int ngroups = (a.B + a.S - 1) / a.S;
  const int sample_elems = a.cin * HW;                 // a multiple of 4 (host)
  const bool leaky_in = a.in_act == USF_ACT_LEAKY_RELU;
  // staging piece `it` of this thread: elements 4 f .. 4 f + 3 of the group's contiguous input chunk, f = tid + 512 it
  // Staging unit u = ((sample, channel group of 8), position): a lane loads the unit's 8 channel values (eight 4-byte loads,
  // each coalesced across the lanes' consecutive positions), splits them and writes ONE 16-byte LDS store per plane --
  // consecutive lanes, consecutive 16-byte slots: conflict-free.  (Per-element 2-byte stores of a float4-per-lane staging
  // landed 8-way on the banks: 0.10 of 0.55 ms at the MNIST configuration's 32 -> 32 layer.)  Unit `it` of this thread:
  // u = tid + 512 it; its source offset and LDS slot are the same in every group.
  constexpr int NCG = CP / 8;
  float pre[MAXIT][8];
  int usrc[MAXIT], udst[MAXIT];                        // element offset of channel 0 of the unit inside the group's chunk; LDS byte offset
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) {
    const int u = tid + 512 * it;
    const int sc = cw_div(u, a.mHW), p = u - sc * HW;   // sc = sample * NCG + channel group
    const int sl = sc / NCG, cg = sc - sl * NCG;
    usrc[it] = (sl * a.cin + 8 * cg) * HW + p;
    udst[it] = cg * cgs + (sl * HW + p) * 16;
  }
  auto issue_loads = [&](int gidx) {
    const int s0 = gidx * a.S;
    const int nu = min(a.S, a.B - s0) * NCG * HW;
    const float* xg = a.x + (size_t)s0 * sample_elems;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      if (tid + 512 * it < nu) {
#pragma unroll
        for (int j = 0; j < 8; ++j) pre[it][j] = xg[usrc[it] + j * HW];
      }
    }
  };
  auto stage = [&](int gidx, int which) {
    const int s0 = gidx * a.S;
    const int nu = min(a.S, a.B - s0) * NCG * HW;
    unsigned char* const buf = img + which * 3 * plane;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      if (tid + 512 * it < nu) {
        cw_bf16x8 h, m, l;
        const int mrem = usrc[it] - cw_div(usrc[it], a.mSE) * sample_elems;    // channel 0's index inside its sample (the mask's)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float v = pre[it][j];
          if (leaky_in) v = v > 0.0f ? v : v * a.in_slope;
          if (a.in_mul) v *= a.in_mul[mrem + j * HW];
          __bf16 hh, mm, ll;
          cw_split(v, hh, mm, ll);
          h[j] = hh; m[j] = mm; l[j] = ll;
        }
        unsigned char* dst = buf + udst[it];
        *reinterpret_cast<cw_bf16x8*>(dst) = h;
        *reinterpret_cast<cw_bf16x8*>(dst + plane) = m;
        *reinterpret_cast<cw_bf16x8*>(dst + 2 * plane) = l;
      }
    }
  };
  // the previous group's staged outputs -> HBM: one contiguous chunk, 16 bytes per lane
  auto flush = [&](int gidx) {
    const int s0 = gidx * a.S;
    const int n4 = min(a.S, a.B - s0) * a.cout * HW / 4;
    f32x4* yg = reinterpret_cast<f32x4*>(a.y + (size_t)s0 * a.cout * HW);
    if (a.res_x) {
      // MaskedCoupling's residual (transforms.py:277-306) joined to the output stream: x +- (1 - mask) * t in the arithmetic of
      // usf_masked_residual_f32 (x + sign * (om * t)) -- the [B, C, H, W] tensor t never reaches HBM
      const int so = a.cout * HW;
      const f32x4* xg = reinterpret_cast<const f32x4*>(a.res_x + (size_t)s0 * so);
      // (all of the thread's residual loads are in flight before the first one is used: a loop that waits for each in
      //  turn pays the HBM latency four times per group)
      constexpr int NF = 4;                              // host: S * cout * HW / 4 <= NF * 512
      f32x4 xv[NF], om[NF];
#pragma unroll
      for (int i = 0; i < NF; ++i) {
        const int f = min(tid + 512 * i, n4 - 1), e0 = 4 * f;
        xv[i] = xg[f];
        om[i] = (f32x4){1.f, 1.f, 1.f, 1.f};
        if (a.res_mul) om[i] = *reinterpret_cast<const f32x4*>(a.res_mul + (e0 - cw_div(e0, a.mSO) * so));
      }
#pragma unroll
      for (int i = 0; i < NF; ++i) {
        const int f = tid + 512 * i;
        if (f < n4) {
          const f32x4 t = *reinterpret_cast<const f32x4*>(ostage + 4 * f);
          f32x4 o;
#pragma unroll
          for (int j = 0; j < 4; ++j)
            o[j] = a.res_mode ? (xv[i][j] > 0.f ? t[j] : t[j] * a.res_sign) * om[i][j] : xv[i][j] + a.res_sign * (om[i][j] * t[j]);
          yg[f] = o;
        }
      }
      return;
    }
    for (int f = tid; f < n4; f += 512) yg[f] = *reinterpret_cast<const f32x4*>(ostage + 4 * f);
  };

  int first = blockIdx.x;
  if (first < ngroups) { issue_loads(first); stage(first, 0); }
  __syncthreads();
  int prev = -1, cur = 0;
  for (int gidx = first; gidx < ngroups; gidx += gridDim.x) {
    const int s0 = gidx * a.S;
    const int R = min(a.S, a.B - s0) * HW;               // live rows of this group
    const int nrt = (R + 15) >> 4;
    const int nxt = gidx + gridDim.x;
    if (nxt < ngroups && !(a.dbg & 8)) issue_loads(nxt);
    const unsigned char* const buf = img + cur * 3 * plane;
    f32x4 res[MAXRT];
#pragma unroll
    for (int t = 0; t < MAXRT; ++t) {
      const int rt = wi + WPC * t;
      res[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (rt < nrt && !(a.dbg & 2)) {                    // wave-uniform
        const int r = min(rt * 16 + li, R - 1);
        const int sl = cw_div(r, a.mHW), p = r - sl * HW;
        const int py = cw_div(p, a.mW), px = p - py * a.W;
        const int base = r * 16;                         // (sl * HW + p == r: the image is the group's rows in order)
        // which of the nine taps stay inside the picture for this row's pixel (bit 9, the K padding, stays clear)
        const unsigned rowok = (py > 0 ? 0x007u : 0u) | 0x038u | (py + 1 < a.H ? 0x1c0u : 0u);
        const unsigned colok = (px > 0 ? 0x049u : 0u) | 0x092u | (px + 1 < a.W ? 0x124u : 0u);
        const unsigned vmask = rowok & colok;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        cw_bf16x8 xf[2][3];
        auto read_blk = [&](int blk, cw_bf16x8 (&xv)[3]) {
          const bool ok = (vmask >> (tapinfo[blk] & 15)) & 1u;
          const int off = ok ? base + (tapinfo[blk] >> 4) : zpos;     // (any channel group's zero position will do)
#pragma unroll
          for (int q = 0; q < 3; ++q) xv[q] = *reinterpret_cast<const cw_bf16x8*>(buf + q * plane + off);
        };
        read_blk(0, xf[0]);
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) {
          if (blk + 1 < NB) read_blk(blk + 1, xf[(blk + 1) & 1]);
          const cw_bf16x8 (&xv)[3] = xf[blk & 1];
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[blk][2], xv[0], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[blk][1], xv[1], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[blk][0], xv[2], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[blk][1], xv[0], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[blk][0], xv[1], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[blk][0], xv[0], acc, 0, 0, 0);
        }
        res[t] = acc;
      }
      if (t == 1 && prev >= 0 && !(a.dbg & 4)) flush(prev);   // the previous group's outputs leave under the matrix work
      // the two waves of a SIMD (w and w + 4) split and store the next group's image at different points of the group:
      // one of them is multiplying while the other one's vector instructions run
      if (t == 1 && wave >= 4 && nxt < ngroups && !(a.dbg & 1)) stage(nxt, cur ^ 1);
    }
    if (wave < 4 && nxt < ngroups && !(a.dbg & 1)) stage(nxt, cur ^ 1);
    if (!(a.dbg & 32)) __syncthreads();                  // every wave is done with the staged outputs of the previous group
    // ---- this group's results -> staging area: lane (li, lg) of a row tile holds channels ct * 16 + 4 lg + (0..3) of row li ----
#pragma unroll
    for (int t = 0; t < MAXRT; ++t) {
      const int rt = wi + WPC * t;
      const int r = rt * 16 + li;
      if (rt < nrt && r < R && !(a.dbg & 16)) {
        const int sl = cw_div(r, a.mHW), p = r - sl * HW;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int co = ct * 16 + 4 * lg + j;
          if (co < a.cout) {
            const float v = res[t][j] + (a.bias ? a.bias[co] : 0.f);
            ostage[(sl * a.cout + co) * HW + p] = act_apply(v, a.out_act, a.out_slope);
          }
        }
      }
    }
    if (!(a.dbg & 32)) __syncthreads();                  // staged outputs and the next group's image are complete
    prev = gidx;
    cur ^= 1;
  }
  if (prev >= 0) flush(prev);
}

// ------------------------------------------------------------------------------------------
// The same convolution with the block's waves SPECIALISED (round 3, second step): in the kernel above every wave multiplies,
// stages, flushes in turn and the two waves of a SIMD run the same program in step, so the phases still add up (18.3 k
// cycles per group of which 6.9 k are MFMA issue, MNIST 32 -> 32).  Here
//   * waves 0-3 (one per SIMD) do nothing but read image fragments and multiply: each owns one 16-channel output tile (its
//     weight fragments in registers, as above) and 4 / NCT of the waves share a tile's row tiles; a row tile's results go
//     straight into the fp32 output staging area;
//   * waves 4-7 (the other wave of each SIMD) do everything else: the next group's input loads (two groups ahead), the
//     activation / mask / three-way split and the image stores of the next group, and the previous group's staged outputs
//     -> HBM with the residual / gate factors.  Their vector and memory instructions issue in the gaps of the other wave's
//     MFMAs (an MFMA holds the SIMD's issue port for 8 of its 16 cycles);
//   * image AND output staging area are double-buffered, ONE barrier per group.
// Arithmetic and summation order are those of the kernel above: bit-identical results.
// ------------------------------------------------------------------------------------------
template <int NB, int CP, int NCT>
__global__ __launch_bounds__(512, 2) void conv2d_same_wsp_kernel(const ConvWArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NMW = (NCT == 1) ? 2 : 4;             // multiplying waves (16 output channels: half the matrix work, the service side is the bottleneck -> 2 + 6)
  constexpr int NSV = 512 - 64 * NMW;                 // service threads
  constexpr int WPC = NMW / NCT;                      // multiplying waves per output tile
  constexpr int MAXRT = 16 / WPC;                     // row tiles per multiplying wave and group (host: ceil(S HW / 16) <= 16)
  constexpr int MAXIT = (1024 + NSV - 1) / NSV;       // staging units per service thread and group (host: at most 1024 units)
  constexpr int NCG = CP / 8;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int HW = a.H * a.W;
  const int cgs = a.cgs;
  const int bufbytes = a.img_bytes;                   // one image buffer: every channel group's positions with the three planes side by side
  // image: [2 buffers][channel group][S HW + 1 positions][3 planes] x 16 bytes.  The planes of a position are neighbours, so a
  // fragment's three reads (and a staging unit's three stores) differ by an IMMEDIATE offset: one address register per k-block --
  // the multiplying wave has the SIMD's MFMA stream to itself and every vector instruction in its loop costs issue time.
  // Positions are 48 bytes apart: 16 consecutive positions start at banks 12 i mod 64, 4 banks each -- all disjoint; the channel-
  // group stride is a multiple of 256 bytes (the ds_read_b128 lane groups mix rows of neighbouring channel groups).
  unsigned char* const img = smem;
  const int ost_floats = a.S * a.cout * HW;
  float* const ostage = reinterpret_cast<float*>(smem + 2 * bufbytes);   // [2][S][cout][HW] fp32
  const int zpos = a.S * HW * 48;
  {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (int i = tid; i < 2 * bufbytes / 16; i += 512) *reinterpret_cast<f32x4*>(img + i * 16) = z;
  }
  __syncthreads();
  const int ngroups = (a.B + a.S - 1) / a.S;
  const int first = blockIdx.x, stride = gridDim.x;

  if (wave < NMW) {
    // ================= multiplying waves =================
    const int ct = wave / WPC, wi = wave % WPC;
    cw_bf16x8 wreg[NB][3];
    {
      const int co = min(ct * 16 + li, a.coutp - 1);
#pragma unroll
      for (int blk = 0; blk < NB; ++blk)
#pragma unroll
        for (int q = 0; q < 3; ++q)
          wreg[blk][q] = *reinterpret_cast<const cw_bf16x8*>(a.wp + ((size_t)(q * a.coutp + co) * a.kp + 32 * blk + 8 * lg));
    }
    int tapinfo[NB];
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
      const int kflat = 32 * blk + 8 * lg;
      const int tap = min(kflat / CP, 9);
      const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
      // packed: bits 0..9 the tap as a one-hot bit (bit 9: K padding -- never set in a row's tap mask), bits 10.. the signed offset
      tapinfo[blk] = (1 << tap) | (((dy * a.W + dx) * 48 + ((kflat - (kflat / CP) * CP) >> 3) * cgs) << 10);
    }
    float bias4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int co = ct * 16 + 4 * lg + j;
      bias4[j] = (a.bias && co < a.cout) ? a.bias[co] : 0.f;
    }
    // this wave's row tiles are the same rows in every group: image offset, tap mask and output offset of the lane's row,
    // once per kernel (a row past a short last group's end is masked by the row count at the store; its reads are harmless)
    struct RowTile { int base, ooff; unsigned vmask; bool live; };
    int tbase[MAXRT], toff[MAXRT];
    unsigned tmask[MAXRT];
#pragma unroll
    for (int t = 0; t < MAXRT; ++t) {
      const int r = min((wi + WPC * t) * 16 + li, a.S * HW - 1);
      const int sl = cw_div(r, a.mHW), p = r - sl * HW;
      const int py = cw_div(p, a.mW), px = p - py * a.W;
      const unsigned rowok = (py > 0 ? 0x007u : 0u) | 0x038u | (py + 1 < a.H ? 0x1c0u : 0u);
      const unsigned colok = (px > 0 ? 0x049u : 0u) | 0x092u | (px + 1 < a.W ? 0x124u : 0u);
      tbase[t] = r * 48;
      tmask[t] = rowok & colok;
      toff[t] = (sl * a.cout + ct * 16 + 4 * lg) * HW + p;
    }
    __syncthreads();                                    // (A) the first group's image is staged
    int cur = 0;
    for (int gidx = first; gidx < ngroups; gidx += stride) {
      const int s0 = gidx * a.S;
      const int R_ = min(a.S, a.B - s0) * HW;             // live rows of this group
      const int nrt = (R_ + 15) >> 4;
      const unsigned char* const buf = img + cur * bufbytes;
      float* const ost = ostage + cur * ost_floats;
      // The fragments of a k-block are read D blocks ahead of its MFMAs, ACROSS row-tile boundaries (a ring of R fragment
      // sets with NB a multiple of R, so the slots are static): one wave per SIMD has nobody to hide its LDS latency behind.
      constexpr int R = (NB % 3 == 0) ? 3 : ((NB % 2 == 0) ? 2 : NB), D = R - 1;   // 9 -> 3, 14 -> 2 (168 weight registers), 5 -> 5
      auto tile_of = [&](int t) {                          // (tables built once per kernel, below; t is a compile-time index)
        RowTile q;
        q.live = (wi + WPC * t) < nrt;
        q.base = tbase[t < MAXRT ? t : MAXRT - 1];
        q.ooff = toff[t < MAXRT ? t : MAXRT - 1];
        q.vmask = (q.live && t < MAXRT) ? tmask[t < MAXRT ? t : MAXRT - 1] : 0u;   // (a tile past the end reads the zero position only)
        return q;
      };
      cw_bf16x8 xf[R][3];
      auto read_blk = [&](const RowTile& q, int blk, cw_bf16x8 (&xv)[3]) {
        const bool ok = (q.vmask & (unsigned)tapinfo[blk]) != 0u;          // (the mask has bits 0..8 only)
        const int off = ok ? q.base + (tapinfo[blk] >> 10) : zpos;
#pragma unroll
        for (int qq = 0; qq < 3; ++qq) xv[qq] = *reinterpret_cast<const cw_bf16x8*>(buf + off + 16 * qq);
      };
      RowTile tc = tile_of(0);
#pragma unroll
      for (int b = 0; b < D; ++b) read_blk(tc, b, xf[b % R]);
#pragma unroll
      for (int t = 0; t < MAXRT; ++t) {
        if (!tc.live || (a.dbg & 2)) break;               // wave-uniform (dbg 2: no matrix work)
        const RowTile tn = tile_of(t + 1);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) {
          if (blk + D < NB) read_blk(tc, blk + D, xf[(blk + D) % R]);
          else read_blk(tn, blk + D - NB, xf[(blk + D) % R]);
          const cw_bf16x8 (&xv)[3] = xf[blk % R];
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[blk][2], xv[0], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[blk][1], xv[1], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[blk][0], xv[2], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[blk][1], xv[0], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[blk][0], xv[1], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[blk][0], xv[0], acc, 0, 0, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);   // this block's three fragment reads (for block blk + D) ...
          __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);   // ... then its six MFMAs: the reads stay D blocks ahead
        }
        // lane (li, lg) holds channels ct * 16 + 4 lg + (0..3) of row li
        if ((wi + WPC * t) * 16 + li < R_ && ct * 16 + 4 * lg < a.cout) {        // (cout is a multiple of 4 here: 16 / 32 / 64)
#pragma unroll
          for (int j = 0; j < 4; ++j) ost[tc.ooff + j * HW] = act_apply(acc[j] + bias4[j], a.out_act, a.out_slope);
        }
        tc = tn;
      }
      __syncthreads();                                  // (B) this group's outputs are staged, the next image is ready
      cur ^= 1;
    }
    __syncthreads();                                    // (C) the service waves' last flush has this group's outputs
    return;
  }

  // ================= service waves =================
  const int st = tid - 64 * NMW;                        // 0 .. NSV - 1
  const int sample_elems = a.cin * HW;
  // the input nonlinearity as a select (slope 1 = none) and the coupling mask as a factor per element that is the same in
  // every group (a unit is a fixed (sample slot, channel group, position)): straight-line staging, no loads, no branches
  const float slope_eff = (a.in_act == USF_ACT_LEAKY_RELU) ? a.in_slope : 1.f;
  float pre[MAXIT][8], mulv[MAXIT][8];
  int usrc[MAXIT], udst[MAXIT];
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) {
    const int u = st + NSV * it;
    const int sc = cw_div(u, a.mHW), p = u - sc * HW;   // sc = sample * NCG + channel group
    const int sl = sc / NCG, cg = sc - sl * NCG;
    usrc[it] = (sl * a.cin + 8 * cg) * HW + p;
    udst[it] = cg * cgs + (sl * HW + p) * 48;
    const bool live = u < a.S * NCG * HW;
#pragma unroll
    for (int j = 0; j < 8; ++j) mulv[it][j] = (a.in_mul && live) ? a.in_mul[(8 * cg + j) * HW + p] : 1.f;
  }
  auto issue_loads = [&](int gidx) {
    const int s0 = gidx * a.S;
    const int nu = min(a.S, a.B - s0) * NCG * HW;
    const float* xg = a.x + (size_t)s0 * sample_elems;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      if (st + NSV * it < nu) {
#pragma unroll
        for (int j = 0; j < 8; ++j) pre[it][j] = xg[usrc[it] + j * HW];
      }
    }
  };
  auto stage = [&](int gidx, int which) {
    const int s0 = gidx * a.S;
    const int nu = min(a.S, a.B - s0) * NCG * HW;
    unsigned char* const buf = img + which * bufbytes;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      if (st + NSV * it < nu) {
        cw_bf16x8 h, m, l;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float v = pre[it][j];
          v = (v > 0.0f ? v : v * slope_eff) * mulv[it][j];
          __bf16 hh, mm, ll;
          cw_split(v, hh, mm, ll);
          h[j] = hh; m[j] = mm; l[j] = ll;
        }
        unsigned char* dst = buf + udst[it];
        *reinterpret_cast<cw_bf16x8*>(dst) = h;
        *reinterpret_cast<cw_bf16x8*>(dst + 16) = m;
        *reinterpret_cast<cw_bf16x8*>(dst + 32) = l;
      }
    }
  };
  auto flush = [&](int gidx, int which) {
    const int s0 = gidx * a.S;
    const int n4 = min(a.S, a.B - s0) * a.cout * HW / 4;
    const float* const ost = ostage + which * ost_floats;
    f32x4* yg = reinterpret_cast<f32x4*>(a.y + (size_t)s0 * a.cout * HW);
    if (a.res_x) {
      const int so = a.cout * HW;
      const f32x4* xg = reinterpret_cast<const f32x4*>(a.res_x + (size_t)s0 * so);
      constexpr int NF = (2048 + NSV - 1) / NSV;         // host: S * cout * HW / 4 <= 2048
      f32x4 xv[NF], om[NF];
#pragma unroll
      for (int i = 0; i < NF; ++i) {
        const int f = min(st + NSV * i, n4 - 1), e0 = 4 * f;
        xv[i] = xg[f];
        om[i] = (f32x4){1.f, 1.f, 1.f, 1.f};
        if (a.res_mode == 2) om[i] = reinterpret_cast<const f32x4*>(a.res_mul + (size_t)s0 * so)[f];       // the addend: a full tensor
        else if (a.res_mul) om[i] = *reinterpret_cast<const f32x4*>(a.res_mul + (e0 - cw_div(e0, a.mSO) * so));
      }
#pragma unroll
      for (int i = 0; i < NF; ++i) {
        const int f = st + NSV * i;
        if (f < n4) {
          const f32x4 t = *reinterpret_cast<const f32x4*>(ost + 4 * f);
          f32x4 o;
#pragma unroll
          for (int j = 0; j < 4; ++j)
            o[j] = a.res_mode == 2 ? om[i][j] + (xv[i][j] > 0.f ? t[j] : t[j] * a.res_sign)
                   : a.res_mode   ? (xv[i][j] > 0.f ? t[j] : t[j] * a.res_sign) * om[i][j]
                                  : xv[i][j] + a.res_sign * (om[i][j] * t[j]);
          yg[f] = o;
        }
      }
      return;
    }
    for (int f = st; f < n4; f += NSV) yg[f] = *reinterpret_cast<const f32x4*>(ost + 4 * f);
  };

  if (first < ngroups) { issue_loads(first); stage(first, 0); }
  if (first + stride < ngroups) issue_loads(first + stride);
  __syncthreads();                                      // (A)
  int cur = 0, prev = -1;
  for (int gidx = first; gidx < ngroups; gidx += stride) {
    const int nxt = gidx + stride;
    if (nxt < ngroups && !(a.dbg & 1)) stage(nxt, cur ^ 1);               // (its loads were issued a group ago)
    if (nxt + stride < ngroups && !(a.dbg & 8)) issue_loads(nxt + stride);
    if (prev >= 0 && !(a.dbg & 4)) flush(prev, cur ^ 1);                  // the previous group's outputs leave while this one multiplies
    __syncthreads();                                      // (B)
    prev = gidx;
    cur ^= 1;
  }
  if (prev >= 0) flush(prev, cur ^ 1);
  __syncthreads();                                        // (C)
}

// samples per group for the register-weight kernel (0: the shape is not served by it)
static int conv_wreg_plan(int cin, int cout, int H, int W, int* cgs, int* img_bytes, int64_t* lds) {
  const int HW = H * W;
  if (!((cin == 16 || cin == 32) && (cout == 16 || cout == 32 || cout == 64)) || HW > 64 || HW < 1 ||
      ((cin * HW) & 3) || ((cout * HW) & 3)) return 0;
  const int wpc = 8 / ((cout + 15) / 16);
  auto group_bytes = [&](int S) { return (int64_t)(((S * HW + 1) * 16 + 255) / 256 * 256); };
  int best = 0; double best_eff = 0.0;
  for (int S = 1; S <= 16; ++S) {
    const int nrt = (S * HW + 15) / 16;
    if (nrt > 4 * wpc) break;
    const int64_t plane = (cin / 8) * group_bytes(S);
    const int64_t bytes = 6 * plane + (int64_t)S * cout * HW * 4;
    if (bytes > 158 * 1024) break;
    if (((int64_t)S * (cin / 8) * HW + 511) / 512 > 2) break;
    // fill of the wave slots of the matrix phase (rows in 16-row tiles, tiles dealt over wpc waves), the more samples the
    // fewer barriers per row: ties go to the larger group
    const double eff = (double)(S * HW) / (16.0 * wpc * ((nrt + wpc - 1) / wpc));
    if (eff >= best_eff - 1e-9) { best_eff = eff; best = S; }
  }
  if (best == 0 || best_eff < 0.7) return 0;
  *cgs = (int)group_bytes(best);
  *img_bytes = (cin / 8) * (*cgs);
  *lds = 6LL * (*img_bytes) + (int64_t)best * cout * HW * 4;
  return best;
}

// samples per group for the wave-specialised kernel (0: not served): 16 row tiles per group at most (dealt over the 4 / NCT
// multiplying waves of an output tile), 4 staging units per service thread, image and output staging area double-buffered
// B / cus > 0: a launch of B samples on cus compute units.  When the groups of the best S would not even fill the chip once,
// the SMALLEST servable S whose groups still fit one round of blocks is taken instead: at 100 samples a block then stages,
// multiplies and stores one sample instead of five -- the launch is a dependent chain of those phases, not a throughput
// problem (same sums per output element whatever S).
static int conv_wsp_plan(int cin, int cout, int H, int W, bool with_res, int* cgs, int* img_bytes, int64_t* lds, int64_t B = 0,
                         int cus = 0) {
  const int HW = H * W;
  if (!((cin == 16 || cin == 32 || cin == 48) && (cout == 16 || cout == 32 || cout == 48 || cout == 64)) || HW > 64 || HW < 1 ||
      ((cin * HW) & 3) || ((cout * HW) & 3)) return 0;
  const int nct_k = (cout + 15) / 16 == 3 ? 4 : (cout + 15) / 16;       // (48 channels: the four-tile instance, one multiplying wave idle)
  const int wpc = (nct_k == 1 ? 2 : 4) / nct_k;                         // (16 channels: two multiplying waves, six service waves)
  auto group_bytes = [&](int S) { return (int64_t)(((S * HW + 1) * 48 + 255) / 256 * 256); };   // three planes per position
  int best = 0, small = 0; double best_eff = 0.0;
  for (int S = 1; S <= 16; ++S) {
    const int nrt = (S * HW + 15) / 16;
    if (nrt > 16) break;
    const int64_t bufb = (cin / 8) * group_bytes(S);
    const int64_t bytes = 2 * bufb + 2LL * S * cout * HW * 4;
    if (bytes > 158 * 1024) break;
    if ((int64_t)S * (cin / 8) * HW > 4 * 256) break;
    if (with_res && (int64_t)S * cout * HW / 4 > 8 * 256) break;
    const double eff = (double)(S * HW) / (16.0 * wpc * ((nrt + wpc - 1) / wpc));
    if (eff >= best_eff - 1e-9) { best_eff = eff; best = S; }
    if (small == 0 && eff >= 0.7 && B > 0 && cus > 0 && (B + S - 1) / S <= cus) small = S;
  }
  if (best == 0 || best_eff < 0.7) return 0;
  const int small_on = (int)tuning("conv_small_s", 1);     // tuning aid: 0 = always the best S
  if (small_on && small > 0 && small < best && B > 0 && (B + best - 1) / best < cus) best = small;
  *cgs = (int)group_bytes(best);
  *img_bytes = (cin / 8) * (*cgs);                      // one image BUFFER (all three planes)
  *lds = 2LL * (*img_bytes) + 2LL * best * cout * HW * 4;
  return best;
}

int conv2d_same_wreg_fits(int64_t cin, int64_t cout, int64_t H, int64_t W) {
  int cgs, ib; int64_t lds;
  if (cin > 64 || cout > 64 || H * W > 256 || cin < 1 || cout < 1) return 0;
  return conv_wreg_plan((int)cin, (int)cout, (int)H, (int)W, &cgs, &ib, &lds);
}

// returns 1 when the launch was made, 0 when the shape is not served (the caller uses the first kernel), < 0 on error
int conv2d_same_wreg(const float* x, float* y, int64_t B, int64_t cin, int64_t cout, int64_t H, int64_t W, const void* wplanes,
                     const float* bias, const float* in_mul, int32_t in_act, float in_slope, int32_t out_act, float out_slope,
                     const float* res_x, const float* res_mul, float res_sign, int res_mode, hipStream_t stream) {
  if (!tuning("conv_wreg", 1)) return 0;                   // tuning aid: 0 = first kernel only
  ConvWArgs a;
  int64_t lds = 0;
  const int wsp = (int)tuning("conv_wsp", 1);              // tuning aid: 0 = the unspecialised kernel
  int S = 0;
  bool specialised = false;
  if (wsp && cin <= 64 && cout <= 64 && H * W <= 64) {
    S = conv_wsp_plan((int)cin, (int)cout, (int)H, (int)W, res_x != nullptr, &a.cgs, &a.img_bytes, &lds, B, device_cu_count());
    specialised = S > 0;
  }
  if (!specialised && cin == 48) return 0;
  if (!specialised)
    S = conv2d_same_wreg_fits(cin, cout, H, W) ? conv_wreg_plan((int)cin, (int)cout, (int)H, (int)W, &a.cgs, &a.img_bytes, &lds) : 0;
  if (S == 0 || !aligned16(x) || !aligned16(y) || (in_mul && !aligned16(in_mul)) ||
      (res_x && ((!res_mul && !res_mode) || !aligned16(res_x) || (res_mul && !aligned16(res_mul)) || res_x == y))) return 0;
  if (!specialised && res_x && ((int64_t)S * cout * H * W / 4 > 4 * 512 || res_mode == 2)) return 0;
  a.res_x = res_x; a.res_mul = res_mul; a.res_sign = res_sign; a.res_mode = res_mode;
  a.mSO = (unsigned)(0x100000000ULL / (uint64_t)(cout * H * W)) + 1u;
  a.x = x; a.y = y; a.wp = reinterpret_cast<const __bf16*>(wplanes); a.bias = bias; a.in_mul = in_mul;
  a.B = (int)B; a.cin = (int)cin; a.cout = (int)cout; a.H = (int)H; a.W = (int)W;
  a.coutp = (int)((cout + 15) / 16 * 16); a.kp = (int)((9 * cin + 31) / 32 * 32);
  a.dbg = (int)tuning("convw_dbg", 0);
  a.S = S; a.in_act = in_act; a.out_act = out_act; a.in_slope = in_slope; a.out_slope = out_slope;
  a.mHW = (unsigned)(0x100000000ULL / (uint64_t)(H * W)) + 1u; a.mW = (unsigned)(0x100000000ULL / (uint64_t)W) + 1u;
  a.mSE = (unsigned)(0x100000000ULL / (uint64_t)(cin * H * W)) + 1u;
  const int cus = device_cu_count();
  const int64_t ngroups = (B + S - 1) / S;
  const unsigned grid = (unsigned)(ngroups < cus ? ngroups : cus);
  const int nct = a.coutp / 16;
#define USF_CW(NB_, CP_, NCT_)                                                                                     \
  do {                                                                                                              \
    static bool attr_done_dev[USF_MAX_DEVICES] = {false};                                                           \
    bool& attr_done = attr_done_dev[current_device_slot()];                                                         \
    if (!attr_done) {                                                                                               \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2d_same_wreg_kernel<NB_, CP_, NCT_>),              \
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {              \
        set_error("usf_conv2d_same_f32: cannot raise the LDS limit");                                               \
        return -4;                                                                                                  \
      }                                                                                                             \
      attr_done = true;                                                                                             \
    }                                                                                                               \
    if (specialised) {                                                                                              \
      static bool attr2_dev[USF_MAX_DEVICES] = {false};                                                             \
      bool& attr2 = attr2_dev[current_device_slot()];                                                               \
      if (!attr2) {                                                                                                 \
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2d_same_wsp_kernel<NB_, CP_, NCT_>),             \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {            \
          set_error("usf_conv2d_same_f32: cannot raise the LDS limit");                                             \
          return -4;                                                                                                \
        }                                                                                                           \
        attr2 = true;                                                                                               \
      }                                                                                                             \
      hipLaunchKernelGGL((conv2d_same_wsp_kernel<NB_, CP_, NCT_>), dim3(grid), dim3(512), (size_t)lds, stream, a);  \
    } else {                                                                                                        \
      hipLaunchKernelGGL((conv2d_same_wreg_kernel<NB_, CP_, NCT_>), dim3(grid), dim3(512), (size_t)lds, stream, a); \
    }                                                                                                               \
  } while (0)
#define USF_CW_SP(NB_, CP_, NCT_)                                                                                  \
  do {                                                                                                              \
    static bool attr3_dev[USF_MAX_DEVICES] = {false};                                                               \
    bool& attr3 = attr3_dev[current_device_slot()];                                                                 \
    if (!attr3) {                                                                                                   \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2d_same_wsp_kernel<NB_, CP_, NCT_>),               \
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {              \
        set_error("usf_conv2d_same_f32: cannot raise the LDS limit");                                               \
        return -4;                                                                                                  \
      }                                                                                                             \
      attr3 = true;                                                                                                 \
    }                                                                                                               \
    hipLaunchKernelGGL((conv2d_same_wsp_kernel<NB_, CP_, NCT_>), dim3(grid), dim3(512), (size_t)lds, stream, a);    \
  } while (0)
#define USF_CW_NCT(NB_, CP_)                                                                                       \
  do { if (nct == 1) USF_CW(NB_, CP_, 1); else if (nct == 2) USF_CW(NB_, CP_, 2); else USF_CW(NB_, CP_, 4); } while (0)
  if (cin == 16) USF_CW_NCT(5, 16);
  else if (cin == 32) USF_CW_NCT(9, 32);
  else {                                                  // 48 input channels: the specialised kernel only
    if (nct == 1) USF_CW_SP(14, 48, 1); else if (nct == 2) USF_CW_SP(14, 48, 2); else USF_CW_SP(14, 48, 4);
  }
#undef USF_CW_NCT
#undef USF_CW_SP
#undef USF_CW
  int rc = check_launch("usf_conv2d_same_f32");
  return rc ? rc : 1;
}

}  // namespace usf
