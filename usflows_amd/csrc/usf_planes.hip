// "Planes pipeline": the dense layers of a flow with activations travelling BETWEEN layers as pre-split bf16 planes in
// an MFMA-fragment-tiled layout (large batches; usf_pack_planes_f32 / usf_gemm_planes_bf16x3 in the header).
//
// Why: in usf_linear_bf16x3.hip every block re-splits its fp32 activation slab into three bf16 planes inside the K
// loop (~100 VALU per 32-k slab and wave, repeated by each of the 5 column blocks of a row panel) and turns
// line-shaped loads into operand fragments through an LDS scratch.  On gfx950 a 16x16x32 MFMA holds the SIMD's vector
// issue for 8 of its 16 cycles, so two waves per SIMD have ~2 filler slots per MFMA: the split + scratch traffic
// saturates vector issue and the matrix pipe idles 30 % of the loop (profiles/r01_mfma_util.json).  Here the PRODUCER
// of an activation splits it once, in its epilogue, and stores the planes in the exact order the consumer's MFMA B
// operand wants them:
//
//   chunk(panel p, k-block kb, plane q) = 1 KiB at ((p * nkb + kb) * 3 + q) * 1024; lane L = 16 g + j holds 16 bytes
//   (8 bf16) at L * 16: batch row 16 p + j, slots 8 g .. 8 g + 7 of the 32-feature block kb.
//   Slot s = 8 g + u of a block holds feature f(s) = 16 (u >> 2) + 4 g + (u & 3) of that block -- the order in which
//   two neighbouring 16 x 16 accumulator tiles of the producing GEMM present a lane's 8 outputs (C^T layout: lane
//   (j, g), register r of tile t = feature 16 t + 4 g + r of row j).  The consumer's weight planes carry the same
//   permutation on their K axis (applied once at pack time), so a producer lane's registers ARE a consumer operand.
//
// A consumer wave loads its B operand with ONE coalesced 16-byte load per lane, plane and batch tile (1 KiB
// contiguous per instruction), straight into registers: no LDS scratch, no split, no conversion in the K loop.
// Weights go through the LDS ring exactly as in usf_linear_bf16x3.hip.  Arithmetic is unchanged: six
// v_mfma_f32_16x16x32_bf16 per fp32-equivalent product, a1 w1 + (a1 w2 + a2 w1) + (a1 w3 + a2 w2 + a3 w1), fp32
// accumulation, smallest terms first.
#include <stdlib.h>

#include "usf_common.h"

namespace usf {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define USF_F16_GUARD 65000.0f
#define USF_PL_LOAD_A(ptr) (*(ptr))
#define USF_PL_STORE_C(v, ptr) (*(ptr) = (v))
// (non-temporal loads / stores of the activation stream and a rotated K walk per column tile were measured in rounds 3 / 4 --
// within noise, profiles/r03_tuning_experiments.md section 2 -- and are gone)

// NPL = 3: bf16 planes, six products per fp32 product (24 significant bits per operand, fp32's exponent range);
// NPL = 2: fp16 planes, three products a1 w1 + (a1 w2 + a2 w1) (22 significant bits per operand; half the matrix
//          instructions -- the chip is power-bound on this instruction mix, tools/exp_mfma_peak.hip)
template <int NPL> struct Planes;
template <> struct Planes<3> {
  typedef bf16x8 vec;
  static __device__ __forceinline__ f32x4 mfma(vec a, vec b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ void split(const float (&x)[8], vec (&o)[3]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const __bf16 h = (__bf16)x[j];
      const float r = x[j] - (float)h;           // exact
      const __bf16 m = (__bf16)r;
      const float r2 = r - (float)m;             // exact
      o[0][j] = h; o[1][j] = m; o[2][j] = (__bf16)r2;
    }
  }
};
template <> struct Planes<2> {
  typedef f16x8 vec;
  static __device__ __forceinline__ f32x4 mfma(vec a, vec b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ void split(const float (&x)[8], vec (&o)[2]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const _Float16 h = (_Float16)x[j];
      const float r = x[j] - (float)h;           // exact while h is finite
      o[0][j] = h; o[1][j] = (_Float16)r;
    }
  }
};

// feature offset (0..31) held by slot s of a 32-feature block
__host__ __device__ __forceinline__ int plane_feature_of_slot(int s) { return 16 * ((s & 7) >> 2) + 4 * (s >> 3) + (s & 3); }

// ------------------------------------------------------------------------------------------------------------
// pack: fp32 row-major [M, ld] -> planes (optionally x / pre_div - pre_sub first, columns gathered through idx)
// ------------------------------------------------------------------------------------------------------------
// one wave per (panel, k-block): lane (j, g) produces its 8 slots of row 16 p + j
template <int NPL>
__global__ __launch_bounds__(256) void pack_planes_kernel(const float* __restrict__ src, int64_t ld, int M, int npanels,
                                                          int nkb, const int32_t* __restrict__ idx,
                                                          const float* __restrict__ pre_div,
                                                          const float* __restrict__ pre_sub, char* __restrict__ dst,
                                                          int32_t* __restrict__ range_flag) {
  typedef Planes<NPL> PT;
  const int lane = threadIdx.x & 63;
  const int lj = lane & 15, lg = lane >> 4;
  const int64_t chunk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (chunk >= (int64_t)npanels * nkb) return;
  const int p = (int)(chunk / nkb), kb = (int)(chunk % nkb);
  const int row = 16 * p + lj;
  float x[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int l = 32 * kb + 16 * (u >> 2) + 4 * lg + (u & 3);          // position in the buffer's logical layout
    const int c = idx[l];                                             // source column (-1: padding)
    float v = 0.f;
    if (c >= 0 && row < M) {
      v = src[(int64_t)row * ld + c];
      if (pre_div) v = v / pre_div[l];
      if (pre_sub) v = v - pre_sub[l];
    }
    x[u] = v;
  }
  if (NPL == 2 && range_flag) {                 // fp16 planes: a value outside fp16's range (or a NaN) voids the run
    bool bad = false;
#pragma unroll
    for (int u = 0; u < 8; ++u) bad = bad || !(fabsf(x[u]) < USF_F16_GUARD);
    if (bad) atomicOr(range_flag, 1);
  }
  typename PT::vec o[NPL];
  PT::split(x, o);
  char* out = dst + (chunk * NPL) * 1024 + lane * 16;
#pragma unroll
  for (int q = 0; q < NPL; ++q) *reinterpret_cast<typename PT::vec*>(out + q * 1024) = o[q];
}

// The same with whole rows read once, coalesced: one block per panel stages its 16 source rows in LDS (row stride src_cols + 1
// floats) and its four waves gather the panel's chunks from there.  The per-element gather above asks the memory system for
// 16 different rows per wave instruction and, with a checkerboard layout index, for every other float of a line: 2.6 TB/s of
// the 8 (profiles/r04_hbm_traffic.json); here the HBM sees 16-byte loads along rows and 1-KiB plane stores only.
// GRAD: the staged value is g = row_weight[m] * d/dz base_c(z) (base_grad_kernel's Laplace / Normal formulas) instead of the
// source value -- the training backward's head in one pass instead of usf_base_logprob_grad_f32 + usf_pack_planes_f32.
template <int NPL, bool GRAD>
__global__ __launch_bounds__(256) void pack_planes_rows_kernel(const float* __restrict__ src, int64_t ld, int M, int nkb, int src_cols,
                                                               const int32_t* __restrict__ idx, const float* __restrict__ pre_div,
                                                               const float* __restrict__ pre_sub, char* __restrict__ dst,
                                                               int32_t* __restrict__ range_flag, const float* __restrict__ row_weight,
                                                               const float* __restrict__ loc, const float* __restrict__ scale, int grad_base,
                                                               int vec) {
  typedef Planes<NPL> PT;
  extern __shared__ float tile[];                 // [16][S] rows, then the layout tables: idx [32 nkb] | pre_div | pre_sub
  const int S = src_cols + 1;
  const int L = 32 * nkb;
  int* const t_idx = reinterpret_cast<int*>(tile + 16 * S);
  float* const t_div = tile + 16 * S + L;
  float* const t_sub = t_div + L;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int p = blockIdx.x;
  // the layout tables once per block, coalesced (a chunk's 8 x 3 dependent global loads per lane were what the first version
  // of this kernel waited for: 203 us, the same as the per-element gather)
  for (int e = tid; e < L; e += 256) {
    t_idx[e] = idx[e];
    t_div[e] = pre_div ? pre_div[e] : 1.0f;
    t_sub[e] = pre_sub ? pre_sub[e] : 0.0f;
  }
  auto xform = [&](float v, int c, float w) {
    if (!GRAD) return v;
    const float t = v - loc[c];
    const float sg = (float)((t > 0.f) - (t < 0.f));
    const float g = (grad_base == 1 + USF_BASE_LAPLACE) ? -sg / scale[c] : -t / (scale[c] * scale[c]);
    return g * w;
  };
  if (vec) {
    const int c4n = src_cols >> 2;
    for (int e = tid; e < 16 * c4n; e += 256) {
      const int r = e / c4n, c = 4 * (e - r * c4n);
      const int row = 16 * p + r;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      float w = 0.f;
      if (row < M) {
        v = *reinterpret_cast<const f32x4*>(src + (int64_t)row * ld + c);
        if (GRAD) w = row_weight[row];
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) tile[r * S + c + q] = (row < M) ? xform(v[q], c + q, w) : 0.f;
    }
    for (int e = tid; e < 16 * (src_cols & 3); e += 256) {      // (a column remainder: scalar)
      const int r = e / (src_cols & 3), c = (src_cols & ~3) + e % (src_cols & 3);
      const int row = 16 * p + r;
      tile[r * S + c] = (row < M) ? xform(src[(int64_t)row * ld + c], c, GRAD ? row_weight[row] : 0.f) : 0.f;
    }
  } else {
    for (int e = tid; e < 16 * src_cols; e += 256) {
      const int r = e / src_cols, c = e - r * src_cols;
      const int row = 16 * p + r;
      tile[r * S + c] = (row < M) ? xform(src[(int64_t)row * ld + c], c, GRAD ? row_weight[row] : 0.f) : 0.f;
    }
  }
  __syncthreads();
  const int lj = lane & 15, lg = lane >> 4;
  const bool row_live = 16 * p + lj < M;
  const bool has_div = pre_div != nullptr, has_sub = pre_sub != nullptr;
  bool bad = false;
  for (int kb = wave; kb < nkb; kb += 4) {
    float x[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int l = 32 * kb + 16 * (u >> 2) + 4 * lg + (u & 3);
      const int c = t_idx[l];
      float v = 0.f;
      if (c >= 0 && row_live) {                       // (rows beyond M hold zeros in the tile and stay zero)
        v = tile[lj * S + c];
        if (has_div) v = v / t_div[l];
        if (has_sub) v = v - t_sub[l];
      }
      x[u] = v;
    }
    if (NPL == 2) {
#pragma unroll
      for (int u = 0; u < 8; ++u) bad = bad || !(fabsf(x[u]) < USF_F16_GUARD);
    }
    typename PT::vec o[NPL];
    PT::split(x, o);
    char* out = dst + ((int64_t)((int64_t)p * nkb + kb) * NPL) * 1024 + lane * 16;
#pragma unroll
    for (int q = 0; q < NPL; ++q) *reinterpret_cast<typename PT::vec*>(out + q * 1024) = o[q];
  }
  if (NPL == 2 && range_flag && bad) atomicOr(range_flag, 1);
}

int pack_planes(const usf_pack_planes_desc* d, hipStream_t stream) {
  if (!d) { set_error("usf_pack_planes_f32: null descriptor"); return -1; }
  if (d->M < 0 || d->nkb <= 0 || d->M > 0x7fffffff || d->nkb > 0x7fffff || d->ld < 1) { set_error("usf_pack_planes_f32: bad sizes"); return -2; }
  if (d->M == 0) return 0;
  if (!d->src || !d->idx || !d->planes) { set_error("usf_pack_planes_f32: null pointer"); return -1; }
  if (!aligned16(d->planes)) { set_error("usf_pack_planes_f32: planes must be 16-byte aligned"); return -2; }
  if (d->format != USF_PLANES_BF16X3 && d->format != USF_PLANES_F16X2) { set_error("usf_pack_planes_f32: unknown format %d", d->format); return -2; }
  const int64_t npanels = (d->M + 15) / 16;
  const int64_t chunks = npanels * d->nkb;
  const int64_t blocks = (chunks + 3) / 4;
  if (blocks > 0x7fffffffLL) { set_error("usf_pack_planes_f32: grid too large"); return -3; }
  if (d->grad_base != 0) {
    if (d->src_cols <= 0 || !d->row_weight || !d->loc || !d->scale || d->format != USF_PLANES_BF16X3 ||
        (d->grad_base != 1 + USF_BASE_LAPLACE && d->grad_base != 1 + USF_BASE_NORMAL)) {
      set_error("usf_pack_planes_f32: grad_base needs src_cols, row_weight, loc, scale, the bf16x3 format and a Laplace / Normal base");
      return -2;
    }
  }
  const int rows_env = (int)tuning("pack_rows", 1);        // tuning aid: 0 = the per-element gather
  const size_t lds = ((size_t)16 * (size_t)(d->src_cols + 1) + (size_t)3 * 32 * (size_t)d->nkb) * sizeof(float);
  if (d->src_cols > 0 && d->src_cols <= d->ld && lds <= 65536 && (rows_env || d->grad_base != 0)) {
    const int vec = (aligned16(d->src) && (d->ld & 3) == 0) ? 1 : 0;
    char* out = reinterpret_cast<char*>(d->planes);
#define USF_PPR(NPL_, GRAD_) hipLaunchKernelGGL((pack_planes_rows_kernel<NPL_, GRAD_>), dim3((unsigned)npanels), dim3(256), lds, stream, d->src, \
                             d->ld, (int)d->M, (int)d->nkb, (int)d->src_cols, d->idx, d->pre_div, d->pre_sub, out, d->range_flag,      \
                             d->row_weight, d->loc, d->scale, (int)d->grad_base, vec)
    if (d->grad_base != 0) USF_PPR(3, true);
    else if (d->format == USF_PLANES_F16X2) USF_PPR(2, false);
    else USF_PPR(3, false);
#undef USF_PPR
    return check_launch("usf_pack_planes_f32");
  }
  if (d->grad_base != 0) { set_error("usf_pack_planes_f32: grad_base needs 16 (src_cols + 1) floats to fit 64 KB of LDS"); return -2; }
  if (d->format == USF_PLANES_F16X2)
    hipLaunchKernelGGL(pack_planes_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, stream, d->src, d->ld, (int)d->M, (int)npanels,
                       (int)d->nkb, d->idx, d->pre_div, d->pre_sub, reinterpret_cast<char*>(d->planes), d->range_flag);
  else
    hipLaunchKernelGGL(pack_planes_kernel<3>, dim3((unsigned)blocks), dim3(256), 0, stream, d->src, d->ld, (int)d->M, (int)npanels,
                       (int)d->nkb, d->idx, d->pre_div, d->pre_sub, reinterpret_cast<char*>(d->planes), d->range_flag);
  return check_launch("usf_pack_planes_f32");
}

// ------------------------------------------------------------------------------------------------------------
// GEMM on planes
// ------------------------------------------------------------------------------------------------------------
struct PlArgs {
  const char* A; const char* Wp; const float* bias; const float* post_mul;
  const char* R; char* Cp; float* Cf;
  int64_t ldc, plane_stride;
  int ldwp, wrows;
  int M, npanels;
  int a_nkb, a_kb0, nk;
  int c_nkb, c_kb0, c_kbn;
  int N, nbm, nbn, nvb;
  float res_sign, slope; int act;
  int32_t* range_flag;
  // base density in the fp32-output epilogue (b_part == nullptr: off): tables loc | 1 / scale | constant, stride b_stride
  const float* b_tab; float* b_part; int b_stride; int b_base;
  unsigned long long* clk;              // usf_set_clock_buffer: [shader cycles, 100 MHz ticks] summed over the blocks' lifetimes
  unsigned long long* dbg;
  unsigned long long* span;             // tuning builds only: [first wave start, last wave end] in s_memrealtime ticks
};

template <int NPL, int TN, bool F32OUT>
__global__ __launch_bounds__(512, 2) void gemm_planes_kernel(const PlArgs p) {
  typedef Planes<NPL> PT;
  typedef typename PT::vec vec8;
  constexpr int NT = 512;
  constexpr int BN = TN * 32;
  constexpr int FT = 2 * TN;                    // 16-feature tiles per wave, each against 2 batch tiles of 16 rows
  constexpr unsigned CHB = NPL * 1024u;         // bytes of one (panel, k-block) chunk group
  // weight stage: NPL planes x 4 k-chunks x BN rows of 16-byte slots, image slot(plane, chunk c, row r) =
  // (plane * 4 + c) * BN + (r ^ 2c)  (conflict-free for the fragment reads and for the row-major deal; see
  // usf_linear_bf16x3.hip)
  constexpr int CS = BN;
  constexpr int NSLOT = NPL * 4 * BN;
  constexpr int NWV = (NSLOT + NT - 1) / NT;
  constexpr int STG = NPL * 4 * CS * 4;         // floats per staging buffer
  constexpr int NB = 3;                         // ring of three weight buffers (see the K loop)
  static_assert(NWV * NT - NSLOT <= NSLOT, "surplus threads wrap once");
  __shared__ __attribute__((aligned(16))) float wring[NB * STG];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int lj = lane & 15, lg = lane >> 4;
#ifdef USF_STAMP
  if (p.span && tid == 0) atomicMin(p.span, __builtin_amdgcn_s_memrealtime());
#endif
  unsigned long long clk_c0 = 0, clk_r0 = 0;    // (scalar registers; block-uniform branch)
#ifndef USF_NO_CLOCK                             // (A/B builds: the kernel without the two counter reads)
  if (p.clk) { clk_c0 = __builtin_amdgcn_s_memtime(); clk_r0 = __builtin_amdgcn_s_memrealtime(); }
#endif

  // Persistent blocks: the grid is one block per CU (a multiple of 8); block b works through the virtual blocks
  // b, b + grid, b + 2 grid, ... -- the order in which the hardware would have dispatched a grid of that size.
  // XCD-aware map of a virtual block: the column blocks of one 256-row panel group run back to back on one XCD.
  for (int bid = blockIdx.x; bid < p.nvb; bid += gridDim.x) {
  const int xcd = bid & 7;
  const int seq = bid >> 3;
  const int pg = (seq / p.nbn) * 8 + xcd;       // group of 16 panels (256 rows)
  const int bn = seq % p.nbn;
  if (pg >= p.nbm) continue;
  if (bid != (int)blockIdx.x) __syncthreads();  // the previous tile's last reads of the weight ring are done
  const int n0 = bn * BN;
  int pw[2];                                    // this wave's two panels (unclamped: >= npanels means "no rows")
  pw[0] = pg * 16 + wave * 2;
  pw[1] = pw[0] + 1;

  // ---- activations: one 16-byte load per lane, plane and batch tile, straight into the MFMA B operand ----
  unsigned aoff[2];
#pragma unroll
  for (int b = 0; b < 2; ++b)
    aoff[b] = ((unsigned)min(pw[b], p.npanels - 1) * (unsigned)p.a_nkb + (unsigned)p.a_kb0) * CHB + (unsigned)lane * 16u;
  auto issue_a = [&](int kb, vec8 (&dst)[2][NPL]) {
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int q = 0; q < NPL; ++q)
        dst[b][q] = USF_PL_LOAD_A(reinterpret_cast<const vec8*>(p.A + (aoff[b] + (unsigned)kb * CHB + (unsigned)q * 1024u)));
  };

  // ---- weights: register-staged into the LDS ring (2-byte elements; offsets in bytes) ----
  unsigned wsrc[NWV];
  int wdst[NWV];
#pragma unroll
  for (int i = 0; i < NWV; ++i) {
    const int idx = tid + NT * i;
    const int idc = (idx < NSLOT) ? idx : idx - NSLOT;
    const int pl = idc / (4 * BN), rem = idc % (4 * BN);
    const int r = rem >> 2, ch = rem & 3;
    wsrc[i] = (unsigned)(2 * (pl * p.plane_stride + (int64_t)min(n0 + r, p.wrows - 1) * p.ldwp + 8 * ch));
    wdst[i] = 4 * ((pl * 4 + ch) * CS + (r ^ (2 * ch)));
  }
  auto issue_w = [&](int k0, f32x4 (&dst)[NWV]) {
#pragma unroll
    for (int i = 0; i < NWV; ++i) dst[i] = *reinterpret_cast<const f32x4*>(p.Wp + (wsrc[i] + 2u * (unsigned)k0));
  };
  auto store_w = [&](float* wb, const f32x4 (&src)[NWV]) {
#pragma unroll
    for (int i = 0; i < NWV; ++i) *reinterpret_cast<f32x4*>(wb + wdst[i]) = src[i];
  };

  // accumulators (C^T: batch row on the lane, 4 consecutive output features per register group) start at the bias
  f32x4 acc[FT][2];
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ft = 0; ft < FT; ++ft) {
    f32x4 bv = zero4;
    if (p.bias) bv = *reinterpret_cast<const f32x4*>(p.bias + min(n0 + ft * 16 + 4 * lg, p.wrows - 4));
    acc[ft][0] = bv;
    acc[ft][1] = bv;
  }

#ifdef USF_STAMP
#define PSTAMP(v) unsigned long long v = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0)
#else
#define PSTAMP(v)
#endif
  PSTAMP(t0);
#ifdef USF_STAMP
  const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
#endif
  vec8 pa[2][NPL], pb[2][NPL];
  vec8 fr[2][NPL], f2[2][NPL];                   // weight fragments: current tile pair / the pair being read
  f32x4 wst[NWV];
  const int nslab = p.nk;
  auto ks = [&](int s_) { return s_; };
  issue_w(ks(0) * 32, wst);
  store_w(wring, wst);
  issue_a(ks(0), pa);
  issue_w(ks(min(1, nslab - 1)) * 32, wst);     // staged at the middle of slab 0
  __syncthreads();
  const float* const wl0 = wring + 4 * (lg * CS + (lj ^ (2 * lg)));
  auto read_pair = [&](const float* wl, int pr, vec8 (&f)[2][NPL]) {
#pragma unroll
    for (int q = 0; q < NPL; ++q) {
      f[0][q] = *reinterpret_cast<const vec8*>(wl + 4 * (q * 4 * CS + (2 * pr) * 16));
      f[1][q] = *reinterpret_cast<const vec8*>(wl + 4 * (q * 4 * CS + (2 * pr + 1) * 16));
    }
  };
  read_pair(wl0, 0, fr);

  // One slab = TN tile pairs, NPR products per feature tile and batch tile (NPL = 3: six, smallest terms first;
  // NPL = 2: three).  Software pipeline: the weight fragments of pair i + 1 are read from LDS under the MFMAs of pair
  // i -- ACROSS the slab boundary too (the last pair of slab s reads the first pair of slab s + 1), so the matrix pipe
  // never waits for an LDS round trip at a slab start.  The block's ONE barrier per slab sits in the MIDDLE of the
  // slab, right behind the stores that stage slab s + 1 into the ring: the waves cross it with their next fragments
  // already in registers and resume with MFMAs.  Ring of three buffers: slab s + 1 is written (mid-slab s) into the
  // buffer slab s - 2 was read from, and every wave has passed barrier s - 1, i.e. has finished slab s - 2.
  constexpr int NPR = (NPL == 3) ? 6 : 3;
  constexpr int NP = TN;                        // tile pairs per slab
  constexpr int MID = (NP - 1) / 2;             // the barrier follows pair MID
  constexpr int NLD = 2 * NPL;                  // operand loads per slab and thread
  constexpr int LPP = (NLD + MID) / (MID + 1);  // ... dealt over the pairs in front of the barrier
#define USF_MM(FT_, W, P)                                        \
  acc[FT_][0] = PT::mfma(W, cur[0][P], acc[FT_][0]);             \
  acc[FT_][1] = PT::mfma(W, cur[1][P], acc[FT_][1])
  auto mm_pair = [&](int pr, const vec8 (&f)[2][NPL], const vec8 (&cur)[2][NPL]) {
    const int ft = 2 * pr;
    if (NPL == 3) {
      USF_MM(ft, f[0][2], 0); USF_MM(ft + 1, f[1][2], 0); USF_MM(ft, f[0][1], 1); USF_MM(ft + 1, f[1][1], 1);
      USF_MM(ft, f[0][0], NPL - 1); USF_MM(ft + 1, f[1][0], NPL - 1); USF_MM(ft, f[0][1], 0); USF_MM(ft + 1, f[1][1], 0);
      USF_MM(ft, f[0][0], 1); USF_MM(ft + 1, f[1][0], 1); USF_MM(ft, f[0][0], 0); USF_MM(ft + 1, f[1][0], 0);
    } else {
      USF_MM(ft, f[0][1], 0); USF_MM(ft + 1, f[1][1], 0); USF_MM(ft, f[0][0], 1); USF_MM(ft + 1, f[1][0], 1);
      USF_MM(ft, f[0][0], 0); USF_MM(ft + 1, f[1][0], 0);
    }
  };
#if defined(USF_STAMP) && USF_STAMP >= 2     // in-loop phase stamps (they drain the LDS queue: a diagnostic build of its own)
  unsigned long long ph[4] = {0, 0, 0, 0};
#define LSTAMP(v) PSTAMP(v)
#define LACC(i, a, b) ph[i] += (b) - (a)
#else
#define LSTAMP(v)
#define LACC(i, a, b)
#endif
  // fA holds the slab's first pair on entry; the two fragment sets alternate as "current" / "being read"; the first
  // pair of the NEXT slab lands in fA when NP is even, in fB when NP is odd (the caller swaps them then)
  auto slab = [&](int s, int ring_s, const vec8 (&cur)[2][NPL], vec8 (&nxt)[2][NPL], vec8 (&fA)[2][NPL], vec8 (&fB)[2][NPL]) {
    LSTAMP(l0);
    const int ring_n = (ring_s == 2) ? 0 : ring_s + 1;
    const float* wl = wl0 + ring_s * STG;
    const float* wln = wl0 + ring_n * STG;
    float* wb = wring + ring_n * STG;
    issue_a(ks(min(s + 1, nslab - 1)), nxt);
    // ---- pairs in front of the barrier; the last of them carries the stores that stage slab s + 1 ----
#pragma unroll
    for (int pr = 0; pr <= MID; ++pr) {
      if (pr & 1) { if (pr + 1 < NP) read_pair(wl, pr + 1, fA); mm_pair(pr, fB, cur); }
      else        { if (pr + 1 < NP) read_pair(wl, pr + 1, fB); mm_pair(pr, fA, cur); }
      if (pr == MID) store_w(wb, wst);
      // pins (masks: 0x008 MFMA, 0x020 VMEM read, 0x100 DS read, 0x200 DS write): the fragment reads first, then the
      // pair's MFMAs with this pair's share of the global loads / the staging stores dealt in
      __builtin_amdgcn_sched_group_barrier(0x100, 2 * NPL, 0);
      if (pr < MID) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * NPR, 0);
        if (pr * LPP < NLD) __builtin_amdgcn_sched_group_barrier(0x020, LPP, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * NPR, 0);
      } else {
        constexpr int MPS = (4 * NPR) / (NWV + 1);        // MFMAs between two staging stores
#pragma unroll
        for (int i = 0; i < NWV; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, MPS, 0);
          __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * NPR - NWV * MPS, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    LSTAMP(l1);
    LSTAMP(l2);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    LSTAMP(l3);
    // ---- pairs behind the barrier; the first carries the global loads of the weights of slab s + 2 (the staging
    // registers are free again: a full slab of latency cover until the middle of slab s + 1 stores them); the last
    // one reads the first pair of the NEXT slab (staged just now) ----
    issue_w(ks(min(s + 2, nslab - 1)) * 32, wst);
#pragma unroll
    for (int pr = MID + 1; pr < NP; ++pr) {
      if (pr & 1) { if (pr + 1 < NP) read_pair(wl, pr + 1, fA); else read_pair(wln, 0, fA); mm_pair(pr, fB, cur); }
      else        { if (pr + 1 < NP) read_pair(wl, pr + 1, fB); else read_pair(wln, 0, fB); mm_pair(pr, fA, cur); }
      __builtin_amdgcn_sched_group_barrier(0x100, 2 * NPL, 0);
      if (pr == MID + 1) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * NPR, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, NWV, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * NPR, 0);
      } else {
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * NPR, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    LSTAMP(l4);
    LACC(0, l0, l1); LACC(1, l1, l2); LACC(2, l2, l3); LACC(3, l3, l4);
  };
  PSTAMP(t1);
  int s = 0, ring = 0;
  for (; s + 2 <= nslab; s += 2) {
    slab(s, ring, pa, pb, fr, f2);
    ring = (ring == 2) ? 0 : ring + 1;
    if (NP & 1) slab(s + 1, ring, pb, pa, f2, fr); else slab(s + 1, ring, pb, pa, fr, f2);
    ring = (ring == 2) ? 0 : ring + 1;
  }
  if (s < nslab) slab(s, ring, pa, pb, fr, f2);
#undef USF_MM
  PSTAMP(t2);

  // ---- epilogue ----
  bool bad = false;                            // fp16 planes only: range guard (see usf_gemm_planes_desc.range_flag)
  if (p.act == USF_ACT_LEAKY_RELU) {           // wave-uniform branch: the affine layers (no activation) skip the selects
#pragma unroll
    for (int ft = 0; ft < FT; ++ft)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[ft][b][j] = act_apply(acc[ft][b][j], USF_ACT_LEAKY_RELU, p.slope);
  }
  if (F32OUT) {
    // fp32 row-major: lane (j, g) of tile (ft, b) holds features n0 + 16 ft + 4 g + (0..3) of row 16 pw[b] + j
    const bool vec_ok = ((p.ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.Cf) & 15u) == 0);
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int row = 16 * pw[b] + lj;
      float bsum = 0.f;                          // this lane's share of sum_d log p(z[row, d]) over the block's columns
#pragma unroll
      for (int ft = 0; ft < FT; ++ft) {
        const int col = n0 + 16 * ft + 4 * lg;
        f32x4 v = acc[ft][b];
        if (p.post_mul) v = v * *reinterpret_cast<const f32x4*>(p.post_mul + min(col, p.wrows - 4));
        if (p.b_part) {
          // the flow's last layer: the base density of the row (usf_base_logprob_f32's terms, 1 / scale from the table) is
          // reduced here instead of by a pass over the stored rows
          const float* t = p.b_tab + min(col, p.b_stride - 4);
          const f32x4 bl = *reinterpret_cast<const f32x4*>(t);
          const f32x4 bi = *reinterpret_cast<const f32x4*>(t + p.b_stride);
          const f32x4 bc = *reinterpret_cast<const f32x4*>(t + 2 * p.b_stride);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float d = (v[j] - bl[j]) * bi[j];
            const float term = (p.b_base == USF_BASE_LAPLACE) ? bc[j] - fabsf(d) : bc[j] - 0.5f * d * d;
            bsum += (col + j < p.N) ? term : 0.f;
          }
        }
        if (NPL == 2 && row < p.M && col < p.N) {
          // (an overflowed weight or activation plane upstream shows up here as inf / NaN)
#pragma unroll
          for (int j = 0; j < 4; ++j) bad = bad || (col + j < p.N && !(fabsf(v[j]) < 3.0e38f));
        }
        if (p.Cf != nullptr && row < p.M && col < p.N) {      // (Cf == nullptr: only the base density is wanted of this layer)
          float* dst = p.Cf + (int64_t)row * p.ldc + col;
          if (vec_ok && col + 3 < p.N) {
            *reinterpret_cast<f32x4*>(dst) = v;
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (col + j < p.N) dst[j] = v[j];
          }
        }
      }
      if (p.b_part) {
        // the four lanes (j, g = 0..3) of a row hold its 16-column tiles' quarters: one partial sum per (row, column block)
        bsum += __shfl_xor(bsum, 16);
        bsum += __shfl_xor(bsum, 32);
        if (lg == 0 && row < p.M) p.b_part[(int64_t)row * 8 + bn] = bsum;
      }
    }
  } else {
    // planes: two neighbouring tiles give a lane the 8 slots of its chunk line; activation, residual (read back
    // from planes: the sum of the planes is the stored fp32 value), split, NPL 16-byte stores -- all lane-local
#pragma unroll
    for (int t = 0; t < TN; ++t) {
      const int kbo = bn * TN + t;
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        if (kbo < p.c_kbn && pw[b] < p.npanels) {
          const size_t off = (((size_t)pw[b] * p.c_nkb + (p.c_kb0 + kbo)) * NPL) * 1024 + (size_t)lane * 16;
          float x[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) x[u] = acc[2 * t + (u >> 2)][b][u & 3];
          if (p.R) {
            vec8 r[NPL];
#pragma unroll
            for (int q = 0; q < NPL; ++q) r[q] = *reinterpret_cast<const vec8*>(p.R + off + q * 1024);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              float rv = (float)r[0][u] + (float)r[1][u];
              if (NPL == 3) rv = rv + (float)r[NPL - 1][u];
              x[u] = rv + p.res_sign * x[u];
            }
          }
          if (NPL == 2) {
#pragma unroll
            for (int u = 0; u < 8; ++u) bad = bad || (16 * pw[b] + lj < p.M && !(fabsf(x[u]) < USF_F16_GUARD));
          }
          vec8 o[NPL];
          PT::split(x, o);
#ifdef USF_NOSTORE                          // tuning aid: how much of the kernel is the output stream?
          if (o[0][0] == (decltype(o[0][0] + o[0][0]))1234.5f && o[1][1] == (decltype(o[0][0] + o[0][0]))77.f)
#endif
          {
#pragma unroll
            for (int q = 0; q < NPL; ++q) USF_PL_STORE_C(o[q], reinterpret_cast<vec8*>(p.Cp + off + q * 1024));
          }
        }
      }
    }
  }
  if (NPL == 2 && p.range_flag && bad) atomicOr(p.range_flag, 1);
#ifdef USF_STAMP
  PSTAMP(t3);
  if (p.dbg && lane == 0) {
    unsigned long long* o = p.dbg + (size_t)((bid % 2048) * 8 + wave) * 8;
    o[0] = t1 - t0; o[1] = t2 - t1; o[2] = t3 - t2; o[3] = t3 - t0; o[4] = 1; o[5] = __builtin_amdgcn_s_memrealtime() - rt0;
    o[6] = rt0; o[7] = __builtin_amdgcn_s_memrealtime();
#if USF_STAMP >= 2
    unsigned long long* o2 = p.dbg + 16384 * 8 + (size_t)((bid % 2048) * 8 + wave) * 8;
    for (int i = 0; i < 4; ++i) o2[i] = ph[i];
#endif
  }
#endif
  }  // virtual blocks
#ifdef USF_STAMP
  if (p.span && tid == 0) atomicMax(p.span + 1, __builtin_amdgcn_s_memrealtime());
#endif
#ifndef USF_NO_CLOCK
  if (p.clk && tid == 0) {
    atomicAdd(p.clk, __builtin_amdgcn_s_memtime() - clk_c0);
    atomicAdd(p.clk + 1, __builtin_amdgcn_s_memrealtime() - clk_r0);
  }
#endif
}

#ifdef USF_STAMP
unsigned long long* g_pdbg = nullptr;
unsigned long long* g_pspan = nullptr;
#endif

template <int NPL, int TN>
static int launch_planes(PlArgs a, bool f32out, hipStream_t stream) {
  constexpr int BN = TN * 32;
  a.nbm = (a.npanels + 15) / 16;
  a.nbn = f32out ? (a.N + BN - 1) / BN : (a.c_kbn + TN - 1) / TN;
  int64_t grid = (((int64_t)a.nbm + 7) / 8) * 8 * a.nbn;
  if (grid > 0x7fffffffLL) { set_error("usf_gemm_planes: grid too large"); return -3; }
  a.nvb = (int)grid;
  const int persist = (int)tuning("planes_persist", 1);    // tuning aid: 0 = one block per tile
  int cus = device_cu_count();
  cus = (cus / 8) * 8 > 0 ? (cus / 8) * 8 : 8;
  if (persist && grid > cus) grid = cus;
  if (f32out) hipLaunchKernelGGL((gemm_planes_kernel<NPL, TN, true>), dim3((unsigned)grid), dim3(512), 0, stream, a);
  else hipLaunchKernelGGL((gemm_planes_kernel<NPL, TN, false>), dim3((unsigned)grid), dim3(512), 0, stream, a);
  return check_launch("usf_gemm_planes");
}

// column-block width (in 32-feature blocks) for an output of nblk blocks: the one that pads less, 5 on a tie
int gemm_planes_tn(int64_t nblk) {
  const int64_t pad5 = (nblk + 4) / 5 * 5 - nblk, pad4 = (nblk + 3) / 4 * 4 - nblk;
  return pad4 < pad5 ? 4 : 5;
}

// which instantiation usf_gemm_planes_bf16x3 launches for this descriptor: 5000 + 10 TN + (1: fp32 output, 0: planes)
int gemm_planes_variant(const usf_gemm_planes_desc* d) {
  if (!d || d->M <= 0) return 0;
  const bool f32out = d->C_f32 != nullptr || d->base_part != nullptr;
  const int64_t nblk = f32out ? (d->N + 31) / 32 : d->c_kbn;
  if (nblk <= 0) return 0;
  return 5000 + 10 * gemm_planes_tn(nblk) + (f32out ? 1 : 0);
}

int gemm_planes(const usf_gemm_planes_desc* d, hipStream_t stream) {
  if (!d) { set_error("usf_gemm_planes_bf16x3: null descriptor"); return -1; }
  if (d->M < 0 || d->M > 0x7fffffff || d->nk <= 0 || d->a_nkb <= 0 || d->a_kb0 < 0 || d->a_kb0 + d->nk > d->a_nkb ||
      d->w_rows < 4 || d->w_rows > 0x7fffffff) {
    set_error("usf_gemm_planes_bf16x3: bad sizes (M=%lld nk=%lld a_nkb=%lld a_kb0=%lld w_rows=%lld)", (long long)d->M,
              (long long)d->nk, (long long)d->a_nkb, (long long)d->a_kb0, (long long)d->w_rows);
    return -2;
  }
  if (d->M == 0) return 0;
  const bool f32out = d->C_f32 != nullptr || d->base_part != nullptr;
  if (!d->A || !d->W_planes || (!f32out && !d->C_planes)) { set_error("usf_gemm_planes_bf16x3: null A / W / C"); return -1; }
  if (f32out && d->C_planes) { set_error("usf_gemm_planes_bf16x3: give C_planes or C_f32, not both"); return -2; }
  if (!aligned16(d->A) || !aligned16(d->W_planes) || (d->C_planes && !aligned16(d->C_planes)) ||
      (d->residual && !aligned16(d->residual)) || (d->bias && !aligned16(d->bias)) || (d->post_mul && !aligned16(d->post_mul))) {
    set_error("usf_gemm_planes_bf16x3: pointers must be 16-byte aligned");
    return -2;
  }
  if ((d->ldw & 7) || d->ldw < 32 * d->nk || (d->w_rows & 3)) { set_error("usf_gemm_planes_bf16x3: ldw must be a multiple of 8 and >= 32 nk; w_rows a multiple of 4"); return -2; }
  if (d->act != USF_ACT_NONE && d->act != USF_ACT_LEAKY_RELU) { set_error("usf_gemm_planes_bf16x3: bad act"); return -2; }
  const int64_t npanels = (d->M + 15) / 16;
  if (d->format != USF_PLANES_BF16X3 && d->format != USF_PLANES_F16X2) { set_error("usf_gemm_planes_bf16x3: unknown format %d", d->format); return -2; }
  const int64_t npl = d->format == USF_PLANES_F16X2 ? 2 : 3;
  if (npanels * d->a_nkb * npl * 1024 >= (1LL << 32) || 2 * npl * d->w_plane_stride >= (1LL << 32)) {
    set_error("usf_gemm_planes_bf16x3: operand larger than the kernel's 32-bit offsets");
    return -3;
  }
  PlArgs a;
  a.A = reinterpret_cast<const char*>(d->A); a.Wp = reinterpret_cast<const char*>(d->W_planes);
  a.bias = d->bias; a.post_mul = d->post_mul; a.R = reinterpret_cast<const char*>(d->residual);
  a.Cp = reinterpret_cast<char*>(d->C_planes); a.Cf = d->C_f32; a.ldc = d->ldc; a.plane_stride = d->w_plane_stride;
  a.ldwp = (int)d->ldw; a.wrows = (int)d->w_rows; a.M = (int)d->M; a.npanels = (int)npanels;
  a.a_nkb = (int)d->a_nkb; a.a_kb0 = (int)d->a_kb0; a.nk = (int)d->nk;
  a.c_nkb = (int)d->c_nkb; a.c_kb0 = (int)d->c_kb0; a.c_kbn = (int)d->c_kbn; a.N = (int)d->N;
  a.res_sign = d->res_sign; a.slope = d->slope; a.act = d->act; a.nbm = a.nbn = a.nvb = 0;
  a.range_flag = d->range_flag;
  a.b_tab = d->base_tab; a.b_part = d->base_part; a.b_stride = (int)d->base_tab_stride; a.b_base = d->base;
  a.dbg = nullptr; a.span = nullptr;
  a.clk = clock_buffer();
#ifdef USF_STAMP
  a.dbg = g_pdbg; a.span = g_pspan;
  if (g_pspan) g_pspan += 2;            // one [start, end] pair per launch
#endif
  int64_t nblk;
  if (f32out) {
    if (d->N <= 0 || d->N > d->w_rows || d->ldc < d->N) { set_error("usf_gemm_planes_bf16x3: bad N / ldc for fp32 output"); return -2; }
    if (d->residual) { set_error("usf_gemm_planes_bf16x3: residual needs planes output"); return -2; }
    nblk = (d->N + 31) / 32;
    if (d->base_part) {
      if (!d->base_tab || !aligned16(d->base_tab) || (d->base_tab_stride & 3) || d->base_tab_stride < ((d->N + 3) & ~(int64_t)3) ||
          d->base_tab_stride > 0x7fffffff || (d->base != USF_BASE_LAPLACE && d->base != USF_BASE_NORMAL)) {
        set_error("usf_gemm_planes_bf16x3: base density in the epilogue needs the tables of usf_base_tables_f32 (16-byte aligned, "
                  "stride a multiple of 4 and >= N) and base = USF_BASE_LAPLACE / USF_BASE_NORMAL");
        return -2;
      }
    }
  } else {
    if (d->base_part) { set_error("usf_gemm_planes_bf16x3: base_part needs the fp32-output form"); return -2; }
    if (d->c_kbn <= 0 || d->c_kb0 < 0 || d->c_kb0 + d->c_kbn > d->c_nkb || d->c_kbn * 32 > d->w_rows) {
      set_error("usf_gemm_planes_bf16x3: bad output block range (c_kb0=%lld c_kbn=%lld c_nkb=%lld w_rows=%lld)",
                (long long)d->c_kb0, (long long)d->c_kbn, (long long)d->c_nkb, (long long)d->w_rows);
      return -2;
    }
    if (npanels * d->c_nkb * npl * 1024 >= (1LL << 40)) { set_error("usf_gemm_planes_bf16x3: output too large"); return -3; }
    nblk = d->c_kbn;
  }
  const int force = (int)tuning("planes_tn", 0);
  const int tn = (force == 4 || force == 5) ? force : gemm_planes_tn(nblk);
  if (d->base_part && (d->N + 32 * tn - 1) / (32 * tn) > 8) {
    set_error("usf_gemm_planes_bf16x3: base_part holds 8 column blocks per row (N <= 1024)");
    return -2;
  }
  if (npl == 2) return tn == 4 ? launch_planes<2, 4>(a, f32out, stream) : launch_planes<2, 5>(a, f32out, stream);
  return tn == 4 ? launch_planes<3, 4>(a, f32out, stream) : launch_planes<3, 5>(a, f32out, stream);
}

}  // namespace usf
