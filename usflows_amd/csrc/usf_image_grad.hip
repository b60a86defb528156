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
#include <stdlib.h>
#include <string.h>

namespace usf {

// ------------------------------------------------------------------------------------------
// Weight gradient.  One wave owns one sample at a time and ALL (cout tile, cin tile, tap) accumulator tiles, so the sample
// loop needs no barrier: the wave stages its sample's x and dy into a private LDS image [channel][padded position] (fp32; row
// stride S = W + 1 and one zero row above / below for kernel 3, so a tap is a constant offset and the zero padding is
// real zeros), then walks the positions four at a time: A = dy[16 co][4 q], B = x[16 ci][4 (q + tap)] -> D[co][ci].
// The channel stride CS is 2 * odd: the 32 lanes of a ds_read_b32 group (16 channels x 2 positions) hit 32 banks.
// Staging is branch-free: element -> LDS index and the input mask come from small tables the block builds once in LDS
// (consecutive lanes take consecutive elements: coalesced loads, conflict-free stores); the nonlinearity is a select
// (identity = leaky slope 1).  The next sample's global loads are in flight during the matrix work (registers), the
// fragments of chunk ch + 1 are read between the MFMAs of chunk ch (sched_group_barrier), 1 wave per SIMD on the
// 512-register budget.  MFMA-issue bound: 2 * B * HW * cin * cout * taps flops at the exact-fp32 matrix rate
// (157 TFLOP/s), padded positions included: (H - 1) * S + W of H * W (MNIST 7 x 7: 56 / 49).
// ------------------------------------------------------------------------------------------
struct WgArgs {
  const float* x;
  const float* dy;
  float* part;             // [waves][nacc]
  const float* in_mul;     // [cin * HW] or NULL
  const float* pre_sub;    // [cin] or NULL
  int B, cin, cout, HW, W, S, base, CS, nch, q0, nex, ney, nacc;
  int rsplit;                // kernel 3: waves per sample (1, 2 or 4): each takes a range of the position chunks (small batches)
  unsigned m_hw, m_w;      // ceil(2^32 / HW), ceil(2^32 / W)  (HW, W >= 2)
  float slope_eff;         // leaky slope of the input nonlinearity; 1 = none
  int toff[9];
  int tab_floats;          // LDS floats in front of the waves' images: index tables (u16), mask, pre_sub
};

// (bx of nblk: the block's place in ITS launch -- blockIdx.x of gridDim.x, or its place among the blocks of its job when many
// weight gradients share one launch: conv_wgrad_jobs_kernel below)
template <int CIT, int COT, int T>
__device__ __forceinline__ void conv_wgrad_body(const WgArgs& a, const int bx, const int nblk) {
  extern __shared__ __attribute__((aligned(16))) float wg_lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int gw = bx * 4 + wave, nw = nblk * 4;
  const int nex = a.nex, ney = a.ney;                          // elements per sample
  // ---- tables (once per block), one entry per (lane, register) slot of the staging below: the LDS index of the slot's
  // element (slots past the sample's end point at a scratch word behind the wave's image), the input mask, pre_sub
  constexpr int NSX = CIT * 1024, NSY = COT * 1024;           // slots: 64 lanes x 16 registers per channel tile
  unsigned short* lut_x = reinterpret_cast<unsigned short*>(wg_lds);
  unsigned short* lut_y = lut_x + NSX;
  float* mul = reinterpret_cast<float*>(lut_y + NSY);
  float* psl = mul + NSX;
  const int wsz = (a.cin + a.cout) * a.CS + 4;                 // (+ the scratch word, 16-byte granularity)
  auto lidx = [&](int e) {
    const int c = (int)__umulhi((unsigned)e, a.m_hw);
    const int p = e - c * a.HW;
    const int r = (int)__umulhi((unsigned)p, a.m_w);
    return c * a.CS + a.base + p + r * (a.S - a.W);
  };
  for (int e = threadIdx.x; e < NSX; e += 256) {
    lut_x[e] = (unsigned short)(e < nex ? lidx(e) : wsz - 4);
    mul[e] = (a.in_mul && e < nex) ? a.in_mul[e] : 1.f;
  }
  for (int e = threadIdx.x; e < NSY; e += 256) lut_y[e] = (unsigned short)(e < ney ? lidx(e) : wsz - 4 - a.cin * a.CS);
  for (int c = threadIdx.x; c < a.cin; c += 256) psl[c] = a.pre_sub ? a.pre_sub[c] : 0.f;
  float* xl = wg_lds + a.tab_floats + wave * wsz;
  float* dl = xl + a.cin * a.CS;
  for (int i = lane; i < wsz; i += 64) xl[i] = 0.f;           // padding positions stay zero for the whole launch
  __syncthreads();

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

  constexpr int VX = CIT * 16, VY = COT * 16;                 // dwords per lane (HW <= 64): lane l holds elements l + 64 j
  float px[VX], py[VY];
  // no predicates anywhere: a slot past the end re-reads the sample's last element and stores it to the scratch word
  auto gload = [&](int s) {
    const float* xs = a.x + (int64_t)s * nex;
    const float* ys = a.dy + (int64_t)s * ney;
    int z;                                                     // (opaque zero: the clamped offsets are not kept in registers)
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));
#pragma unroll
    for (int j = 0; j < VX; ++j) {
      const int e = lane + 64 * j + z;
      px[j] = xs[e < nex ? e : nex - 1];
    }
#pragma unroll
    for (int j = 0; j < VY; ++j) {
      const int e = lane + 64 * j + z;
      py[j] = ys[e < ney ? e : ney - 1];
    }
  };
  const bool has_ps = a.pre_sub != nullptr;
  auto stage = [&]() {
    // the table entries are re-read per sample (two LDS reads per element under the matrix work) instead of living in ~150
    // registers for the whole launch: `z` is an opaque zero the compiler cannot hoist the reads across
    int z;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));
#pragma unroll
    for (int j = 0; j < VX; ++j) {
      const int e = lane + 64 * j + z;
      float val = px[j];
      if (has_ps) val -= psl[__umulhi((unsigned)(e < nex ? e : 0), a.m_hw)];   // (wave-uniform; the 1 x 1 fallback path only)
      val = val > 0.f ? val : val * a.slope_eff;
      xl[lut_x[e]] = val * mul[e];
      if ((j & 7) == 7) __builtin_amdgcn_sched_barrier(0);      // eight elements' table reads in flight, not all of them
    }
#pragma unroll
    for (int j = 0; j < VY; ++j) {
      dl[lut_y[lane + 64 * j + z]] = py[j];
      if ((j & 7) == 7) __builtin_amdgcn_sched_barrier(0);
    }
  };

  // rsplit waves share a sample (few samples: 32 of them would occupy 32 of the chip's 1024 wave slots for ~600 dependent
  // MFMAs each): every wave stages the sample but multiplies only its range [c0, c1) of the position chunks -- partial sums
  // like any other wave's.  4 * gridDim.x is a multiple of rsplit, so a wave keeps its part for the whole launch.
  const int rs = a.rsplit, part = gw % rs;
  const int c0 = part * a.nch / rs, c1 = (part + 1) * a.nch / rs;
  const int nsw = nw / rs;                                     // samples in flight over the grid
  int s = gw / rs;
  if (s < a.B) gload(s);
  const int loff = (lane & 15) * a.CS + a.q0 + (lane >> 4);
  constexpr int NR = COT + CIT * T, NM = COT * CIT * T;        // fragment reads / MFMAs per chunk
  for (; s < a.B; s += nsw) {
    stage();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (s + nsw < a.B) gload(s + nsw);
    float av[2][COT], bv[2][CIT][T];
    auto frags = [&](int ch, int slot) {                       // (a chunk past the last one reads slack, never used)
      const int o = loff + 4 * ch;
#pragma unroll
      for (int i = 0; i < COT; ++i) av[slot][i] = dl[i * 16 * a.CS + o];
#pragma unroll
      for (int j = 0; j < CIT; ++j)
#pragma unroll
        for (int t = 0; t < T; ++t) bv[slot][j][t] = xl[j * 16 * a.CS + o + a.toff[t]];
    };
    auto mults = [&](int slot) {
#pragma unroll
      for (int i = 0; i < COT; ++i) {
        bs[i] += av[slot][i];
#pragma unroll
        for (int j = 0; j < CIT; ++j)
#pragma unroll
          for (int t = 0; t < T; ++t)
            acc[i][j][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[slot][i], bv[slot][j][t], acc[i][j][t], 0, 0, 0);
      }
    };
    auto interleave = [&]() {                                  // one fragment read of the next chunk per MFMA of this one
#pragma unroll
      for (int r = 0; r < (NR < NM ? NR : NM); ++r) {
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      }
    };
    frags(c0, 0);
    for (int ch = c0; ch + 1 < c1; ch += 2) {
      frags(ch + 1, 1);
      mults(0);
      interleave();
      frags(ch + 2, 0);
      mults(1);
      interleave();
    }
    if ((c1 - c0) & 1) mults(0);
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

template <int CIT, int COT, int T>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgArgs a) {
  conv_wgrad_body<CIT, COT, T>(a, (int)blockIdx.x, (int)gridDim.x);
}

// MANY weight gradients of one tile shape in ONE launch (usf_conv_wgrad_jobs_f32): block b works on job block_job[b] as block
// b - first_block of that job's own launch would -- same partial slots, same bits.  At the reference's training batch (32
// rows) one weight gradient is 128 waves (an eighth of the chip) for 10-14 us, and the 75 of a backward pass of the live MNIST
// configuration follow each other in stream order although nothing but the parameter update waits for them: queued
// (usf_conv_wgrad_plan_f32) they leave the chain of dependent launches and fill the chip together when the pass ends.
static_assert(sizeof(WgArgs) <= sizeof(((usf_wgrad_job*)nullptr)->args), "usf_wgrad_job.args too small for the kernel's arguments");

template <int CIT, int COT, int T>
__global__ __launch_bounds__(256) void conv_wgrad_jobs_kernel(const usf_wgrad_job* __restrict__ jobs, const int32_t* __restrict__ block_job) {
  const usf_wgrad_job* j = jobs + block_job[blockIdx.x];
  const WgArgs a = *reinterpret_cast<const WgArgs*>(j->args);
  conv_wgrad_body<CIT, COT, T>(a, (int)blockIdx.x - j->first_block, j->blocks);
}

// Kernel 1 (the pointwise convolution of GatedConv, the 1 x 1 convolution of BlockAffineTransform) without an input mask,
// 48 < HW <= 64: no tap shifts, so the order of the positions inside the sum is free as long as both operands use the same
// one.  Lane (channel c = l & 15, quad kq = l >> 4) reads the 16 bytes of positions 16 m + 4 kq + {0..3} of its channel row
// from HBM (64 contiguous bytes per row and instruction) and MFMA (m, r) takes component r of every lane: no LDS, no
// staging, no shuffles, no branches; a handful of waves per SIMD hide the latency.  A row that is not a multiple of 16 long
// reads its LAST 16 positions as the fourth block (whole loads, all inside the row) and the positions the third block
// already covered are zeroed in the dy operand, so every position enters the sums once.
// HBM-bound: 4 (cin + cout) bytes per pixel.
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));     // (channel rows of 49 floats are only 4-byte aligned)

template <int CIT, int COT>
__global__ __launch_bounds__(256) void conv_wgrad_k1_kernel(const WgArgs a) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int gw = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
  const int ch_l = lane & 15, kq = lane >> 4;
  f32x4 acc[COT][CIT];
  float bs[COT], ps[CIT];
#pragma unroll
  for (int i = 0; i < COT; ++i) {
    bs[i] = 0.f;
#pragma unroll
    for (int j = 0; j < CIT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int j = 0; j < CIT; ++j) ps[j] = a.pre_sub ? a.pre_sub[j * 16 + ch_l] : 0.f;
  const int HW = a.HW;
  int p0[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) p0[m] = 16 * m + 4 * kq;
  p0[3] = HW - 16 + 4 * kq;                                    // (== 48 + 4 kq when HW == 64)
  f32x4 keep;                                                  // 1 where the fourth block's position is new, 0 where block 2 had it
#pragma unroll
  for (int r = 0; r < 4; ++r) keep[r] = (p0[3] + r >= 48) ? 1.f : 0.f;
  for (int s = gw; s < a.B; s += nw) {
    const float* xs = a.x + ((int64_t)s * a.cin + ch_l) * HW;
    const float* ys = a.dy + ((int64_t)s * a.cout + ch_l) * HW;
    f32x4 fa[COT][4], fb[CIT][4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
#pragma unroll
      for (int i = 0; i < COT; ++i) fa[i][m] = *reinterpret_cast<const f32x4u*>(ys + i * 16 * HW + p0[m]);
#pragma unroll
      for (int j = 0; j < CIT; ++j) fb[j][m] = *reinterpret_cast<const f32x4u*>(xs + j * 16 * HW + p0[m]);
    }
#pragma unroll
    for (int i = 0; i < COT; ++i) fa[i][3] = fa[i][3] * keep;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float bvv[CIT];
#pragma unroll
        for (int j = 0; j < CIT; ++j) {
          const float v = fb[j][m][r] - ps[j];
          bvv[j] = v > 0.f ? v : v * a.slope_eff;
        }
#pragma unroll
        for (int i = 0; i < COT; ++i) {
          bs[i] += fa[i][m][r];
#pragma unroll
          for (int j = 0; j < CIT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][m][r], bvv[j], acc[i][j], 0, 0, 0);
        }
      }
  }
  float* pw = a.part + (int64_t)gw * a.nacc;
#pragma unroll
  for (int i = 0; i < COT; ++i)
#pragma unroll
    for (int j = 0; j < CIT; ++j) {
      const int tile = i * CIT + j;
#pragma unroll
      for (int r = 0; r < 4; ++r) pw[tile * 256 + r * 64 + lane] = acc[i][j][r];
    }
#pragma unroll
  for (int i = 0; i < COT; ++i) {
    float v = bs[i];
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    if (lane < 16) pw[COT * CIT * 256 + i * 16 + lane] = v;
  }
}

// out[j] = sum over the partial slots in a fixed order (four interleaved chains, then ((0 + 1) + 2) + 3).
// mode 0: out[j] for j < n.  mode 1 (weight gradient): slot layout of conv_wgrad_kernel -> dW [cout][cin][T], db [cout].
// blockIdx.y selects a slice of `per` slots and (mode 0) its own output row: reduce_partials below adds many slots in two
// rounds so that the sum over thousands of slots is not left to a handful of blocks.
__device__ __forceinline__ void partial_sum_block(const float* __restrict__ part, int nparts, int per, int n, float* __restrict__ out,
                                                  float* __restrict__ out2, int mode, int cin, int cout, int CIT, int T, int ntile,
                                                  int bx, int by) {
  __shared__ float red[4][64];
  const int jj = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int j = bx * 64 + jj;
  const int p0 = by * per, p1 = (p0 + per < nparts) ? p0 + per : nparts;
  float s = 0.f;
  if (j < n)
    for (int p = p0 + g; p < p1; p += 4) s += part[(int64_t)p * n + j];
  red[g][jj] = s;
  __syncthreads();
  if (g != 0 || j >= n) return;
  const float total = ((red[0][jj] + red[1][jj]) + red[2][jj]) + red[3][jj];
  if (mode == 0) {
    out[(int64_t)by * n + j] = total;
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

__global__ __launch_bounds__(256) void partial_sum_kernel(const float* __restrict__ part, int nparts, int per, int n,
                                                          float* __restrict__ out, float* __restrict__ out2, int mode, int cin,
                                                          int cout, int CIT, int T, int ntile) {
  partial_sum_block(part, nparts, per, n, out, out2, mode, cin, cout, CIT, T, ntile, (int)blockIdx.x, (int)blockIdx.y);
}

// MANY single-round sums in one launch (usf_partial_sum_jobs_f32): block b serves job block_job[b], as block
// b - first_block of partial_sum_kernel would -- same order of additions, same bits.  At the reference's training batch
// (32 rows) a backward pass of the live MNIST configuration asks for ~165 of these sums of a few microseconds each; queued
// (usf_conv_wgrad_deferred_f32) they leave the chain of dependent launches and run as ONE launch when the pass ends.
// mode 0 with n % 4 == 0 and 16-byte aligned rows (job.vec4): a thread adds FOUR neighbouring columns (one 16-byte load per
// slot), a block 256 columns -- the same additions in the same order per column as partial_sum_block: same bits.  The first
// rounds of a pass's weight gradients are ~50 000 blocks of 4 KB in the scalar form (latency-bound: 117 us for the live
// MNIST step at batch 32); a quarter of the blocks with four times the bytes in flight each.
__device__ __forceinline__ void partial_sum_block_vec4(const float* __restrict__ part, int nparts, int per, int n, float* __restrict__ out,
                                                       int bx, int by) {
  __shared__ f32x4 red4[4][64];
  const int jj = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int j = bx * 256 + jj * 4;
  const int p0 = by * per, p1 = (p0 + per < nparts) ? p0 + per : nparts;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (j < n)
    for (int p = p0 + g; p < p1; p += 4) s += *reinterpret_cast<const f32x4*>(part + (int64_t)p * n + j);
  red4[g][jj] = s;
  __syncthreads();
  if (g != 0 || j >= n) return;
  *reinterpret_cast<f32x4*>(out + (int64_t)by * n + j) = ((red4[0][jj] + red4[1][jj]) + red4[2][jj]) + red4[3][jj];
}

__global__ __launch_bounds__(256) void partial_sum_jobs_kernel(const usf_psum_job* __restrict__ jobs, const int32_t* __restrict__ block_job) {
  const usf_psum_job j = jobs[block_job[blockIdx.x]];
  const int local = (int)blockIdx.x - j.first_block;
  if (j.vec4) {
    const int gx = (j.n + 255) / 256;
    partial_sum_block_vec4(j.part, j.nparts, j.per, j.n, j.out, local % gx, local / gx);
    return;
  }
  const int gx = (j.n + 63) / 64;
  partial_sum_block(j.part, j.nparts, j.per, j.n, j.out, j.out2, j.mode, j.cin, j.cout, j.CIT, j.T, j.ntile, local % gx, local / gx);
}

constexpr int kReduceRows = 64;        // scratch rows of the two-round sum (workspace: kReduceRows * n floats behind the slots)

// sum of `nparts` slots of n floats: one round up to 64 slots, else a first round into <= 64 rows of scratch
static int reduce_partials(const float* part, int nparts, int n, float* scratch, float* out, float* out2, int mode, int cin, int cout,
                           int CIT, int T, int ntile, hipStream_t stream, const char* what) {
  const unsigned gx = (unsigned)((n + 63) / 64);
  if (nparts > 64) {
    const int rows = nparts / 16 < kReduceRows ? (nparts + 15) / 16 : kReduceRows;
    const int per = (nparts + rows - 1) / rows;
    const int used = (nparts + per - 1) / per;
    hipLaunchKernelGGL(partial_sum_kernel, dim3(gx, (unsigned)used), dim3(256), 0, stream, part, nparts, per, n, scratch,
                       (float*)nullptr, 0, 0, 0, 0, 0, 0);
    part = scratch;
    nparts = used;
  }
  hipLaunchKernelGGL(partial_sum_kernel, dim3(gx, 1u), dim3(256), 0, stream, part, nparts, nparts, n, out, out2, mode, cin, cout, CIT, T, ntile);
  return check_launch(what);
}

// the same sum for other kernels' per-block slots (usf_gated_tail.hip), mode 0: at once (job == NULL), or handed to the caller as
// jobs of usf_partial_sum_jobs_f32 -- job[0] the first round (nparts == 0: none), job[1] the last
int64_t sum_slots_scratch(int64_t n) { return (int64_t)kReduceRows * n; }

int sum_slots(const float* part, int nparts, int n, float* scratch, float* out, usf_psum_job* job, hipStream_t stream, const char* what) {
  if (!job) return reduce_partials(part, nparts, n, scratch, out, nullptr, 0, 0, 0, 0, 0, 0, stream, what);
  job[0] = usf_psum_job{};
  job[1] = usf_psum_job{};
  if (nparts > 64) {
    const int rows = nparts / 16 < kReduceRows ? (nparts + 15) / 16 : kReduceRows;
    const int per = (nparts + rows - 1) / rows;
    const int used = (nparts + per - 1) / per;
    usf_psum_job& a0 = job[0];
    a0.part = part; a0.out = scratch; a0.nparts = nparts; a0.n = n; a0.per = per; a0.rows = used;
    a0.vec4 = (n % 4 == 0 && aligned16(part) && aligned16(scratch)) ? 1 : 0;
    part = scratch;
    nparts = used;
  }
  usf_psum_job& a1 = job[1];
  a1.part = part; a1.out = out; a1.nparts = nparts; a1.n = n; a1.per = nparts; a1.rows = 1;
  a1.vec4 = (n % 4 == 0 && aligned16(part) && aligned16(out)) ? 1 : 0;
  return 0;
}

// ceil(2^32 / d), d >= 2: floor(n / d) == umulhi(n, magic) for n * d < 2^32
static unsigned magic_div(int d) { return (unsigned)((0x100000000ULL + (unsigned)d - 1) / (unsigned)d); }

// workspace floats needed by conv_wgrad for this shape / batch (0: shape not served)
int64_t conv_wgrad_workspace(int64_t B, int64_t cin, int64_t cout, int64_t H, int64_t W, int64_t ks);

#define USF_WG_INSTANCES(X)                                                                                                \
  X(1, 1, 9) X(1, 2, 9) X(2, 1, 9) X(2, 2, 9) X(3, 2, 9) X(2, 3, 9) X(1, 3, 9) X(3, 1, 9)                                  \
  X(1, 1, 1) X(1, 2, 1) X(2, 1, 1) X(2, 2, 1) X(2, 4, 1) X(4, 2, 1) X(3, 3, 1) X(4, 4, 1)                                  \
  X(1, 4, 1) X(4, 1, 1) X(3, 2, 1) X(2, 3, 1) X(1, 3, 1) X(3, 1, 1) X(3, 4, 1) X(4, 3, 1)

#define USF_WG_K1_INSTANCES(X)                                                                                             \
  X(1, 1) X(1, 2) X(1, 3) X(1, 4) X(2, 1) X(2, 2) X(2, 3) X(2, 4) X(3, 1) X(3, 2) X(3, 3) X(3, 4) X(4, 1) X(4, 2) X(4, 3) X(4, 4)

namespace {
int wgrad_k1_blocks_per_cu(int CIT, int COT) {
  int nb = 0;
#define USF_WG_OCC1(CIT_, COT_)                                                                                            \
  if (CIT == CIT_ && COT == COT_ &&                                                                                        \
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(&conv_wgrad_k1_kernel<CIT_, COT_>),  \
                                                   256, 0) != hipSuccess) nb = 1;
  USF_WG_K1_INSTANCES(USF_WG_OCC1)
#undef USF_WG_OCC1
  return nb < 1 ? 1 : nb;
}

// resident blocks per CU of an instance at this LDS size (registers and LDS decide); 0: no such instance
int wgrad_blocks_per_cu(int CIT, int COT, int T, int lds_bytes) {
  static bool attr_done[USF_MAX_DEVICES][5][5][2];
  const int dev = current_device_slot();
  int nb = 0;
#define USF_WG_OCC(CIT_, COT_, T_)                                                                                         \
  if (CIT == CIT_ && COT == COT_ && T == T_) {                                                                             \
    const void* fn = reinterpret_cast<const void*>(&conv_wgrad_kernel<CIT_, COT_, T_>);                                    \
    if (!attr_done[dev][CIT_][COT_][T_ == 9]) {                                                                            \
      if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return 0;         \
      attr_done[dev][CIT_][COT_][T_ == 9] = true;                                                                          \
    }                                                                                                                      \
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, 256, (size_t)lds_bytes) != hipSuccess) nb = 1;               \
    if (nb < 1) nb = 1;                                                                                                    \
  }
  USF_WG_INSTANCES(USF_WG_OCC)
#undef USF_WG_OCC
  return nb;
}

struct WgPlan {
  int CIT, COT, T, S, base, CS, nch, q0, nacc, lds_bytes, blocks, tab_floats, rsplit;
  bool direct;               // kernel 1 without an input mask: conv_wgrad_k1_kernel
};
bool wgrad_plan(int64_t B, int64_t cin, int64_t cout, int64_t H, int64_t W, int64_t ks, bool masked, WgPlan& pl) {
  if (B <= 0 || cin <= 0 || cout <= 0 || (cin & 15) || (cout & 15) || H <= 0 || W < 2 || H * W > 64 || (ks != 1 && ks != 3))
    return false;                                                // (W >= 2: the index tables divide by W and H * W with 32-bit magic numbers)
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
    need = pl.q0 + 4 * (pl.nch + 1) + pl.S + 1;                   // last position read (one chunk of slack) + 1
  } else {
    pl.S = (int)W;
    pl.base = 0;
    pl.q0 = 0;
    pl.nch = (int)((H * W + 3) / 4);
    need = 4 * (pl.nch + 1);
  }
  const int full = (ks == 3) ? (int)((H + 2) * pl.S + 1) : (int)(H * W);
  if (need < full) need = full;
  int cs = (need + 1) / 2;                                        // CS = 2 * odd >= need
  if (!(cs & 1)) ++cs;
  pl.CS = 2 * cs;
  pl.nacc = pl.COT * pl.CIT * pl.T * 256 + pl.COT * 16;
  pl.direct = (ks == 1 && !masked && H * W > 48);
  pl.rsplit = 1;
  if (pl.direct) {
    pl.lds_bytes = 0;
    int per_cu = wgrad_k1_blocks_per_cu(pl.CIT, pl.COT);
    if (per_cu > 8) per_cu = 8;
    int64_t blocks = (B + 3) / 4;
    const int64_t cap = (int64_t)device_cu_count() * per_cu;
    pl.blocks = (int)(blocks > cap ? cap : blocks);
    return true;
  }
  pl.tab_floats = (pl.CIT * 1024 + pl.COT * 1024) / 2 + pl.CIT * 1024 + (int)cin;      // index tables (u16), mask, pre_sub
  pl.tab_floats = (pl.tab_floats + 3) & ~3;
  pl.lds_bytes = (pl.tab_floats + 4 * ((int)(cin + cout) * pl.CS + 4)) * 4;
  if (pl.lds_bytes > 160 * 1024) return false;
  int per_cu = wgrad_blocks_per_cu(pl.CIT, pl.COT, pl.T, pl.lds_bytes);
  if (per_cu <= 0) return false;
  if (per_cu > 4) per_cu = 4;                                    // (more partial slots than that buy nothing)
  const int64_t cap = (int64_t)device_cu_count() * per_cu;
  // few samples: 2 or 4 waves per sample (position chunks dealt over them) while that still fits one round of blocks
  const int rs_on = (int)tuning("conv_wgrad_rsplit", 1);   // tuning aid: 0 = one wave per sample
  if (rs_on && pl.T == 9) {
    if (B <= 64 && B <= cap && pl.nch >= 8) pl.rsplit = 4;          // (<= 256 partial slots: still one reduction round)
    else if (B <= 128 && B <= 2 * cap && pl.nch >= 4) pl.rsplit = 2;
  }
  int64_t blocks = (B * pl.rsplit + 3) / 4;
  if (blocks > cap) blocks = cap;
  pl.blocks = (int)blocks;
  return true;
}
}  // namespace

int64_t conv_wgrad_workspace(int64_t B, int64_t cin, int64_t cout, int64_t H, int64_t W, int64_t ks) {
  // (the masked form of kernel 1 needs at least the slots of the direct form: both sized here)
  WgPlan pl, pm;
  if (!wgrad_plan(B, cin, cout, H, W, ks, false, pl) || !wgrad_plan(B, cin, cout, H, W, ks, true, pm)) return 0;
  return ((int64_t)(pl.blocks > pm.blocks ? pl.blocks : pm.blocks) * 4 + kReduceRows) * pl.nacc;
}

int conv_wgrad(const float* x, const float* dy, int64_t B, int64_t cin, int64_t cout, int64_t H, int64_t W, int64_t ks,
               const float* in_mul, const float* pre_sub, int32_t in_act, float in_slope, float* dW, float* db, float* workspace,
               int64_t workspace_floats, usf_psum_job* job, usf_wgrad_job* wjob, hipStream_t stream) {
  WgPlan pl;
  if (job) job[0].nparts = job[1].nparts = 0;
  if (wjob) wjob->blocks = 0;
  if (B < 0) { set_error("usf_conv_wgrad_f32: bad sizes"); return -2; }
  if (B == 0 || !wgrad_plan(B, cin, cout, H, W, ks, in_mul != nullptr, pl)) {
    if (B == 0) { set_error("usf_conv_wgrad_f32: empty batch"); return -2; }
    return 1;                                                     // shape not served
  }
  if (!x || !dy || !dW || !workspace) { set_error("usf_conv_wgrad_f32: null pointer"); return -1; }
  if (!aligned16(x) || !aligned16(dy) || (in_mul && !aligned16(in_mul))) { set_error("usf_conv_wgrad_f32: tensors must be 16-byte aligned"); return -2; }
  if (in_act != USF_ACT_NONE && in_act != USF_ACT_LEAKY_RELU) { set_error("usf_conv_wgrad_f32: bad act"); return -2; }
  if (workspace_floats < ((int64_t)pl.blocks * 4 + kReduceRows) * pl.nacc) { set_error("usf_conv_wgrad_f32: workspace too small"); return -2; }
  WgArgs a;
  a.x = x; a.dy = dy; a.part = workspace; a.in_mul = in_mul; a.pre_sub = pre_sub;
  a.B = (int)B; a.cin = (int)cin; a.cout = (int)cout; a.HW = (int)(H * W); a.W = (int)W; a.S = pl.S; a.base = pl.base;
  a.CS = pl.CS; a.nch = pl.nch; a.q0 = pl.q0; a.nex = (int)(cin * H * W); a.ney = (int)(cout * H * W); a.nacc = pl.nacc;
  a.rsplit = pl.rsplit;
  a.m_hw = magic_div(a.HW); a.m_w = magic_div(a.W); a.slope_eff = (in_act == USF_ACT_LEAKY_RELU) ? in_slope : 1.f;
  a.tab_floats = pl.tab_floats;
  for (int t = 0; t < 9; ++t) a.toff[t] = (ks == 3) ? ((t / 3) - 1) * pl.S + ((t % 3) - 1) : 0;
  const dim3 g((unsigned)pl.blocks), b(256);
#define USF_WG(CIT_, COT_, T_)                                                                                             \
  if (pl.CIT == CIT_ && pl.COT == COT_ && pl.T == T_) {                                                                     \
    hipLaunchKernelGGL((conv_wgrad_kernel<CIT_, COT_, T_>), g, b, (size_t)pl.lds_bytes, stream, a);                        \
    launched = true;                                                                                                       \
  }
  bool launched = false;
  if (wjob && !pl.direct) {
    // not launched: the caller queues the job for usf_conv_wgrad_jobs_f32 (the sums below follow it)
    memset(wjob, 0, sizeof(*wjob));
    memcpy(wjob->args, &a, sizeof(a));
    wjob->CIT = pl.CIT; wjob->COT = pl.COT; wjob->T = pl.T; wjob->blocks = pl.blocks; wjob->lds_bytes = pl.lds_bytes;
    launched = true;
  } else if (pl.direct) {
#define USF_WG1(CIT_, COT_)                                                                                                \
  if (pl.CIT == CIT_ && pl.COT == COT_) {                                                                                  \
    hipLaunchKernelGGL((conv_wgrad_k1_kernel<CIT_, COT_>), g, b, 0, stream, a);                                            \
    launched = true;                                                                                                       \
  }
    USF_WG_K1_INSTANCES(USF_WG1)
#undef USF_WG1
  } else {
    USF_WG_INSTANCES(USF_WG)
  }
#undef USF_WG
  if (!launched) return 1;
  int rc = check_launch("usf_conv_wgrad_f32");
  if (rc) return rc;
  const int ntile = pl.COT * pl.CIT * pl.T;
  if (job) {
    // the sums are handed to the caller as jobs of usf_partial_sum_jobs_f32 (the outputs stay unwritten until then): job[1] the
    // final round, job[0] the first round into the scratch rows when there are more than 64 slots (reduce_partials' two rounds)
    int nparts = pl.blocks * 4;
    const float* src = workspace;
    float* scratch = workspace + (int64_t)pl.blocks * 4 * pl.nacc;
    job[0].nparts = 0;
    if (nparts > 64) {
      const int rows = nparts / 16 < kReduceRows ? (nparts + 15) / 16 : kReduceRows;
      const int per = (nparts + rows - 1) / rows;
      const int used = (nparts + per - 1) / per;
      usf_psum_job& a0 = job[0];
      a0.part = workspace; a0.out = scratch; a0.out2 = nullptr; a0.nparts = nparts; a0.n = pl.nacc; a0.mode = 0; a0.cin = a0.cout = 0;
      a0.CIT = a0.T = a0.ntile = 0; a0.first_block = 0; a0.per = per; a0.rows = used;
      a0.vec4 = (pl.nacc % 4 == 0 && aligned16(workspace) && aligned16(scratch)) ? 1 : 0;
      src = scratch;
      nparts = used;
    }
    usf_psum_job& a1 = job[1];
    a1.part = src; a1.out = dW; a1.out2 = db; a1.nparts = nparts; a1.n = pl.nacc; a1.mode = 1; a1.cin = (int)cin; a1.cout = (int)cout;
    a1.CIT = pl.CIT; a1.T = pl.T; a1.ntile = ntile; a1.first_block = 0; a1.per = nparts; a1.rows = 1; a1.vec4 = 0;
    return 0;
  }
  return reduce_partials(workspace, pl.blocks * 4, pl.nacc, workspace + (int64_t)pl.blocks * 4 * pl.nacc, dW, db, 1, (int)cin, (int)cout,
                         pl.CIT, pl.T, ntile, stream, "usf_conv_wgrad_f32 (reduce)");
}

int conv_wgrad_jobs(const usf_wgrad_job* jobs, const int32_t* block_job, int64_t n_blocks, int32_t CIT, int32_t COT, int32_t T,
                    int32_t lds_bytes, hipStream_t stream) {
  if (n_blocks < 0 || n_blocks > 0x7fffffff || (n_blocks > 0 && (!jobs || !block_job)) || lds_bytes < 0 || lds_bytes > 160 * 1024) {
    set_error("usf_conv_wgrad_jobs_f32: bad arguments");
    return -1;
  }
  if (n_blocks == 0) return 0;
  static bool attr_done[USF_MAX_DEVICES][5][5][2];
  const int dev = current_device_slot();
  bool launched = false;
#define USF_WGJ(CIT_, COT_, T_)                                                                                            \
  if (CIT == CIT_ && COT == COT_ && T == T_) {                                                                             \
    const void* fn = reinterpret_cast<const void*>(&conv_wgrad_jobs_kernel<CIT_, COT_, T_>);                               \
    if (!attr_done[dev][CIT_][COT_][T_ == 9]) {                                                                            \
      if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {                 \
        set_error("usf_conv_wgrad_jobs_f32: cannot raise the kernel's LDS limit");                                        \
        return -3;                                                                                                         \
      }                                                                                                                    \
      attr_done[dev][CIT_][COT_][T_ == 9] = true;                                                                          \
    }                                                                                                                      \
    hipLaunchKernelGGL((conv_wgrad_jobs_kernel<CIT_, COT_, T_>), dim3((unsigned)n_blocks), dim3(256), (size_t)lds_bytes, stream, jobs,  \
                       block_job);                                                                                         \
    launched = true;                                                                                                       \
  }
  USF_WG_INSTANCES(USF_WGJ)
#undef USF_WGJ
  if (!launched) { set_error("usf_conv_wgrad_jobs_f32: no kernel instance for tiles (%d, %d, %d)", CIT, COT, T); return -2; }
  return check_launch("usf_conv_wgrad_jobs_f32");
}

int partial_sum_jobs(const usf_psum_job* jobs, const int32_t* block_job, int64_t n_blocks, hipStream_t stream) {
  if (n_blocks < 0 || n_blocks > 0x7fffffff || (n_blocks > 0 && (!jobs || !block_job))) {
    set_error("usf_partial_sum_jobs_f32: bad arguments");
    return -1;
  }
  if (n_blocks == 0) return 0;
  hipLaunchKernelGGL(partial_sum_jobs_kernel, dim3((unsigned)n_blocks), dim3(256), 0, stream, jobs, block_job);
  return check_launch("usf_partial_sum_jobs_f32");
}

// ------------------------------------------------------------------------------------------
// LayerNormChannels backward (with the folded (Leaky)ReLU): per pixel over the channel axis
//   a = act(x), xh = (a - mean) / den, y = xh * gamma + beta
//   g = dy * gamma, da = (g - mean_c g - xh * mean_c (g * xh)) / den, dx = da * act'(x)
//   dgamma[c] = sum_{b, p} dy * xh, dbeta[c] = sum_{b, p} dy
// One thread per pixel, grid-stride; the parameter sums stay in registers and leave once per wave.
// HBM-bound: 12 bytes per element.
// ------------------------------------------------------------------------------------------
template <int CMAX, bool FULL>
__global__ __launch_bounds__(256) void layernorm_channels_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                     float* __restrict__ dx, int64_t BP, int C_, int64_t P,
                                                                     const float* __restrict__ gamma, float eps, int act,
                                                                     float slope, float* __restrict__ part) {
  const int C = FULL ? CMAX : C_;                              // FULL: every channel slot is real -- no predicates, straight-line loads
  float dg[CMAX], dbt[CMAX], gam[CMAX];
#pragma unroll
  for (int c = 0; c < CMAX; ++c) {
    dg[c] = dbt[c] = 0.f;
    gam[c] = (FULL || c < C) ? gamma[c] : 0.f;
  }
  const float inv_c = 1.f / (float)C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < BP; i += (int64_t)gridDim.x * 256) {
    const int64_t b = i / P, p = i - b * P;
    const float* xb = x + b * C * P + p;
    const float* gb = dy + b * C * P + p;
    float v[CMAX], d[CMAX];
    unsigned long long pos = 0;                                // bit c: x[c] > 0 (the nonlinearity's gate)
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {                           // both tensors' loads are in flight together
      v[c] = (FULL || c < C) ? xb[(int64_t)c * P] : 0.f;
      d[c] = (FULL || c < C) ? gb[(int64_t)c * P] : 0.f;
    }
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
      if (v[c] > 0.f) pos |= 1ull << c;
      if (FULL || c < C) v[c] = act_apply(v[c], act, slope);
      sum += v[c];
    }
    const float mean = sum * inv_c;
    float sq = 0.f;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
      const float t = (FULL || c < C) ? v[c] - mean : 0.f;
      sq += t * t;
    }
    const float rden = 1.f / sqrtf(sq * inv_c + eps);
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
      v[c] = (FULL || c < C) ? (v[c] - mean) * rden : 0.f;    // xh
      const float g = d[c] * gam[c];
      m1 += g;
      m2 += g * v[c];
      dg[c] += d[c] * v[c];
      dbt[c] += d[c];
      d[c] = g;
    }
    m1 *= inv_c;
    m2 *= inv_c;
    float* ob = dx + b * C * P + p;
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
      if (FULL || c < C) {
        float da = (d[c] - m1 - v[c] * m2) * rden;
        if (act == USF_ACT_LEAKY_RELU && !((pos >> c) & 1ull)) da *= slope;
        ob[(int64_t)c * P] = da;
      }
  }
  const int lane = threadIdx.x & 63;
  float* pw = part + ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * (2 * C);
#pragma unroll
  for (int c = 0; c < CMAX; ++c)
    if (FULL || c < C) {
      const float a = wave_sum(dg[c]), bsum = wave_sum(dbt[c]);
      if (lane == 0) { pw[c] = a; pw[C + c] = bsum; }
    }
}

// C == 2 * CPT channels, two lanes per pixel (lane l and l ^ 32 share pixel l & 31; the lower half owns channels [0, CPT), the
// upper half [CPT, 2 CPT)): half the registers of the one-thread-per-pixel form, so 3-4 waves per SIMD keep the loads in
// flight; every wave instruction reads two 128-byte runs.  The four per-pixel sums cross the halves by one exchange each.
template <int CPT>
__global__ __launch_bounds__(256) void layernorm_channels_bwd2_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                      float* __restrict__ dx, int64_t BP, int64_t P,
                                                                      const float* __restrict__ gamma, float eps, int act,
                                                                      float slope, float* __restrict__ part) {
  constexpr int C = 2 * CPT;
  const int lane = threadIdx.x & 63, half = lane >> 5, pl = lane & 31;
  const int c0 = half * CPT;
  float dg[CPT], dbt[CPT], gam[CPT];
#pragma unroll
  for (int c = 0; c < CPT; ++c) {
    dg[c] = dbt[c] = 0.f;
    gam[c] = gamma[c0 + c];
  }
  const float inv_c = 1.f / (float)C;
  const int64_t wv = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwv = (int64_t)gridDim.x * 4;
  for (int64_t base = wv * 32; base < BP; base += nwv * 32) {       // wave-uniform trip count (the exchanges need every lane)
    const bool valid = base + pl < BP;
    const int64_t i = valid ? base + pl : BP - 1;
    const int64_t b = i / P, p = i - b * P;
    const float* xb = x + (b * C + c0) * P + p;
    const float* gb = dy + (b * C + c0) * P + p;
    float v[CPT], d[CPT];
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      v[c] = xb[(int64_t)c * P];
      d[c] = valid ? gb[(int64_t)c * P] : 0.f;
    }
    unsigned pos = 0;
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      if (v[c] > 0.f) pos |= 1u << c;
      v[c] = act_apply(v[c], act, slope);
      sum += v[c];
    }
    sum += __shfl_xor(sum, 32, 64);
    const float mean = sum * inv_c;
    float sq = 0.f;
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      const float t = v[c] - mean;
      sq += t * t;
    }
    sq += __shfl_xor(sq, 32, 64);
    const float rden = 1.f / sqrtf(sq * inv_c + eps);
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      v[c] = (v[c] - mean) * rden;                               // xh
      const float g = d[c] * gam[c];
      m1 += g;
      m2 += g * v[c];
      dg[c] += d[c] * v[c];
      dbt[c] += d[c];
      d[c] = g;
    }
    m1 += __shfl_xor(m1, 32, 64);
    m2 += __shfl_xor(m2, 32, 64);
    m1 *= inv_c;
    m2 *= inv_c;
    if (valid) {
      float* ob = dx + (b * C + c0) * P + p;
#pragma unroll
      for (int c = 0; c < CPT; ++c) {
        float da = (d[c] - m1 - v[c] * m2) * rden;
        if (act == USF_ACT_LEAKY_RELU && !((pos >> c) & 1u)) da *= slope;
        ob[(int64_t)c * P] = da;
      }
    }
  }
  float* pw = part + wv * (2 * C);
#pragma unroll
  for (int c = 0; c < CPT; ++c) {
    float a = dg[c], bsum = dbt[c];
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {                           // within each half
      a += __shfl_xor(a, o, 64);
      bsum += __shfl_xor(bsum, o, 64);
    }
    if (pl == 0) { pw[c0 + c] = a; pw[C + c0 + c] = bsum; }
  }
}

int64_t layernorm_channels_bwd_workspace(int64_t B, int64_t C, int64_t P) {
  if (B <= 0 || C <= 0 || C > 64 || P <= 0) return 0;
  int64_t blocks = (B * P + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  return (blocks * 4 + kReduceRows) * 2 * C;
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
  const int blocks = (int)(need / (2 * C) - kReduceRows) / 4;
  const dim3 g((unsigned)blocks), b(256);
#define USF_LNB(CM_, FULL_) hipLaunchKernelGGL((layernorm_channels_bwd_kernel<CM_, FULL_>), g, b, 0, stream, x, dy, dx, BP, (int)C, P, gamma, eps, act, slope, workspace)
#define USF_LNB2(CPT_) hipLaunchKernelGGL((layernorm_channels_bwd2_kernel<CPT_>), g, b, 0, stream, x, dy, dx, BP, P, gamma, eps, act, slope, workspace)
  if (C == 16) USF_LNB2(8);
  else if (C == 32) USF_LNB2(16);
  else if (C == 48) USF_LNB2(24);
  else if (C == 64) USF_LNB2(32);
  else if (C < 16) USF_LNB(16, false);
  else if (C < 32) USF_LNB(32, false);
  else USF_LNB(64, false);
#undef USF_LNB
#undef USF_LNB2
  int rc = check_launch("usf_layernorm_channels_bwd_f32");
  if (rc) return rc;
  const int n = (int)(2 * C);
  return reduce_partials(workspace, blocks * 4, n, workspace + (int64_t)blocks * 4 * n, dgamma, nullptr, 0, 0, 0, 0, 0, 0, stream,
                         "usf_layernorm_channels_bwd_f32 (reduce)");
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

__device__ __forceinline__ void conv_weight_planes_elem(const float* __restrict__ w, unsigned short* __restrict__ planes, int i, int rows,
                                                        int cols, int ks2, int cp, int kp, int coutp, int transposed) {
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

__global__ __launch_bounds__(256) void conv_weight_planes_kernel(const float* __restrict__ w, unsigned short* __restrict__ planes, int rows,
                                                                 int cols, int ks2, int cp, int kp, int coutp, int transposed) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < coutp * kp) conv_weight_planes_elem(w, planes, i, rows, cols, ks2, cp, kp, coutp, transposed);
}

// both orientations of one weight in ONE launch (a training step needs the forward planes and, for the data gradient, the
// transposed ones; at small batches every launch saved is ~5 us of a dependent chain): planes | planes_t, back to back
__global__ __launch_bounds__(256) void conv_weight_planes_pair_kernel(const float* __restrict__ w, unsigned short* __restrict__ planes,
                                                                      int cin, int cout, int ks2, int cp_f, int kp_f, int coutp_f,
                                                                      int cp_t, int kp_t, int coutp_t) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < coutp_f * kp_f) conv_weight_planes_elem(w, planes, i, cout, cin, ks2, cp_f, kp_f, coutp_f, 0);
  if (i < coutp_t * kp_t) conv_weight_planes_elem(w, planes + 3 * (int64_t)coutp_f * kp_f, i, cin, cout, ks2, cp_t, kp_t, coutp_t, 1);
}

int conv2d_weight_planes(const float* w, void* planes, int64_t cin, int64_t cout, int64_t ks, int32_t transposed, hipStream_t stream) {
  if (cin <= 0 || cout <= 0 || cin > 4096 || cout > 4096 || (ks != 1 && ks != 3)) { set_error("usf_conv2d_weight_planes_f32: bad sizes"); return -2; }
  if (!w || !planes) { set_error("usf_conv2d_weight_planes_f32: null pointer"); return -1; }
  if (transposed < 0 || transposed > 2) { set_error("usf_conv2d_weight_planes_f32: transposed must be 0, 1 or 2 (both)"); return -2; }
  // rows / cols of the packed matrix: the convolution the planes are FOR maps `cols` channels to `rows` channels
  auto dims = [&](bool t, int& cp, int& kp, int& coutp) {
    const int rows = (int)(t ? cin : cout), cols = (int)(t ? cout : cin);
    cp = (cols + 7) / 8 * 8;
    kp = (int)((ks * ks * cp + 31) / 32 * 32);
    coutp = (rows + 15) / 16 * 16;
  };
  if (transposed == 2) {
    int cp_f, kp_f, coutp_f, cp_t, kp_t, coutp_t;
    dims(false, cp_f, kp_f, coutp_f);
    dims(true, cp_t, kp_t, coutp_t);
    const int n = coutp_f * kp_f > coutp_t * kp_t ? coutp_f * kp_f : coutp_t * kp_t;
    hipLaunchKernelGGL(conv_weight_planes_pair_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, w,
                       reinterpret_cast<unsigned short*>(planes), (int)cin, (int)cout, (int)(ks * ks), cp_f, kp_f, coutp_f, cp_t,
                       kp_t, coutp_t);
    return check_launch("usf_conv2d_weight_planes_f32");
  }
  const int rows = (int)(transposed ? cin : cout), cols = (int)(transposed ? cout : cin);
  int cp, kp, coutp;
  dims(transposed != 0, cp, kp, coutp);
  const int n = coutp * kp;
  hipLaunchKernelGGL(conv_weight_planes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, w,
                     reinterpret_cast<unsigned short*>(planes), rows, cols, (int)(ks * ks), cp, kp, coutp, (int)transposed);
  return check_launch("usf_conv2d_weight_planes_f32");
}

// MANY weights' plane pairs in ONE launch (usf_conv2d_weight_planes_batch_f32): block b serves job block_job[b] as block
// b - first_block of conv_weight_planes_pair_kernel would (same bits); job j's pair lands at planes_base + out_off elements.
// A training step of the live MNIST configuration at batch 32 splits 75 convolution weights: 75 launches of ~5 us on the
// chain of dependent launches that bounds the step -> 1.
__global__ __launch_bounds__(256) void conv_weight_planes_batch_kernel(const usf_wplanes_job* __restrict__ jobs,
                                                                       const int32_t* __restrict__ block_job,
                                                                       unsigned short* __restrict__ planes_base) {
  const usf_wplanes_job j = jobs[block_job[blockIdx.x]];
  const int i = ((int)blockIdx.x - j.first_block) * 256 + threadIdx.x;
  const int ks2 = j.ks * j.ks;
  const int cp_f = (j.cin + 7) / 8 * 8, kp_f = (ks2 * cp_f + 31) / 32 * 32, coutp_f = (j.cout + 15) / 16 * 16;
  const int cp_t = (j.cout + 7) / 8 * 8, kp_t = (ks2 * cp_t + 31) / 32 * 32, coutp_t = (j.cin + 15) / 16 * 16;
  unsigned short* planes = planes_base + j.out_off;
  if (i < coutp_f * kp_f) conv_weight_planes_elem(j.w, planes, i, j.cout, j.cin, ks2, cp_f, kp_f, coutp_f, 0);
  if (i < coutp_t * kp_t) conv_weight_planes_elem(j.w, planes + 3 * (int64_t)coutp_f * kp_f, i, j.cin, j.cout, ks2, cp_t, kp_t, coutp_t, 1);
}

int conv2d_weight_planes_batch(const usf_wplanes_job* jobs, const int32_t* block_job, int64_t n_blocks, void* planes_base,
                               hipStream_t stream) {
  if (n_blocks < 0 || n_blocks > 0x7fffffff || (n_blocks > 0 && (!jobs || !block_job || !planes_base))) {
    set_error("usf_conv2d_weight_planes_batch_f32: bad arguments");
    return -1;
  }
  if (n_blocks == 0) return 0;
  hipLaunchKernelGGL(conv_weight_planes_batch_kernel, dim3((unsigned)n_blocks), dim3(256), 0, stream, jobs, block_job,
                     reinterpret_cast<unsigned short*>(planes_base));
  return check_launch("usf_conv2d_weight_planes_batch_f32");
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
