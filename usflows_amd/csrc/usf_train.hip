// Backward-pass kernels of the training step: what autograd derives from Flow.log_prob
// (flows.py:196-199: loss = -log_prob(batch).mean() - log_prior) for the layers of the hot path.
//
//   wgrad        G[n,k] = sum_m Y[m,n] * A[m,k]        weight gradient of every F.linear on the path
//                (BlockAffineTransform transforms.py:913-962, conditioner Linear layers networks.py:739-751):
//                a GEMM whose reduction runs over the BATCH, exact-f32 MFMA (v_mfma_f32_16x16x4_f32), split over
//                row ranges into partials + a deterministic reduction (no atomics: bitwise reproducible gradients)
//   colsum       g_bias[n] = sum_m Y[m,n]               bias gradient, same two-stage scheme; in the bf16x3 kernels the
//                column sums of Y can ride in the weight-gradient pass itself (usf_wgrad_bias_f32, round 4: three more MFMAs
//                per fragment row against an operand of ones).  The weight gradient from PRE-SPLIT operands (the planes the
//                layer's own GEMMs write) is usf_wgrad_planes.hip.
//   act_grad     d[m,j] *= (h[m,j] > 0 ? 1 : slope)     LeakyReLU / ReLU backward from the saved OUTPUT h
//                (slope >= 0: sign(h) == sign(pre-activation); ATen leaky_relu_backward uses x > 0 ? 1 : slope)
//   base_grad    g[m,d] = g_lp[m] * d/dz base_d(z[m,d])  Laplace / Normal (torch Laplace.log_prob, Normal.log_prob)
//
// Data gradients (dgrad) are usf_linear_f32 launches with the transposed weight image (usf_pack_weight_f32,
// transpose = 1) -- for the fused coupling layers one launch of the coupling kernel itself on the transposed weight set
// (USF_ACT_GATE, usf_coupling_bf16x3.hip); the parameter-sized chain rule through M^-1 = U^-1 L^-1 is usf_gemm_f64.
#include "usf_common.h"
#include <type_traits>

namespace usf {

constexpr int WG_T = 128;      // output tile (n and k)
constexpr int WG_S = 16;       // batch rows per slab
constexpr int WG_LD = 144;     // LDS row stride (floats): 4 consecutive rows start 16 banks apart

struct WgradArgs {
  const float* Y; int64_t ldy;
  const float* A; int64_t lda;
  float* part;                 // [splits][N][K]
  int M, N, K;
  int rows_per_split;
  float* G; int64_t ldg;       // single row range: the kernel writes alpha * acc + beta * G itself (no reduce launch)
  float alpha, beta;
  int direct;
  int tiles, splits;           // 1-D grid, XCD-aware: see wg_decode
  unsigned long long* dbg;     // tuning builds (-DUSF_STAMP) only
  float* cs_part;              // loader-wave kernel: [splits][N] partial column sums of Y, or NULL (usf_wgrad_bias_f32)
};
#ifdef USF_STAMP
#define WSTAMP() __builtin_amdgcn_s_memtime()
unsigned long long* g_wdbg = nullptr;
#endif

// Block -> (tile, row range).  Every tile column re-reads the Y slab and every tile row the A slab (7 x each at
// 784 x 784), so per launch the blocks ask for 14 x the operand bytes; whether that comes from HBM or from L2 decides
// the kernel (32 flop/byte per block: HBM-bound at ~90 TFLOP/s).  Consecutive block ids are dealt round-robin to the
// 8 XCDs, each with an L2 of its own: all tiles of one row range are therefore given to ONE XCD (ids b, b + 8,
// b + 16, ...), so that XCD's L2 serves the re-reads and HBM sees each operand row once.
__device__ __forceinline__ bool wg_decode(const WgradArgs& a, int& tile, int& split) {
  const int b = blockIdx.x;
  const int xcd = b & 7, q = b >> 3;
  tile = q % a.tiles;
  split = (q / a.tiles) * 8 + xcd;
  return split < a.splits;
}

// 16 x 128 slab of a row-major matrix (rows m0.., columns c0..) -> two float4 per thread, zero outside [M) x [ncols)
__device__ __forceinline__ void wg_load(const float* __restrict__ P, int64_t ld, int m0, int m_end, int c0, int ncols,
                                        int tid, f32x4 (&r)[2]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (tid >> 5) + 8 * i;
    const int col = c0 + (tid & 31) * 4;
    const int m = m0 + row;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (m < m_end) {
      const float* p = P + (int64_t)m * ld + col;
      if (col + 3 < ncols) {
        v = *reinterpret_cast<const f32x4*>(p);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (col + e < ncols) v[e] = p[e];
      }
    }
    r[i] = v;
  }
}

// one 128 x 128 output tile over one row range (the body of wgrad_kernel; grad_jobs_kernel runs it per queued job)
__device__ __forceinline__ void wgrad_tile(const WgradArgs& a, const int tile, const int split) {
  __shared__ __attribute__((aligned(16))) float Ys[2][WG_S][WG_LD];
  __shared__ __attribute__((aligned(16))) float As[2][WG_S][WG_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wk = wave & 1;
  const int tilesK = (a.K + WG_T - 1) / WG_T;
  const int n0 = (tile / tilesK) * WG_T, k0 = (tile % tilesK) * WG_T;
  const int m_begin = split * a.rows_per_split;
  const int m_end = (m_begin + a.rows_per_split < a.M) ? m_begin + a.rows_per_split : a.M;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // number of 16-wide sub-tiles of this wave's 64 x 64 patch that reach into the matrix (wave-uniform)
  const int rem_n = a.N - (n0 + wn * 64), rem_k = a.K - (k0 + wk * 64);
  const int ni = rem_n <= 0 ? 0 : (rem_n >= 64 ? 4 : (rem_n + 15) / 16);
  const int nj = rem_k <= 0 ? 0 : (rem_k >= 64 ? 4 : (rem_k + 15) / 16);
  const bool full = (ni == 4 && nj == 4);
  f32x4 ry[2], ra[2];
  auto stage = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = (tid >> 5) + 8 * i, col = (tid & 31) * 4;
      *reinterpret_cast<f32x4*>(&Ys[buf][row][col]) = ry[i];
      *reinterpret_cast<f32x4*>(&As[buf][row][col]) = ra[i];
    }
  };
  int buf = 0;
  if (m_begin < m_end) {
    wg_load(a.Y, a.ldy, m_begin, m_end, n0, a.N, tid, ry);
    wg_load(a.A, a.lda, m_begin, m_end, k0, a.K, tid, ra);
    stage(0);
  }
  __syncthreads();
  for (int m0 = m_begin; m0 < m_end; m0 += WG_S) {
    const bool more = m0 + WG_S < m_end;
    if (more) {
      wg_load(a.Y, a.ldy, m0 + WG_S, m_end, n0, a.N, tid, ry);
      wg_load(a.A, a.lda, m0 + WG_S, m_end, k0, a.K, tid, ra);
    }
    if (full) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        float fy[4], fa[4];
        const int r = kk * 4 + (lane >> 4);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          fy[t] = Ys[buf][r][wn * 64 + t * 16 + (lane & 15)];
          fa[t] = As[buf][r][wk * 64 + t * 16 + (lane & 15)];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fy[i], fa[j], acc[i][j], 0, 0, 0);
      }
    } else {
      // edge tile (784 = 6 x 128 + 16): only the 16 x 16 sub-tiles that hold real outputs are multiplied
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        float fy[4], fa[4];
        const int r = kk * 4 + (lane >> 4);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          fy[t] = Ys[buf][r][wn * 64 + t * 16 + (lane & 15)];
          fa[t] = As[buf][r][wk * 64 + t * 16 + (lane & 15)];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (i < ni && j < nj) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fy[i], fa[j], acc[i][j], 0, 0, 0);
      }
    }
    if (more) stage(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  // accumulator layout (f32 16x16): row = 4 * (lane >> 4) + reg, col = lane & 15
  float* out = a.part + (int64_t)split * a.N * a.K;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wn * 64 + i * 16 + 4 * (lane >> 4) + r;
        const int k = k0 + wk * 64 + j * 16 + (lane & 15);
        if (n < a.N && k < a.K) {
          if (a.direct) {
            float* dst = a.G + (int64_t)n * a.ldg + k;
            float v = a.alpha * acc[i][j][r];
            if (a.beta != 0.f) v += a.beta * *dst;
            *dst = v;
          } else {
            out[(int64_t)n * a.K + k] = acc[i][j][r];
          }
        }
      }
}

__global__ __launch_bounds__(256) void wgrad_kernel(WgradArgs a) {
  int tile, split;
  if (!wg_decode(a, tile, split)) return;
  wgrad_tile(a, tile, split);
}

// ---------------------------------------------------------------------------------------------------------
// Many small weight / bias gradients in ONE launch (usf_grad_jobs_f32).  At the reference's training batch of 32 rows
// (tests/explib/mnist.yaml:34) a flow of 32 blocks asks for ~130 weight gradients and as many bias gradients of a few
// microseconds each: the step is bound by their dispatch, not by their work.  Every block of this launch looks its job up
// in `block_job` (block -> job index; the job names its first block) and runs
//   A != NULL: one 128 x 128 tile of G = alpha Y^T A + beta G  -- wgrad_tile above, single row range, written directly
//              (bit-identical to usf_wgrad_f32 mode 0 on the same operands at M <= 256),
//   A == NULL: 64 columns of gsum = alpha colsum(Y) + beta gsum (4 row lanes, summed in a fixed order).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void grad_jobs_kernel(const usf_grad_job* __restrict__ jobs, const int32_t* __restrict__ block_job) {
  const usf_grad_job jb = jobs[block_job[blockIdx.x]];
  const int local = (int)blockIdx.x - jb.first_block;
  if (jb.A) {
    const int rows = (jb.M + 31) / 32 * 32;
    const WgradArgs a{jb.Y, jb.ldy, jb.A, jb.lda, nullptr, jb.M, jb.N, jb.K, rows > 0 ? rows : 32, jb.G, jb.ldg, jb.alpha,
                      jb.beta, 1, 0, 1, nullptr, nullptr};
    wgrad_tile(a, local, 0);
    return;
  }
  __shared__ float red[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = local * 64 + cl;
  float s = 0.f;
  if (c < jb.N)
    for (int m = rl; m < jb.M; m += 4) s += jb.Y[(int64_t)m * jb.ldy + c];
  red[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && c < jb.N) {
    const float t = ((red[0][cl] + red[1][cl]) + red[2][cl]) + red[3][cl];
    jb.G[c] = jb.alpha * t + (jb.beta != 0.f ? jb.beta * jb.G[c] : 0.f);
  }
}


// ---------------------------------------------------------------------------------------------------------
// bf16x3 variant of wgrad (mode 1): same tiling, six v_mfma_f32_16x16x32_bf16 per fp32-equivalent product
// (x = x1 + x2 + x3 in bf16, terms below 2^-16 of the leading one dropped, fp32 accumulation).
// The reduction index is the batch row, so an MFMA operand fragment is a COLUMN slice of Y / A: lane (i, g)
// needs rows 8 g .. 8 g + 7 of column i.
// ---------------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
constexpr int WB_S = 32;       // batch rows per slab (one MFMA k-extent)

__device__ __forceinline__ void wb_split(const float (&x)[8], bf16x8_t& p1, bf16x8_t& p2, bf16x8_t& p3) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const __bf16 h = (__bf16)x[j];
    const float r = x[j] - (float)h;           // exact
    const __bf16 m = (__bf16)r;
    const float r2 = r - (float)m;             // exact
    p1[j] = h; p2[j] = m; p3[j] = (__bf16)r2;
  }
}

// Cooperative split: the block turns each 32 x 128 slab of Y and of A into bf16 planes ONCE (thread = 8 rows x 4
// columns: 8 coalesced 16-byte row loads, 4 column slices of 8 rows -> 4 x 3 bf16x8 units) and leaves them in LDS in
// fragment order [plane][k-group g][column] -- a fragment read is then a single conflict-free ds_read_b128 per plane
// instead of 8 ds_read_b32 + a private split per wave (the two waves sharing a fragment used to split it twice).
__global__ __launch_bounds__(256, 2) void wgrad_bf16x3_kernel(WgradArgs a) {
  __shared__ __attribute__((aligned(16))) bf16x8_t Yp[3][4][WG_T];
  __shared__ __attribute__((aligned(16))) bf16x8_t Ap[3][4][WG_T];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int wn = wave >> 1, wk = wave & 1;
  const int tilesK = (a.K + WG_T - 1) / WG_T;
  int tile, split;
  if (!wg_decode(a, tile, split)) return;
  const int n0 = (tile / tilesK) * WG_T, k0 = (tile % tilesK) * WG_T;
  const int m_begin = split * a.rows_per_split;
  const int m_end = (m_begin + a.rows_per_split < a.M) ? m_begin + a.rows_per_split : a.M;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int rem_n = a.N - (n0 + wn * 64), rem_k = a.K - (k0 + wk * 64);
  const int ni = rem_n <= 0 ? 0 : (rem_n >= 64 ? 4 : (rem_n + 15) / 16);
  const int nj = rem_k <= 0 ? 0 : (rem_k >= 64 ? 4 : (rem_k + 15) / 16);

  // loader role: threads 0..127 carry Y, 128..255 carry A; (row group rg = k-group, column group cg)
  const bool isA = tid >= 128;
  const int lt = tid & 127, rg = lt >> 5, cg = lt & 31;
  const float* __restrict__ src = isA ? a.A : a.Y;
  const int64_t ld = isA ? a.lda : a.ldy;
  const int c_base = (isA ? k0 : n0) + 4 * cg;
  const int ncols = isA ? a.K : a.N;
  bf16x8_t (*dstp)[4][WG_T] = isA ? Ap : Yp;
  f32x4 v[8];
  auto fetch = [&](int m0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int m = m0 + 8 * rg + e;
      f32x4 x = {0.f, 0.f, 0.f, 0.f};
      if (m < m_end) {
        const float* p = src + (int64_t)m * ld + c_base;
        if (c_base + 3 < ncols) {
          x = *reinterpret_cast<const f32x4*>(p);
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (c_base + q < ncols) x[q] = p[q];
        }
      }
      v[e] = x;
    }
  };
  auto split_store = [&]() {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float col[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) col[e] = v[e][q];
      bf16x8_t p1, p2, p3;
      wb_split(col, p1, p2, p3);
      dstp[0][rg][4 * cg + q] = p1;
      dstp[1][rg][4 * cg + q] = p2;
      dstp[2][rg][4 * cg + q] = p3;
    }
  };
  // usf_wgrad_bias_f32: the waves of tile column 0 also sum their Y fragments over the batch (three MFMAs per fragment row
  // against an operand of ones: every column of the 16 x 16 result is the column sum)
  const bool do_cs = a.cs_part != nullptr && k0 == 0 && wk == 0;      // wave-uniform
  f32x4 cs[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) cs[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8_t ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
  if (m_begin < m_end) fetch(m_begin);
  for (int m0 = m_begin; m0 < m_end; m0 += WB_S) {
    __syncthreads();                       // the previous slab's fragment reads are done
    split_store();
    __syncthreads();
    if (m0 + WB_S < m_end) fetch(m0 + WB_S);
    bf16x8_t yp[4][3];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) yp[t][pl] = Yp[pl][lg][wn * 64 + t * 16 + li];
    if (do_cs) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (t < ni) {
#pragma unroll
          for (int pl = 2; pl >= 0; --pl) cs[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yp[t][pl], ones, cs[t], 0, 0, 0);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j >= nj) break;
      bf16x8_t ap[3];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) ap[pl] = Ap[pl][lg][wk * 64 + j * 16 + li];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (i < ni) {
#define USF_WB(P, Q) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yp[i][P], ap[Q], acc[i][j], 0, 0, 0)
          USF_WB(2, 0); USF_WB(1, 1); USF_WB(0, 2); USF_WB(1, 0); USF_WB(0, 1); USF_WB(0, 0);   // smallest terms first
#undef USF_WB
        }
    }
  }
  float* out = a.part + (int64_t)split * a.N * a.K;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wn * 64 + i * 16 + 4 * (lane >> 4) + r;
        const int k = k0 + wk * 64 + j * 16 + (lane & 15);
        if (n < a.N && k < a.K) {
          if (a.direct) {
            float* dst = a.G + (int64_t)n * a.ldg + k;
            float vv = a.alpha * acc[i][j][r];
            if (a.beta != 0.f) vv += a.beta * *dst;
            *dst = vv;
          } else {
            out[(int64_t)n * a.K + k] = acc[i][j][r];
          }
        }
      }
  if (do_cs && (lane & 15) == 0) {
    float* co = a.cs_part + (int64_t)split * a.N;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wn * 64 + i * 16 + 4 * (lane >> 4) + r;
        if (n < a.N && i < ni) co[n] = cs[i][r];
      }
  }
}

// ---------------------------------------------------------------------------------------------------------
// bf16x3 wgrad with LOADER WAVES (mode 1 from 2048 rows; round 2).  The kernel above runs fetch -> barrier ->
// split -> barrier -> MFMA in every wave: the matrix pipe idles while the block splits, and the next slab's loads
// have one MFMA phase to come back from HBM.  Here a 768-thread block (one per CU, 128 x 128 tile) has two roles:
//   waves 0..3 (one per SIMD): nothing but ds_read_b128 + MFMA on a 64 x 64 patch (96 MFMAs per 32-row slab); the
//                first fragments of slab s + 1 are read under the last MFMAs of slab s (one set of Y fragment
//                registers, refilled row by row in the slab's last column);
//   waves 4..11 (two per SIMD; 4..7 carry Y, 8..11 carry A): fetch fp32 rows ahead of their use (USF_WL_DEPTH slabs in
//                flight per thread; two measured better than four), split them into bf16 planes and store them in
//                fragment order two slabs ahead into a ring of three images.  (Round 2 first shipped four loader waves
//                with 8 x 4-column units: eight with 8 x 2 measured 1.4 % faster -- the split is bound by vector issue,
//                not by its dependency chains.)
// Measured (tools/exp_wgrad.hip stamps, 784 x 784 x 65 536): an MFMA wave needs ~1600 cycles per slab; with the split
// switched off the kernel runs at 164 TFLOP/s, with it at 115-125 -- the loaders' VALU and the MFMAs of the same SIMD
// largely take turns instead of overlapping (an MFMA holds the SIMD's vector issue for half of its cycles), so the
// split (5.25 instructions per value, every value split by the 7 blocks that share its row slab) is what is left.
// One barrier per slab.  The image is swizzled (unit u of a line sits at u ^ ((u >> 4) & 3)) so that both the
// loaders' stores (lane stride 2 units) and the fragment reads (16 consecutive units) are bank-conflict free.
// Loads are unconditional 8-byte buffer loads: rows beyond M are out of the resource's range and read as zeros
// (row ranges are whole slabs, so inside a range "beyond m_end" means "beyond M"); columns beyond N / K read
// whatever lies there (row padding or the next row) -- they only ever reach output columns that are not stored.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int wl_swz(int c) { return c ^ ((c >> 4) & 3); }
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct WlShared {
  bf16x8_t Yp[3][3][4][WG_T];                 // [ring][plane][8-row group][column]
  bf16x8_t Ap[3][3][4][WG_T];
};

// fp32 -> three bf16 planes for 8 values, two at a time (v_cvt_pk_bf16_f32 / v_pk_add_f32): 5.25 instructions per value
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void wl_split(const float (&x)[8], bf16x8_t& p1, bf16x8_t& p2, bf16x8_t& p3) {
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const f32x2 v = {x[2 * t], x[2 * t + 1]};
    const bf16x2_t h = __builtin_convertvector(v, bf16x2_t);
    const f32x2 r = v - __builtin_convertvector(h, f32x2);            // exact
    const bf16x2_t m = __builtin_convertvector(r, bf16x2_t);
    const f32x2 r2 = r - __builtin_convertvector(m, f32x2);           // exact
    const bf16x2_t l = __builtin_convertvector(r2, bf16x2_t);
    p1[2 * t] = h[0]; p1[2 * t + 1] = h[1];
    p2[2 * t] = m[0]; p2[2 * t + 1] = m[1];
    p3[2 * t] = l[0]; p3[2 * t + 1] = l[1];
  }
}

#ifdef USF_STAMP
#define WL_Q(v) __builtin_amdgcn_sched_barrier(0); const unsigned long long v = WSTAMP()
#else
#define WL_Q(v)
#endif

// the MFMA waves' main loop for a patch of NI x NJ live 16 x 16 sub-tiles (rounded up to a power of two; columns
// beyond N / K produce values that are not stored)
// CS: the wave also sums its Y fragments over the batch (the layer's bias gradient): three more MFMAs per fragment row and
// slab against a B operand of ones -- every column of the 16 x 16 result is the column sum of Y
template <int NI, int NJ, bool CS = false>
__device__ __forceinline__ void wl_mfma_loop(WlShared& sh, f32x4 (&acc)[4][4], int nslab, int nslab4, int wn, int wk, int li, int lg,
                                             unsigned long long* dbg_slot, f32x4 (&cs)[4]) {
  int ycol[NI], acol[NJ];
#pragma unroll
  for (int t = 0; t < NI; ++t) ycol[t] = wl_swz(wn * 64 + t * 16 + li);
#pragma unroll
  for (int t = 0; t < NJ; ++t) acol[t] = wl_swz(wk * 64 + t * 16 + li);
  bf16x8_t yp[NI][3], ap[2][3];
  bf16x8_t ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
  auto read_y1 = [&](int ring, int t) {
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) yp[t][pl] = sh.Yp[ring][pl][lg][ycol[t]];
  };
  auto read_a = [&](int ring, int j, bf16x8_t (&f)[3]) {
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) f[pl] = sh.Ap[ring][pl][lg][acol[j]];
  };
  // B0: which of the two A-fragment registers holds column 0 of this slab (alternates from slab to slab when NJ is odd)
  // (one set of Y fragments: in the slab's last column each row's registers are refilled with the next slab's fragments
  // as soon as the row's products are issued -- 48 registers less than a second set, which is what lets twelve waves share a CU)
  auto slab = [&](int ring, int ring_next, auto b0) {
    constexpr int B0 = decltype(b0)::value;
#define USF_WL(P, Q) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yp[i][P], ap[(j + B0) & 1][Q], acc[i][j], 0, 0, 0)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if (j + 1 < NJ) {
        read_a(ring, j + 1, ap[(j + 1 + B0) & 1]);
      } else {                                  // the next slab's first fragments (its image is complete since the last barrier)
        read_a(ring_next, 0, ap[(j + 1 + B0) & 1]);
      }
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        USF_WL(2, 0); USF_WL(1, 1); USF_WL(0, 2); USF_WL(1, 0); USF_WL(0, 1); USF_WL(0, 0);   // smallest terms first
        if (CS && j == 0) {
#pragma unroll
          for (int pl = 2; pl >= 0; --pl) cs[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yp[i][pl], ones, cs[i], 0, 0, 0);
        }
        if (j + 1 == NJ) read_y1(ring_next, i);
      }
    }
#undef USF_WL
    // per A fragment: the next fragment's reads, then its MFMAs (the last column also carries the next slab's Y reads)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      constexpr int XC = CS ? 3 : 0;            // column 0's extra MFMAs per fragment row
      __builtin_amdgcn_sched_group_barrier(0x008, 2 * NI, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
      if (j + 1 == NJ) {
#pragma unroll
        for (int t = 0; t < NI; ++t) {
          if (j == 0) __builtin_amdgcn_sched_group_barrier(0x008, 4 + XC, 0);
          else __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
        }
      } else {
        if (j == 0) __builtin_amdgcn_sched_group_barrier(0x008, (4 + XC) * NI, 0);
        else __builtin_amdgcn_sched_group_barrier(0x008, 4 * NI, 0);
      }
    }
  };
  typedef std::integral_constant<int, 0> C0;
  typedef std::integral_constant<int, NJ & 1> C1;      // after an odd number of columns the roles of ap[0] / ap[1] swap
#pragma unroll
  for (int t = 0; t < NI; ++t) read_y1(0, t);
  read_a(0, 0, ap[0]);
#ifdef USF_STAMP
  unsigned long long tw = 0, tb = 0;
#endif
  int s = 0, ring = 0;
  auto nxt = [](int r) { return r == 2 ? 0 : r + 1; };
  for (; s < nslab4; s += 2) {
    WL_Q(q0);
    if (s < nslab) slab(ring, nxt(ring), C0());
    WL_Q(q1);
    __syncthreads();
    WL_Q(q2);
    ring = nxt(ring);
    if (s + 1 < nslab) slab(ring, nxt(ring), C1());
    WL_Q(q3);
    __syncthreads();
    ring = nxt(ring);
#ifdef USF_STAMP
    tw += (q1 - q0) + (q3 - q2); tb += (q2 - q1) + (WSTAMP() - q3);
#endif
  }
#ifdef USF_STAMP
  if (dbg_slot) { dbg_slot[0] = tw; dbg_slot[1] = tb; dbg_slot[2] = nslab; dbg_slot[3] = 1; }
#endif
}

__global__ __launch_bounds__(768) void wgrad_lw_kernel(WgradArgs a) {
  __shared__ __attribute__((aligned(16))) WlShared sh;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tilesK = (a.K + WG_T - 1) / WG_T;
  int tile, split;
  if (!wg_decode(a, tile, split)) return;
  const int n0 = (tile / tilesK) * WG_T, k0 = (tile % tilesK) * WG_T;
  const int m_begin = split * a.rows_per_split;
  const int m_end = (m_begin + a.rows_per_split < a.M) ? m_begin + a.rows_per_split : a.M;
  const int nslab = (m_end - m_begin + WB_S - 1) / WB_S;
  const int nslab4 = (nslab + 3) & ~3;        // iterations every wave runs (barrier count): see the loader loop
  unsigned long long* dbg_slot = nullptr;
#ifdef USF_STAMP
  if (a.dbg && lane == 0) dbg_slot = a.dbg + (size_t)((blockIdx.x % 1024) * 12 + wave) * 4;
#endif

  if (wave >= 4) {
    // ------------------------------- loader waves -------------------------------
    // thread = one (8 rows x 2 columns) unit of the slab: waves 4 .. 7 carry Y, waves 8 .. 11 carry A -- two loader waves per
    // SIMD beside its MFMA wave: the split is a chain of dependent conversions, one wave alone runs it at its latency
    const int lt = tid - 256;
    const bool isA = wave >= 8;                 // wave-uniform (the buffer resource must sit in scalar registers)
    const int u = lt & 255, rg = u >> 6, cg = u & 63;
    const unsigned ld = (unsigned)(isA ? a.lda : a.ldy);
    const unsigned c0 = (unsigned)((isA ? k0 : n0) + 2 * cg);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(isA ? a.A : a.Y), 0, (int)((((unsigned)a.M - 1u) * ld + (unsigned)(isA ? a.K : a.N)) * 4u), 0x00020000);
    // (the row within the 8-row unit goes into the instruction's scalar offset: one vector add per slab, not per load)
    const unsigned vo = (((unsigned)m_begin + 8u * rg) * ld + c0) * 4u;
    auto fetch = [&](int sl, f32x2 (&v)[8]) {
      const unsigned o = vo + (unsigned)sl * (WB_S * 4u) * ld;
#ifdef USF_WL_X_NOLOAD
      if (sl > 8) return;
#endif
#pragma unroll
      for (int e = 0; e < 8; ++e)
        v[e] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)o, (int)((unsigned)e * ld * 4u), 0));
    };
    auto split_store = [&](int ring, const f32x2 (&v)[8]) {
      bf16x8_t* d = isA ? &sh.Ap[ring][0][rg][0] : &sh.Yp[ring][0][rg][0];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        float col[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) col[e] = v[e][q];
        bf16x8_t p1, p2, p3;
#ifdef USF_WL_X_NOSPLIT
        p1 = __builtin_bit_cast(bf16x8_t, (f32x4){v[q][0], v[q][1], v[q + 2][0], v[q + 2][1]}); p2 = p1; p3 = p1;
#else
        wl_split(col, p1, p2, p3);
#endif
        const int us = wl_swz(2 * cg + q);
        d[us] = p1; d[4 * WG_T + us] = p2; d[8 * WG_T + us] = p3;
      }
    };
#ifndef USF_WL_DEPTH
#define USF_WL_DEPTH 2
#endif
    constexpr int DEPTH = USF_WL_DEPTH;         // slabs in flight per thread: slab t lives in set t % DEPTH from fetch to split
    f32x2 vs[DEPTH][8];
#pragma unroll
    for (int t = 0; t < DEPTH; ++t) fetch(t, vs[t]);
    split_store(0, vs[0]); fetch(DEPTH, vs[0]);
    split_store(1, vs[1]); fetch(DEPTH + 1, vs[1]);
    __syncthreads();
    // iteration s: slab s is multiplied out of ring s % 3 while slab s + 2 is split into ring (s + 2) % 3 and slab
    // s + 2 + DEPTH is fetched
#ifdef USF_STAMP
    unsigned long long tw = 0, tb = 0;
#endif
    // (straight-line groups of DEPTH iterations, no switch on s: the compiler's vmcnt bookkeeping stays exact, so a
    // wait for slab s + 2 does not also drain the younger loads of the slabs behind it; every wave of the block runs
    // the slab count rounded up to a multiple of four -- the surplus iterations only keep the barriers)
    int ring2 = 2;                              // (s + 2) % 3
    for (int s0 = 0; s0 < nslab4; s0 += DEPTH) {
#pragma unroll
      for (int k = 0; k < DEPTH; ++k) {
        WL_Q(q0);
        split_store(ring2, vs[(k + 2) % DEPTH]);
        fetch(s0 + k + 2 + DEPTH, vs[(k + 2) % DEPTH]);
        WL_Q(q1);
        __syncthreads();
        ring2 = ring2 == 2 ? 0 : ring2 + 1;
#ifdef USF_STAMP
        tw += q1 - q0; tb += WSTAMP() - q1;
#endif
      }
    }
#ifdef USF_STAMP
    if (dbg_slot) { dbg_slot[0] = tw; dbg_slot[1] = tb; dbg_slot[2] = nslab; dbg_slot[3] = 1; }
#endif
    return;
  }

  // ------------------------------- MFMA waves -------------------------------
  const int li = lane & 15, lg = lane >> 4;
  const int wn = wave >> 1, wk = wave & 1;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int rem_n = a.N - (n0 + wn * 64), rem_k = a.K - (k0 + wk * 64);
  const int ni = rem_n <= 0 ? 0 : (rem_n >= 64 ? 4 : (rem_n + 15) / 16);
  const int nj = rem_k <= 0 ? 0 : (rem_k >= 64 ? 4 : (rem_k + 15) / 16);
  const bool do_cs = a.cs_part != nullptr && k0 == 0 && wk == 0;      // wave-uniform
  f32x4 cs[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) cs[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  // wave-uniform choice of the patch size: live sub-tiles rounded up to {1, 2, 4} x {1, 2, 4}; a wave whose patch lies
  // outside the matrix only keeps the barriers
  if (ni == 0 || nj == 0) {
    for (int s = 0; s < nslab4; ++s) __syncthreads();
  } else {
#define WL_GO(NI_, NJ_) wl_mfma_loop<NI_, NJ_>(sh, acc, nslab, nslab4, wn, wk, li, lg, dbg_slot, cs)
#define WL_ROW(NI_) do { if (nj > 2) WL_GO(NI_, 4); else if (nj > 1) WL_GO(NI_, 2); else WL_GO(NI_, 1); } while (0)
#define WL_CS(NI_) wl_mfma_loop<NI_, 4, true>(sh, acc, nslab, nslab4, wn, wk, li, lg, dbg_slot, cs)
    if (do_cs && nj == 4) {                     // (the host asks for column sums only where K >= 64: nj == 4 in wave column 0)
      if (ni > 2) WL_CS(4); else if (ni > 1) WL_CS(2); else WL_CS(1);
    } else if (ni > 2) WL_ROW(4); else if (ni > 1) WL_ROW(2); else WL_ROW(1);
#undef WL_CS
#undef WL_ROW
#undef WL_GO
  }
  if (do_cs && (lane & 15) == 0) {
    float* co = a.cs_part + (int64_t)split * a.N;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wn * 64 + i * 16 + 4 * (lane >> 4) + r;
        if (n < a.N && i < ni) co[n] = cs[i][r];
      }
  }
  float* out = a.part + (int64_t)split * a.N * a.K;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wn * 64 + i * 16 + 4 * (lane >> 4) + r;
        const int k = k0 + wk * 64 + j * 16 + (lane & 15);
        if (n < a.N && k < a.K) {
          if (a.direct) {
            float* dst = a.G + (int64_t)n * a.ldg + k;
            float vv = a.alpha * acc[i][j][r];
            if (a.beta != 0.f) vv += a.beta * *dst;
            *dst = vv;
          } else {
            out[(int64_t)n * a.K + k] = acc[i][j][r];
          }
        }
      }
}
#undef WL_Q

// out[r*ldo + c] = alpha * sum_s part[s][r][c] + beta * out[...]   (rows x cols elements per partial)
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ part, int splits, int64_t rows,
                                                              int64_t cols, float* __restrict__ out, int64_t ldo,
                                                              float alpha, float beta, const float* __restrict__ cs_part = nullptr,
                                                              float* __restrict__ cs_out = nullptr, float cs_alpha = 0.f,
                                                              float cs_beta = 0.f) {
  const int64_t total = rows * cols;
  if (cs_out)                                   // the column sums of Y (usf_wgrad_bias_f32): a loop of its own, the main one as it was
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < rows; r += (int64_t)gridDim.x * 256) {
      float s = 0.f;
      // (16 loads in flight, added in order: one wave walks all the row ranges here -- a load per iteration would cost its latency each time)
#pragma unroll 16
      for (int p = 0; p < splits; ++p) s += cs_part[(int64_t)p * rows + r];
      float v = cs_alpha * s;
      if (cs_beta != 0.f) v += cs_beta * cs_out[r];
      cs_out[r] = v;
    }
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    float s = 0.f;
    for (int p = 0; p < splits; ++p) s += part[(int64_t)p * total + e];
    const int64_t r = e / cols, c = e - r * cols;
    float v = alpha * s;
    if (beta != 0.f) v += beta * out[r * ldo + c];
    out[r * ldo + c] = v;
  }
}

// column sums over a range of 256 rows: block = 64 columns x 16 row lanes (coalesced 256-byte row segments).
// final == 0: part[split][n] = sum;  final != 0 (single split): out[n] = alpha * sum + beta * out[n]
__global__ __launch_bounds__(1024) void colsum_kernel(const float* __restrict__ Y, int64_t ldy, int M, int N,
                                                      float* __restrict__ dst, int final, float alpha, float beta) {
  __shared__ float red[16][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const int m_begin = blockIdx.y * 256;
  const int m_end = (m_begin + 256 < M) ? m_begin + 256 : M;
  float s = 0.f;
  if (c < N)
#pragma unroll 4
    for (int m = m_begin + rl; m < m_end; m += 16) s += Y[(int64_t)m * ldy + c];
  red[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && c < N) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) t += red[r][cl];
    if (final) dst[c] = alpha * t + (beta != 0.f ? beta * dst[c] : 0.f);
    else dst[(int64_t)blockIdx.y * N + c] = t;
  }
}

// Gradients of sum_m g_lp[m] * sum_d base_d(z[m,d]; loc_d, scale_d) w.r.t. loc and scale, over a range of 256 rows (colsum_kernel's
// block shape): part[split][0][d] = sum_m g_lp[m] * d/dloc,  part[split][1][d] = sum_m g_lp[m] * d/dscale
//   Laplace (torch.distributions.Laplace.log_prob): d/dloc = sign(t) / b,  d/db = |t| / b^2 - 1 / b      (t = z - loc)
//   Normal:                                         d/dloc = t / s^2,      d/ds = t^2 / s^3 - 1 / s
__global__ __launch_bounds__(1024) void base_param_grad_kernel(const float* __restrict__ z, int64_t ldz, const float* __restrict__ g_lp,
                                                               int M, int D, int base, const float* __restrict__ loc,
                                                               const float* __restrict__ scale, float* __restrict__ part) {
  __shared__ float red[2][16][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const int m_begin = blockIdx.y * 256;
  const int m_end = (m_begin + 256 < M) ? m_begin + 256 : M;
  float s0 = 0.f, s1 = 0.f;
  if (c < D) {
    const float mu = loc[c], b = scale[c];
    const float ib = 1.0f / b;
#pragma unroll 4
    for (int m = m_begin + rl; m < m_end; m += 16) {
      const float t = z[(int64_t)m * ldz + c] - mu;
      const float w = g_lp[m];
      if (base == USF_BASE_LAPLACE) {
        const float sg = (float)((t > 0.f) - (t < 0.f));
        s0 += w * (sg * ib);
        s1 += w * (fabsf(t) * ib * ib - ib);
      } else {
        s0 += w * (t * ib * ib);
        s1 += w * (t * t * ib * ib * ib - ib);
      }
    }
  }
  red[0][rl][cl] = s0;
  red[1][rl][cl] = s1;
  __syncthreads();
  if (rl < 2 && c < D) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) t += red[rl][r][cl];
    part[((int64_t)blockIdx.y * 2 + rl) * D + c] = t;
  }
}

__global__ __launch_bounds__(256) void act_grad_kernel(float* __restrict__ d, int64_t ldd, const float* __restrict__ h,
                                                       int64_t ldh, int64_t M, int64_t H, float slope) {
  const int64_t total = M * H;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t m = e / H, j = e - m * H;
    const float hv = h[m * ldh + j];
    if (!(hv > 0.f)) d[m * ldd + j] *= slope;
  }
}

// g[m,d] = g_lp[m] * d/dz base_d(z[m,d]);  columns D..ldg-1 (layout padding) are written as zeros.
// LPNORM* (RadialDistribution, distributions.py:501-505): g_lp is the gradient at the radius r[m] = ||z - loc||_p and
// `scale` carries r [M]:  p = 1: sign(t),  p = 2: t / r,  p = inf: sign(t) where |t| == r (ATen's norm backward).
__global__ __launch_bounds__(256) void base_grad_kernel(const float* __restrict__ z, int64_t ldz, const float* __restrict__ g_lp,
                                                        int64_t M, int64_t D, int base, const float* __restrict__ loc,
                                                        const float* __restrict__ scale, float* __restrict__ g,
                                                        int64_t ldg) {
  const int64_t total = M * ldg;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t m = e / ldg, d = e - m * ldg;
    float v = 0.f;
    if (d < D) {
      const float t = z[m * ldz + d] - loc[d];
      const float sg = (float)((t > 0.f) - (t < 0.f));
      if (base == USF_BASE_LAPLACE) v = -sg / scale[d];                           // d/dz -|z-loc|/b ; 0 at the kink (ATen)
      else if (base == USF_BASE_NORMAL) v = -t / (scale[d] * scale[d]);           // d/dz -(z-loc)^2 / (2 s^2)
      else if (base == USF_BASE_LPNORM1) v = sg;
      else if (base == USF_BASE_LPNORM2) v = scale[m] > 0.f ? t / scale[m] : 0.f;
      else v = (fabsf(t) == scale[m]) ? sg : 0.f;
      v *= g_lp[m];
    }
    g[e] = v;
  }
}

// ---------------------------------------------------------------------------------------------------------
static int pick_splits(int64_t M, int64_t tiles, bool lw) {
  // row ranges of at least 256 rows.  256-thread kernels (two blocks per CU): enough to give every CU ~2 blocks.
  // Loader-wave kernel (one 512-thread block per CU): ~3 rounds of blocks, so that the short edge tiles (784 = 6 x 128
  // + 16) and the full ones even out over the CUs, at >= 64 slabs per block.
  int lw_blocks = (int)tuning("wgrad_blocks", 768);
  if (lw_blocks < 1) lw_blocks = 768;
  const int64_t target = lw ? lw_blocks : 640;
  int64_t s = (target + tiles - 1) / tiles;
  const int64_t smax = lw ? (M + 1023) / 1024 : (M + 255) / 256;
  if (s > smax) s = smax;
  if (s < 1) s = 1;
  if (s > 256) s = 256;
  return (int)s;
}
// The loader-wave kernel: from the measured cross-over against the 256-thread kernel (at least 8192 rows and enough
// work to fill its one block per CU; 4096 x 784 x 784 and 8192 x 256 x 392 are still faster on the old one), and
// while its 32-bit byte offsets hold (prefetch distance included).  wgrad_lw_min: tuning knob.
static bool use_lw(int64_t M, int32_t mode, int64_t tiles, int64_t ldy, int64_t lda) {
  const int64_t lw_min = tuning("wgrad_lw_min", 8192);
  return mode == 1 && M >= lw_min && M * tiles >= 160000 && (M + 448) * (ldy > lda ? ldy : lda) * 4 < (1LL << 32);
}

static int64_t wg_tiles(int64_t N, int64_t K) { return ((N + WG_T - 1) / WG_T) * ((K + WG_T - 1) / WG_T); }

int wgrad_workspace_floats(int64_t M, int64_t N, int64_t K, int64_t* out) {
  if (M < 0 || N <= 0 || K <= 0) return -1;
  const int sa = pick_splits(M, wg_tiles(N, K), false), sb = pick_splits(M, wg_tiles(N, K), true);      // either kernel may be chosen
  *out = (int64_t)(sa > sb ? sa : sb) * N * (K + 1);            // + the partial column sums of usf_wgrad_bias_f32
  return 0;
}

// which kernel usf_wgrad_f32 launches: 0 exact-f32 MFMA, 1 bf16x3 (256 threads), 2 bf16x3 with loader waves
int wgrad_variant(int64_t M, int64_t N, int64_t K, int64_t ldy, int64_t lda, int32_t mode) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const int old_kernel = (int)tuning("wgrad_old", 0);      // tuning aid: 1 keeps the round-1 bf16x3 kernel
  if (use_lw(M, mode, wg_tiles(N, K), ldy, lda) && !old_kernel) return 2;
  return (mode == 1 && M >= 2048) ? 1 : 0;
}

// usf_wgrad_bias_f32: where a bf16x3 kernel runs (mode 1 from 2048 rows) and K >= 64 the column sums of Y ride along
int wgrad_bias_ok(int64_t M, int64_t N, int64_t K, int64_t ldy, int64_t lda, int32_t mode) {
  return (wgrad_variant(M, N, K, ldy, lda, mode) >= 1 && K >= 64) ? 1 : 0;     // either bf16x3 kernel
}

int wgrad(const float* Y, int64_t ldy, const float* A, int64_t lda, int64_t M, int64_t N, int64_t K, float* G,
          int64_t ldg, float alpha, float beta, int32_t mode, float* workspace, int64_t workspace_floats,
          hipStream_t stream, float* colsum_out, float cs_alpha, float cs_beta) {
  if (colsum_out && !wgrad_bias_ok(M, N, K, ldy, lda, mode)) {
    set_error("usf_wgrad_bias_f32: the column sums need a bf16x3 kernel (mode 1, M >= 2048) and K >= 64 (usf_wgrad_bias_ok)");
    return -2;
  }
  if (((!Y || !A) && M > 0) || !G || !workspace || M < 0 || N <= 0 || K <= 0 || ldg < K || ldy < N || lda < K) {
    set_error("usf_wgrad_f32: bad arguments");
    return -1;
  }
  if (M > 0x7fffffff || N > (1 << 20) || K > (1 << 20)) { set_error("usf_wgrad_f32: size out of range"); return -2; }
  if ((ldy & 3) || (lda & 3) || !aligned16(Y) || !aligned16(A)) {
    set_error("usf_wgrad_f32: Y / A need 16-byte aligned rows (ld %% 4 == 0, aligned base)");
    return -3;
  }
  if (mode != 0 && mode != 1) { set_error("usf_wgrad_f32: mode must be 0 (exact f32) or 1 (bf16x3)"); return -2; }
  // the split-precision kernels pay off once the chip has real work (the operand split costs VALU per slab)
  const int64_t tiles = wg_tiles(N, K);
  const int variant = wgrad_variant(M, N, K, ldy, lda, mode);
  const bool lw = variant == 2;
  const int splits = pick_splits(M, tiles, lw);
  if (workspace_floats < (int64_t)splits * N * (K + (colsum_out ? 1 : 0))) {
    set_error("usf_wgrad_f32: workspace too small (%lld < %lld floats)", (long long)workspace_floats,
              (long long)splits * N * (K + 1));
    return -4;
  }
  int rows = (int)((M + splits - 1) / splits);
  rows = (rows + WB_S - 1) / WB_S * WB_S;
  WgradArgs a{Y, ldy, A, lda, workspace, (int)M, (int)N, (int)K, rows > 0 ? rows : WB_S, G, ldg, alpha, beta,
              (splits == 1 && !colsum_out) ? 1 : 0, (int)tiles, splits, nullptr, nullptr};
  if (colsum_out) a.cs_part = workspace + (int64_t)splits * N * K;
#ifdef USF_STAMP
  a.dbg = g_wdbg;
#endif
  const unsigned grid = (unsigned)(tiles * ((splits + 7) / 8) * 8);
  if (lw) wgrad_lw_kernel<<<grid, 768, 0, stream>>>(a);
  else if (variant == 1) wgrad_bf16x3_kernel<<<grid, 256, 0, stream>>>(a);
  else wgrad_kernel<<<grid, 256, 0, stream>>>(a);
  if (splits > 1 || colsum_out) {
    int64_t rb = (N * K + 255) / 256;
    if (rb > 4096) rb = 4096;
    reduce_partials_kernel<<<(unsigned)rb, 256, 0, stream>>>(workspace, splits, N, K, G, ldg, alpha, beta, a.cs_part, colsum_out,
                                                              cs_alpha, cs_beta);
  }
  return check_launch("usf_wgrad_f32");
}

int colsum(const float* Y, int64_t ldy, int64_t M, int64_t N, float* out, float alpha, float beta, float* workspace,
           int64_t workspace_floats, hipStream_t stream) {
  if ((!Y && M > 0) || !out || !workspace || M < 0 || N <= 0 || ldy < N || M > 0x7fffffff) {
    set_error("usf_colsum_f32: bad arguments");
    return -1;
  }
  // levels of 256-row ranges: M -> ceil(M/256) partial rows -> ... -> 1 (fixed order: reproducible)
  const float* src = Y;
  int64_t rows = M, ld = ldy, used = 0;
  const unsigned gx = (unsigned)((N + 63) / 64);
  while (true) {
    const int64_t splits = rows > 0 ? (rows + 255) / 256 : 1;
    if (splits == 1) {
      colsum_kernel<<<dim3(gx, 1), 1024, 0, stream>>>(src, ld, (int)rows, (int)N, out, 1, alpha, beta);
      break;
    }
    if (used + splits * N > workspace_floats) { set_error("usf_colsum_f32: workspace too small"); return -4; }
    float* part = workspace + used;
    colsum_kernel<<<dim3(gx, (unsigned)splits), 1024, 0, stream>>>(src, ld, (int)rows, (int)N, part, 0, 1.f, 0.f);
    used += splits * N;
    src = part;
    rows = splits;
    ld = N;
  }
  return check_launch("usf_colsum_f32");
}

// usf_base_param_grad_f32: see include/usflows_hip.h
int base_param_grad(const float* z, int64_t ldz, const float* g_lp, int64_t M, int64_t D, int32_t base, const float* loc,
                    const float* scale, float* d_loc_scale, float* workspace, int64_t workspace_floats, hipStream_t stream) {
  if (M < 0 || D <= 0 || ldz < D || (M > 0 && (!z || !g_lp)) || !loc || !scale || !d_loc_scale || M > 0x7fffffff || D > 0x3fffffff) {
    set_error("usf_base_param_grad_f32: bad arguments");
    return -1;
  }
  if (base != USF_BASE_LAPLACE && base != USF_BASE_NORMAL) { set_error("usf_base_param_grad_f32: Laplace / Normal only"); return -2; }
  const int64_t splits = M > 0 ? (M + 255) / 256 : 0;
  if (splits == 0) return hipMemsetAsync(d_loc_scale, 0, (size_t)(2 * D) * sizeof(float), stream) == hipSuccess ? 0 : -5;
  if (!workspace || splits * 2 * D > workspace_floats) { set_error("usf_base_param_grad_f32: workspace too small"); return -4; }
  base_param_grad_kernel<<<dim3((unsigned)((D + 63) / 64), (unsigned)splits), 1024, 0, stream>>>(z, ldz, g_lp, (int)M, (int)D, base, loc,
                                                                                              scale, workspace);
  // the partials' rows [split][2 D] summed in colsum's fixed order (bit-reproducible)
  return colsum(workspace, 2 * D, splits, 2 * D, d_loc_scale, 1.f, 0.f, workspace + splits * 2 * D, workspace_floats - splits * 2 * D, stream);
}

int grad_jobs(const usf_grad_job* jobs, const int32_t* block_job, int64_t n_blocks, hipStream_t stream) {
  if (n_blocks < 0 || n_blocks > 0x7fffffff || (n_blocks > 0 && (!jobs || !block_job))) {
    set_error("usf_grad_jobs_f32: bad arguments");
    return -1;
  }
  if (n_blocks == 0) return 0;
  grad_jobs_kernel<<<(unsigned)n_blocks, 256, 0, stream>>>(jobs, block_job);
  return check_launch("usf_grad_jobs_f32");
}

// ---------------------------------------------------------------------------------------------------------
// SophiaG (sophia.py:151-199: _single_tensor_sophiag, and update_hessian sophia.py:39-58) over ALL parameter tensors
// of a model in one launch: a table of chunks (one block each) replaces the reference's per-tensor loop of seven
// elementwise ATen ops.  HBM-bound: 24 bytes per parameter and step (read p, g, m, h; write p, m).
//   step:     p *= decay;  m = m * beta1 + g' * (1 - beta1);  ratio = min(|m| / (rho_bs * h + 1e-15), 1);
//             p += neg_lr * sign(m) * ratio            (g' = -g with maximize; the operation order of the reference)
//   hessian:  h = h * beta2 + (1 - beta2) * g * g
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sophiag_step_kernel(const usf_mt_chunk* __restrict__ chunks, float decay, float beta1,
                                                           float omb1, float rho_bs, float neg_lr, int maximize) {
  const usf_mt_chunk c = chunks[blockIdx.x];
  for (int i = threadIdx.x; i < c.n; i += 256) {
    float g = c.g[i];
    if (maximize) g = -g;
    const float p = c.p[i] * decay;
    const float m = fmaf(g, omb1, c.m[i] * beta1);        // exp_avg.mul_(beta1).add_(grad, alpha): ATen's add-with-alpha is a fused multiply-add
    const float ratio = fminf(fabsf(m) / (rho_bs * c.h[i] + 1e-15f), 1.f);
    const float sg = (float)((m > 0.f) - (m < 0.f));
    c.m[i] = m;
    c.p[i] = p + neg_lr * (sg * ratio);
  }
}

__global__ __launch_bounds__(256) void sophiag_hessian_kernel(const usf_mt_chunk* __restrict__ chunks, float beta2, float omb2) {
  const usf_mt_chunk c = chunks[blockIdx.x];
  for (int i = threadIdx.x; i < c.n; i += 256) {
    const float g = c.g[i];
    c.h[i] = c.h[i] * beta2 + omb2 * (g * g);
  }
}

int sophiag_step(const usf_mt_chunk* chunks, int64_t n_chunks, float decay, float beta1, float one_minus_beta1, float rho_bs,
                 float neg_lr, int32_t maximize, hipStream_t stream) {
  if (n_chunks < 0 || n_chunks > 0x7fffffff || (!chunks && n_chunks > 0)) { set_error("usf_sophiag_step_f32: bad arguments"); return -1; }
  if (n_chunks == 0) return 0;
  sophiag_step_kernel<<<(unsigned)n_chunks, 256, 0, stream>>>(chunks, decay, beta1, one_minus_beta1, rho_bs, neg_lr, maximize);
  return check_launch("usf_sophiag_step_f32");
}

int sophiag_hessian(const usf_mt_chunk* chunks, int64_t n_chunks, float beta2, float one_minus_beta2, hipStream_t stream) {
  if (n_chunks < 0 || n_chunks > 0x7fffffff || (!chunks && n_chunks > 0)) { set_error("usf_sophiag_hessian_f32: bad arguments"); return -1; }
  if (n_chunks == 0) return 0;
  sophiag_hessian_kernel<<<(unsigned)n_chunks, 256, 0, stream>>>(chunks, beta2, one_minus_beta2);
  return check_launch("usf_sophiag_hessian_f32");
}

int act_grad(float* d, int64_t ldd, const float* h, int64_t ldh, int64_t M, int64_t H, int32_t act, float slope,
             hipStream_t stream) {
  if (!d || !h || M < 0 || H < 0 || ldd < H || ldh < H) { set_error("usf_act_grad_f32: bad arguments"); return -1; }
  if (act == USF_ACT_NONE || M == 0 || H == 0) return 0;
  if (act != USF_ACT_LEAKY_RELU || slope < 0.f) { set_error("usf_act_grad_f32: unsupported activation"); return -2; }
  int64_t blocks = (M * H + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  act_grad_kernel<<<(unsigned)blocks, 256, 0, stream>>>(d, ldd, h, ldh, M, H, slope);
  return check_launch("usf_act_grad_f32");
}

int base_grad(const float* z, int64_t ldz, const float* g_lp, int64_t M, int64_t D, int32_t base, const float* loc,
              const float* scale, float* g, int64_t ldg, hipStream_t stream) {
  if (!z || !g_lp || !loc || !scale || !g || M < 0 || D <= 0 || ldz < D || ldg < D) {
    set_error("usf_base_logprob_grad_f32: bad arguments");
    return -1;
  }
  if (base < USF_BASE_LAPLACE || base > USF_BASE_LPNORMINF) {
    set_error("usf_base_logprob_grad_f32: unknown base %d", base);
    return -2;
  }
  if (M == 0) return 0;
  int64_t blocks = (M * ldg + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  base_grad_kernel<<<(unsigned)blocks, 256, 0, stream>>>(z, ldz, g_lp, M, D, base, loc, scale, g, ldg);
  return check_launch("usf_base_logprob_grad_f32");
}

}  // namespace usf
