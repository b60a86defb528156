// Backward-pass kernels of the training step: what autograd derives from Flow.log_prob
// (flows.py:196-199: loss = -log_prob(batch).mean() - log_prior) for the layers of the hot path.
//
//   wgrad        G[n,k] = sum_m Y[m,n] * A[m,k]        weight gradient of every F.linear on the path
//                (BlockAffineTransform transforms.py:913-962, conditioner Linear layers networks.py:739-751):
//                a GEMM whose reduction runs over the BATCH, exact-f32 MFMA (v_mfma_f32_16x16x4_f32), split over
//                row ranges into partials + a deterministic reduction (no atomics: bitwise reproducible gradients)
//   colsum       g_bias[n] = sum_m Y[m,n]               bias gradient, same two-stage scheme
//   act_grad     d[m,j] *= (h[m,j] > 0 ? 1 : slope)     LeakyReLU / ReLU backward from the saved OUTPUT h
//                (slope >= 0: sign(h) == sign(pre-activation); ATen leaky_relu_backward uses x > 0 ? 1 : slope)
//   base_grad    g[m,d] = g_lp[m] * d/dz base_d(z[m,d])  Laplace / Normal (torch Laplace.log_prob, Normal.log_prob)
//
// Data gradients (dgrad) are usf_linear_f32 launches with the transposed weight image (usf_pack_weight_f32,
// transpose = 1); the parameter-sized chain rule through M^-1 = U^-1 L^-1 is usf_gemm_f64.
#include "usf_common.h"
#include <stdlib.h>
#include <string.h>
#include <type_traits>

namespace usf {

constexpr int WG_T = 128;      // output tile (n and k)
constexpr int WG_S = 16;       // batch rows per slab
constexpr int WG_LD = 144;     // LDS row stride (floats): 4 consecutive rows start 16 banks apart

// ---- balanced schedule of the loader-wave kernels (round 4) ----------------------------------------------------------
// 784 = 6 x 128 + 16: 13 of the 49 output tiles of a 784 x 784 weight gradient are edge tiles whose blocks take a
// fraction of a full block's time, and 49 tiles x 16 row ranges = 784 blocks are 3.06 rounds on 256 CUs -- with the plain
// (tile, row range) grid the last blocks run beside idle CUs (measured makespan 3.7 full-block times against 2.6 of
// work per CU).  Here the grid is ONE block per CU and every block gets one item of (nearly) equal duration: the tiles
// fall into four classes (full, edge in k, edge in n, corner), a tile of class c is cut into nseg[c] row ranges with
// nseg proportional to the class's cost per slab (greedy min-max on the host), items are numbered class by class and,
// inside a class, row range by row range (so the 32 blocks of an XCD -- consecutive items -- work on the same rows and
// share them in its L2).  Partial s of a tile lives in part[s]; the reduction knows how many each class wrote.
struct WgSched {
  int items;                   // 0: the plain grid (wg_decode)
  int per_xcd;                 // blocks per XCD (grid / 8)
  int fN, fK;                  // full 128-wide tile rows / columns
  int T[4];                    // tiles per class: full, edge in k, edge in n, corner
  int off[5];                  // first item of each class
  int nseg[4], rows[4];        // row ranges per tile, rows per range (multiples of 32)
  short start[4][8], cnt[4][8]; // items of class c on XCD x: start[c][x] .. + cnt[c][x] (numbered inside the class)
};

__device__ __forceinline__ bool wg_item(const WgSched& sc, int M, int& n0, int& k0, int& m_begin, int& m_end, int& seg) {
  // XCD x takes an eighth of EVERY class's items (an XCD with nothing but edge tiles, which share no operand columns
  // with each other, is bound by its own memory-side bandwidth: measured 0.75 of a full tile's time per slab)
  const int b = blockIdx.x;
  const int x = b & 7;
  int slot = b >> 3, cls = 0, le = -1;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int n = sc.cnt[c][x];
    if (le < 0) {
      if (slot < n) { le = sc.start[c][x] + slot; cls = c; }
      else slot -= n;
    }
  }
  if (le < 0) return false;
  seg = le / sc.T[cls];
  const int t = le - seg * sc.T[cls];
  int tn, tk;
  if (cls == 0) { tn = t / sc.fK; tk = t - tn * sc.fK; }
  else if (cls == 1) { tn = t; tk = sc.fK; }
  else if (cls == 2) { tn = sc.fN; tk = t; }
  else { tn = sc.fN; tk = sc.fK; }
  n0 = tn * WG_T; k0 = tk * WG_T;
  m_begin = seg * sc.rows[cls];
  m_end = (m_begin + sc.rows[cls] < M) ? m_begin + sc.rows[cls] : M;
  return m_begin < m_end;
}

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
  WgSched sched;               // loader-wave kernel only
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
                      jb.beta, 1, 0, 1, nullptr};
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
template <int NI, int NJ>
__device__ __forceinline__ void wl_mfma_loop(WlShared& sh, f32x4 (&acc)[4][4], int nslab, int nslab4, int wn, int wk, int li, int lg,
                                             unsigned long long* dbg_slot) {
  int ycol[NI], acol[NJ];
#pragma unroll
  for (int t = 0; t < NI; ++t) ycol[t] = wl_swz(wn * 64 + t * 16 + li);
#pragma unroll
  for (int t = 0; t < NJ; ++t) acol[t] = wl_swz(wk * 64 + t * 16 + li);
  bf16x8_t yp[NI][3], ap[2][3];
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
        if (j + 1 == NJ) read_y1(ring_next, i);
      }
    }
#undef USF_WL
    // per A fragment: the next fragment's reads, then its MFMAs (the last column also carries the next slab's Y reads)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      __builtin_amdgcn_sched_group_barrier(0x008, 2 * NI, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
      if (j + 1 == NJ) {
#pragma unroll
        for (int t = 0; t < NI; ++t) {
          __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
        }
      } else {
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * NI, 0);
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
  int n0, k0, m_begin, m_end, split;
  if (a.sched.items) {
    if (!wg_item(a.sched, a.M, n0, k0, m_begin, m_end, split)) return;
  } else {
    int tile;
    if (!wg_decode(a, tile, split)) return;
    n0 = (tile / tilesK) * WG_T; k0 = (tile % tilesK) * WG_T;
    m_begin = split * a.rows_per_split;
    m_end = (m_begin + a.rows_per_split < a.M) ? m_begin + a.rows_per_split : a.M;
  }
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
  __syncthreads();
  // wave-uniform choice of the patch size: live sub-tiles rounded up to {1, 2, 4} x {1, 2, 4}; a wave whose patch lies
  // outside the matrix only keeps the barriers
  if (ni == 0 || nj == 0) {
    for (int s = 0; s < nslab4; ++s) __syncthreads();
  } else {
#define WL_GO(NI_, NJ_) wl_mfma_loop<NI_, NJ_>(sh, acc, nslab, nslab4, wn, wk, li, lg, dbg_slot)
#define WL_ROW(NI_) do { if (nj > 2) WL_GO(NI_, 4); else if (nj > 1) WL_GO(NI_, 2); else WL_GO(NI_, 1); } while (0)
    if (ni > 2) WL_ROW(4); else if (ni > 1) WL_ROW(2); else WL_ROW(1);
#undef WL_ROW
#undef WL_GO
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

// ---------------------------------------------------------------------------------------------------------
// bf16x3 wgrad from PRE-SPLIT operands (round 4): Y and A arrive as three row-major bf16 planes each (the side output
// of the GEMM that consumed them -- usf_linear_f32's a_planes_out --, or usf_split_planes_f32).  Same block shape, ring,
// barrier scheme, product order and output path as wgrad_lw_kernel (bit-identical gradients), but
//   * the loader waves only COPY: six 16-byte buffer loads and six ds_write_b128 per thread and slab -- no conversion
//     (the operand split the seven blocks of a row slab each repeated, 5.25 vector instructions per value, was what
//     bound wgrad_lw_kernel: r02_tuning_experiments.md section 4);
//   * the LDS image keeps the rows as they are in HBM: [32 batch rows][128 columns] bf16 per plane, 16-byte chunk ch of
//     row r at position ch ^ (((r & 3) << 2) | ((r >> 2) & 3)), and the MFMA waves read their operand fragments (8
//     consecutive batch rows of one column per lane) with the transposing read ds_read_b64_tr_b16, two per fragment.
// Rows [M, ceil32(M)) of the planes must be zero (both producers write them so); columns beyond N / K read whatever
// follows (padding, the next row, zeros beyond the buffer) and only reach outputs that are not stored.
// ---------------------------------------------------------------------------------------------------------
struct WpArgs {
  const __bf16* Yp; int64_t ldyp, ystride;     // planes [3][rows][ld]: plane stride in elements
  const __bf16* Ap; int64_t ldap, astride;
  int y_col0, a_col0;                          // first column of the operands inside their planes (multiples of 8)
  unsigned ybytes, abytes;                     // readable bytes from Yp / Ap (buffer bounds: beyond reads as zeros)
  float* part;
  int M, N, K;
  int rows_per_split;
  float* G; int64_t ldg;
  float alpha, beta;
  int direct;
  int tiles, splits;
  unsigned long long* dbg;                     // tuning builds (-DUSF_STAMP) only
  WgSched sched;
  int trim;                                    // both operands' planes end below 2 GiB: dead chunks of edge tiles are skipped
};

struct WpShared {
  uint4 img[3][2][3][512];                     // [ring][Y / A][plane][32 rows x 16 chunks, swizzled]
};

typedef short wp_s16x4 __attribute__((ext_vector_type(4)));
typedef short wp_s16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ int wp_swz(int r) { return ((r & 3) << 2) | ((r >> 2) & 3); }
__device__ __forceinline__ bf16x8_t wp_frag(const char* tile, int o0, int o1) {
  typedef __attribute__((address_space(3))) wp_s16x4 lds_v;
  const wp_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v*)(tile + o0));
  const wp_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v*)(tile + o1));
  const wp_s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}

template <int NI, int NJ>
__device__ __forceinline__ void wp_mfma_loop(WpShared& sh, f32x4 (&acc)[4][4], int nslab, int nslab4, int wn, int wk, int lane,
                                             unsigned long long* dbg_slot) {
  // lane 16 g + 4 q + p supplies row 8 g + 4 hh + q, columns 16 t + 4 p .. + 3 of the sub-tile (hh = 0, 1: the two reads)
  const int lg = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
  int yo[NI][2], ao[NJ][2];
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) {
    const int r = 8 * lg + 4 * hh + q;
#pragma unroll
    for (int t = 0; t < NI; ++t) yo[t][hh] = 256 * r + 16 * ((8 * wn + 2 * t + (pp >> 1)) ^ wp_swz(r)) + 8 * (pp & 1);
#pragma unroll
    for (int t = 0; t < NJ; ++t) ao[t][hh] = 256 * r + 16 * ((8 * wk + 2 * t + (pp >> 1)) ^ wp_swz(r)) + 8 * (pp & 1);
  }
  bf16x8_t yp[NI][3], ap[2][3];
  auto read_y1 = [&](int ring, int t) {
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) yp[t][pl] = wp_frag(reinterpret_cast<const char*>(&sh.img[ring][0][pl][0]), yo[t][0], yo[t][1]);
  };
  auto read_a = [&](int ring, int j, bf16x8_t (&f)[3]) {
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) f[pl] = wp_frag(reinterpret_cast<const char*>(&sh.img[ring][1][pl][0]), ao[j][0], ao[j][1]);
  };
  auto slab = [&](int ring, int ring_next, auto b0) {
    constexpr int B0 = decltype(b0)::value;
#define USF_WP(P, Q) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yp[i][P], ap[(j + B0) & 1][Q], acc[i][j], 0, 0, 0)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if (j + 1 < NJ) {
        read_a(ring, j + 1, ap[(j + 1 + B0) & 1]);
      } else {
        read_a(ring_next, 0, ap[(j + 1 + B0) & 1]);
      }
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        USF_WP(2, 0); USF_WP(1, 1); USF_WP(0, 2); USF_WP(1, 0); USF_WP(0, 1); USF_WP(0, 0);   // the order of wgrad_lw_kernel
        if (j + 1 == NJ) read_y1(ring_next, i);
      }
    }
#undef USF_WP
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      __builtin_amdgcn_sched_group_barrier(0x008, 2 * NI, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
      if (j + 1 == NJ) {
#pragma unroll
        for (int t = 0; t < NI; ++t) {
          __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
        }
      } else {
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * NI, 0);
      }
    }
  };
  typedef std::integral_constant<int, 0> C0;
  typedef std::integral_constant<int, NJ & 1> C1;
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

__global__ __launch_bounds__(768) void wgrad_planes_kernel(WpArgs a) {
  __shared__ __attribute__((aligned(16))) WpShared sh;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tilesK = (a.K + WG_T - 1) / WG_T;
  int n0, k0, m_begin, m_end, split;
  if (a.sched.items) {
    if (!wg_item(a.sched, a.M, n0, k0, m_begin, m_end, split)) return;
  } else {
    const int b = blockIdx.x;
    const int tile = (b >> 3) % a.tiles;                                                 // wg_decode's map
    split = ((b >> 3) / a.tiles) * 8 + (b & 7);
    if (split >= a.splits) return;
    n0 = (tile / tilesK) * WG_T; k0 = (tile % tilesK) * WG_T;
    m_begin = split * a.rows_per_split;
    m_end = (m_begin + a.rows_per_split < a.M) ? m_begin + a.rows_per_split : a.M;
  }
  const int nslab = (m_end - m_begin + WB_S - 1) / WB_S;
  const int nslab4 = (nslab + 3) & ~3;
  unsigned long long* dbg_slot = nullptr;
#ifdef USF_STAMP
  if (a.dbg && lane == 0) dbg_slot = a.dbg + (size_t)((blockIdx.x % 1024) * 12 + wave) * 4;
#endif

  if (wave >= 4) {
    // ------------------------------- loader waves: waves 4 .. 7 copy Y, waves 8 .. 11 copy A -------------------------------
    const bool isA = wave >= 8;                 // wave-uniform
    const int u = (tid - 256) & 255;
    const unsigned ld = (unsigned)(isA ? a.ldap : a.ldyp);
    const unsigned pstride = (unsigned)(isA ? a.astride : a.ystride) * 2u;           // bytes
    const unsigned c0 = (unsigned)(isA ? k0 + a.a_col0 : n0 + a.y_col0);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<__bf16*>(isA ? a.Ap : a.Yp), 0, (int)(isA ? a.abytes : a.ybytes), 0x00020000);
    // chunks beyond the tile's live columns (edge tiles: 784 = 6 x 128 + 16) are not fetched: their lanes ask for an offset
    // beyond the buffer (answered with zeros, no memory access); the MFMA waves never read those columns
    const int live = (isA ? a.K - k0 : a.N - n0);
    const int live_ch = a.trim ? 2 * ((live + 15) >> 4) : 16;
    unsigned vo[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int c = u + 256 * h, r = c >> 4, ch = (c & 15) ^ wp_swz(r);
      vo[h] = ch < live_ch ? (((unsigned)m_begin + (unsigned)r) * ld + c0 + 8u * (unsigned)ch) * 2u : 0x80000000u;
    }
    typedef unsigned wp_u32x4 __attribute__((ext_vector_type(4)));
    auto fetch = [&](int sl, wp_u32x4 (&v)[6]) {
      const unsigned adv = (unsigned)sl * (WB_S * 2u) * ld;
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int h = 0; h < 2; ++h)
          v[2 * pl + h] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(vo[h] + adv), (int)((unsigned)pl * pstride), 0);
    };
    auto store = [&](int ring, const wp_u32x4 (&v)[6]) {
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int h = 0; h < 2; ++h)
          sh.img[ring][isA ? 1 : 0][pl][u + 256 * h] = __builtin_bit_cast(uint4, v[2 * pl + h]);
    };
#ifndef USF_WP_DEPTH
#define USF_WP_DEPTH 2
#endif
    constexpr int DEPTH = USF_WP_DEPTH;         // slabs in flight per thread (2 or 4: the slab count is rounded up to a multiple of four)
    wp_u32x4 vs[DEPTH][6];
#pragma unroll
    for (int t = 0; t < DEPTH; ++t) fetch(t, vs[t]);
    store(0, vs[0]); fetch(DEPTH, vs[0]);
    store(1, vs[1 % DEPTH]); fetch(DEPTH + 1, vs[1 % DEPTH]);
    __syncthreads();
    int ring2 = 2;
#ifdef USF_STAMP
    unsigned long long tw = 0, tb = 0;
#endif
    for (int s0 = 0; s0 < nslab4; s0 += DEPTH) {
#pragma unroll
      for (int k = 0; k < DEPTH; ++k) {
        WL_Q(q0);
        store(ring2, vs[(k + 2) % DEPTH]);
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
  const int wn = wave >> 1, wk = wave & 1;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int rem_n = a.N - (n0 + wn * 64), rem_k = a.K - (k0 + wk * 64);
  const int ni = rem_n <= 0 ? 0 : (rem_n >= 64 ? 4 : (rem_n + 15) / 16);
  const int nj = rem_k <= 0 ? 0 : (rem_k >= 64 ? 4 : (rem_k + 15) / 16);
  __syncthreads();
  if (ni == 0 || nj == 0) {
    for (int s = 0; s < nslab4; ++s) __syncthreads();
  } else {
#define WP_GO(NI_, NJ_) wp_mfma_loop<NI_, NJ_>(sh, acc, nslab, nslab4, wn, wk, lane, dbg_slot)
#define WP_ROW(NI_) do { if (nj > 2) WP_GO(NI_, 4); else if (nj > 1) WP_GO(NI_, 2); else WP_GO(NI_, 1); } while (0)
    if (ni > 2) WP_ROW(4); else if (ni > 1) WP_ROW(2); else WP_ROW(1);
#undef WP_ROW
#undef WP_GO
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

// fp32 rows -> three row-major bf16 planes (round-to-nearest residual split, x = p1 + p2 + p3 exactly): P[pl][m][c] for
// m < rows_pad, c < ldp; zeros for m >= M or c >= N.  One thread = 8 columns of a row.
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ X, int64_t ldx, int M, int N, __bf16* __restrict__ P,
                                                           int64_t ldp, int64_t pstride, int rows_pad, int vec) {
  const int cpr = (int)(ldp >> 3);
  const int64_t total = (int64_t)rows_pad * cpr;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int m = (int)(e / cpr), c = (int)(e - (int64_t)m * cpr) * 8;
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = 0.f;
    if (m < M) {
      if (c + 8 <= N && vec) {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(X + (int64_t)m * ldx + c), v1 = *reinterpret_cast<const f32x4*>(X + (int64_t)m * ldx + c + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { x[j] = v0[j]; x[4 + j] = v1[j]; }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) if (c + j < N) x[j] = X[(int64_t)m * ldx + c + j];
      }
    }
    bf16x8_t p1, p2, p3;
    wl_split(x, p1, p2, p3);
    __bf16* d = P + (int64_t)m * ldp + c;
    *reinterpret_cast<bf16x8_t*>(d) = p1;
    *reinterpret_cast<bf16x8_t*>(d + pstride) = p2;
    *reinterpret_cast<bf16x8_t*>(d + 2 * pstride) = p3;
  }
}

// out[r*ldo + c] = alpha * sum_s part[s][r][c] + beta * out[...]   (rows x cols elements per partial)
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ part, int splits, int64_t rows,
                                                              int64_t cols, float* __restrict__ out, int64_t ldo,
                                                              float alpha, float beta) {
  const int64_t total = rows * cols;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    float s = 0.f;
    for (int p = 0; p < splits; ++p) s += part[(int64_t)p * total + e];
    const int64_t r = e / cols, c = e - r * cols;
    float v = alpha * s;
    if (beta != 0.f) v += beta * out[r * ldo + c];
    out[r * ldo + c] = v;
  }
}

// the same for the balanced schedule: element (r, c) sums the nseg[class of its tile] partials its tile wrote
__global__ __launch_bounds__(256) void reduce_partials_cls_kernel(const float* __restrict__ part, WgSched sc, int64_t rows, int64_t cols,
                                                                  float* __restrict__ out, int64_t ldo, float alpha, float beta) {
  const int64_t total = rows * cols;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t r = e / cols, c = e - r * cols;
    const int cls = (r >= (int64_t)sc.fN * WG_T ? 2 : 0) + (c >= (int64_t)sc.fK * WG_T ? 1 : 0);
    const int n = sc.nseg[cls];
    float s = 0.f;
    for (int p = 0; p < n; ++p) s += part[(int64_t)p * total + e];
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
  static int lw_blocks = -1;
  if (lw_blocks < 0) { const char* e = getenv("USF_WGRAD_BLOCKS"); lw_blocks = e ? atoi(e) : 768; if (lw_blocks < 1) lw_blocks = 768; }
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
// while its 32-bit byte offsets hold (prefetch distance included).  USF_WGRAD_LW_MIN: tuning aid.
static bool use_lw(int64_t M, int32_t mode, int64_t tiles, int64_t ldy, int64_t lda) {
  static int64_t lw_min = -1;
  if (lw_min < 0) { const char* e = getenv("USF_WGRAD_LW_MIN"); lw_min = e ? atoll(e) : 8192; }
  return mode == 1 && M >= lw_min && M * tiles >= 160000 && (M + 448) * (ldy > lda ? ldy : lda) * 4 < (1LL << 32);
}

static int64_t wg_tiles(int64_t N, int64_t K) { return ((N + WG_T - 1) / WG_T) * ((K + WG_T - 1) / WG_T); }

// One item per CU (see WgSched).  edge_cost: time of an edge tile's slab relative to a full tile's, in percent.
static int wg_cus() {
  static int n = -1;
  if (n < 0) {
    int dev = 0; hipDeviceProp_t pr;
    n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ? pr.multiProcessorCount : 256;
    const char* e = getenv("USF_WGRAD_CUS");
    if (e && atoi(e) > 0) n = atoi(e);
    n &= ~7;
    if (n < 8) n = 8;
  }
  return n;
}
static bool wg_schedule(int64_t M, int64_t N, int64_t K, int edge_pct, int corner_pct, WgSched& sc) {
  memset(&sc, 0, sizeof(sc));
  const int cus = wg_cus();
  const int S = (int)((M + WB_S - 1) / WB_S);                  // slabs
  sc.per_xcd = cus / 8;
  sc.fN = (int)(N / WG_T); sc.fK = (int)(K / WG_T);
  const int eN = (N % WG_T) ? 1 : 0, eK = (K % WG_T) ? 1 : 0;
  sc.T[0] = sc.fN * sc.fK; sc.T[1] = sc.fN * eK; sc.T[2] = eN * sc.fK; sc.T[3] = eN * eK;
  const int cost[4] = {100, edge_pct, edge_pct, corner_pct};
  int tiles = 0;
  for (int c = 0; c < 4; ++c) { sc.nseg[c] = sc.T[c] ? 1 : 0; tiles += sc.T[c]; }
  if (tiles > cus || S < 4) return false;
  int used = tiles;
  while (true) {                                               // greedy min-max: one more row range for the class whose items run longest
    int best = -1; int64_t bt = -1;
    for (int c = 0; c < 4; ++c) {
      if (!sc.T[c] || sc.nseg[c] >= S || sc.nseg[c] >= 256) continue;
      const int64_t t = (int64_t)((S + sc.nseg[c] - 1) / sc.nseg[c]) * cost[c];
      if (t > bt) { bt = t; best = c; }
    }
    if (best < 0 || used + sc.T[best] > cus) break;
    ++sc.nseg[best]; used += sc.T[best];
  }
  sc.items = 0;
  for (int c = 0; c < 4; ++c) {
    sc.off[c] = sc.items;
    if (sc.T[c]) {
      const int slabs = (S + sc.nseg[c] - 1) / sc.nseg[c];
      sc.rows[c] = slabs * WB_S;
      sc.nseg[c] = (S + slabs - 1) / slabs;
    }
    sc.items += sc.T[c] * sc.nseg[c];
  }
  sc.off[4] = sc.items;
  // class by class, an eighth to every XCD; the remainders go to the XCDs with the fewest items so far
  int load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int c = 0; c < 4; ++c) {
    const int n = sc.T[c] * sc.nseg[c];
    int cn[8];
    for (int x = 0; x < 8; ++x) cn[x] = n / 8;
    for (int r = 0; r < n % 8; ++r) {
      int bx = 0;
      for (int x = 1; x < 8; ++x) if (load[x] + cn[x] < load[bx] + cn[bx]) bx = x;
      ++cn[bx];
    }
    int st = 0;
    for (int x = 0; x < 8; ++x) { sc.start[c][x] = (short)st; sc.cnt[c][x] = (short)cn[x]; st += cn[x]; load[x] += cn[x]; }
  }
  for (int x = 0; x < 8; ++x) if (load[x] > sc.per_xcd) return false;
  return true;
}
static int wg_sched_max_parts(const WgSched& sc) {
  int m = 1;
  for (int c = 0; c < 4; ++c) if (sc.nseg[c] > m) m = sc.nseg[c];
  return m;
}
static int wg_env_pct(const char* name, int dflt) {
  const char* e = getenv(name);
  return (e && atoi(e) > 0) ? atoi(e) : dflt;
}

int wgrad_workspace_floats(int64_t M, int64_t N, int64_t K, int64_t* out) {
  if (M < 0 || N <= 0 || K <= 0) return -1;
  const int sa = pick_splits(M, wg_tiles(N, K), false), sb = pick_splits(M, wg_tiles(N, K), true);      // either kernel may be chosen
  int sc_parts = 0;
  WgSched sc;
  if (wg_schedule(M, N, K, 20, 20, sc)) sc_parts = wg_sched_max_parts(sc);    // the cheapest edge cost any kernel uses: most row ranges
  int m = sa > sb ? sa : sb;
  if (sc_parts > m) m = sc_parts;
  *out = (int64_t)m * N * K;
  return 0;
}

// which kernel usf_wgrad_f32 launches: 0 exact-f32 MFMA, 1 bf16x3 (256 threads), 2 bf16x3 with loader waves
int wgrad_variant(int64_t M, int64_t N, int64_t K, int64_t ldy, int64_t lda, int32_t mode) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  static int old_kernel = -1;                 // tuning aid: USF_WGRAD_OLD=1 keeps the round-1 bf16x3 kernel
  if (old_kernel < 0) { const char* e = getenv("USF_WGRAD_OLD"); old_kernel = e ? atoi(e) : 0; }
  if (use_lw(M, mode, wg_tiles(N, K), ldy, lda) && !old_kernel) return 2;
  return (mode == 1 && M >= 2048) ? 1 : 0;
}

int wgrad(const float* Y, int64_t ldy, const float* A, int64_t lda, int64_t M, int64_t N, int64_t K, float* G,
          int64_t ldg, float alpha, float beta, int32_t mode, float* workspace, int64_t workspace_floats,
          hipStream_t stream) {
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
  if (workspace_floats < (int64_t)splits * N * K) {
    set_error("usf_wgrad_f32: workspace too small (%lld < %lld floats)", (long long)workspace_floats,
              (long long)splits * N * K);
    return -4;
  }
  int rows = (int)((M + splits - 1) / splits);
  rows = (rows + WB_S - 1) / WB_S * WB_S;
  WgradArgs a{Y, ldy, A, lda, workspace, (int)M, (int)N, (int)K, rows > 0 ? rows : WB_S, G, ldg, alpha, beta,
              splits == 1 ? 1 : 0, (int)tiles, splits, nullptr};
  memset(&a.sched, 0, sizeof(a.sched));
#ifdef USF_STAMP
  a.dbg = g_wdbg;
#endif
  static int use_sched = -1;                  // USF_WGRAD_SCHED=0: the plain (tile, row range) grid (tuning aid)
  if (use_sched < 0) { const char* e = getenv("USF_WGRAD_SCHED_F32"); use_sched = e ? atoi(e) : 0; }
  if (lw && use_sched && wg_schedule(M, N, K, wg_env_pct("USF_WGRAD_EDGE_PCT", 40), wg_env_pct("USF_WGRAD_CORNER_PCT", 35), a.sched) &&
      (int64_t)wg_sched_max_parts(a.sched) * N * K <= workspace_floats) {
    a.direct = 0;
    wgrad_lw_kernel<<<(unsigned)(a.sched.per_xcd * 8), 768, 0, stream>>>(a);
    int64_t rb = (N * K + 255) / 256;
    if (rb > 4096) rb = 4096;
    reduce_partials_cls_kernel<<<(unsigned)rb, 256, 0, stream>>>(workspace, a.sched, N, K, G, ldg, alpha, beta);
    return check_launch("usf_wgrad_f32");
  }
  a.sched.items = 0;
  const unsigned grid = (unsigned)(tiles * ((splits + 7) / 8) * 8);
  if (lw) wgrad_lw_kernel<<<grid, 768, 0, stream>>>(a);
  else if (variant == 1) wgrad_bf16x3_kernel<<<grid, 256, 0, stream>>>(a);
  else wgrad_kernel<<<grid, 256, 0, stream>>>(a);
  if (splits > 1) {
    int64_t rb = (N * K + 255) / 256;
    if (rb > 4096) rb = 4096;
    reduce_partials_kernel<<<(unsigned)rb, 256, 0, stream>>>(workspace, splits, N, K, G, ldg, alpha, beta);
  }
  return check_launch("usf_wgrad_f32");
}

// usf_split_planes_f32 / usf_wgrad_planes_f32: see include/usflows_hip.h
int split_planes(const float* X, int64_t ldx, int64_t M, int64_t N, void* P, int64_t ldp, int64_t plane_stride, hipStream_t stream) {
  if ((!X && M > 0) || !P || M < 0 || N <= 0 || ldx < N || ldp < N || (ldp & 7) || !aligned16(P) || (plane_stride & 7)) {
    set_error("usf_split_planes_f32: bad arguments (ldp and plane_stride multiples of 8, 16-byte aligned planes)");
    return -1;
  }
  const int64_t rows_pad = (M + WB_S - 1) / WB_S * WB_S;
  if (M > 0x7fffffff - WB_S || plane_stride < rows_pad * ldp) { set_error("usf_split_planes_f32: plane_stride < ceil32(M) * ldp"); return -2; }
  if (rows_pad == 0) return 0;
  int64_t nb = (rows_pad * (ldp >> 3) + 255) / 256;
  if (nb > 65536) nb = 65536;
  split_planes_kernel<<<(unsigned)nb, 256, 0, stream>>>(X, ldx, (int)M, (int)N, (__bf16*)P, ldp, plane_stride, (int)rows_pad,
                                                      (aligned16(X) && !(ldx & 3)) ? 1 : 0);
  return check_launch("usf_split_planes_f32");
}

int64_t wgrad_planes_workspace_floats(int64_t M, int64_t N, int64_t K) {
  if (M < 0 || N <= 0 || K <= 0) return -1;
  int m = pick_splits(M, wg_tiles(N, K), true);
  WgSched sc;
  if (wg_schedule(M, N, K, 20, 20, sc) && wg_sched_max_parts(sc) > m) m = wg_sched_max_parts(sc);
  return (int64_t)m * N * K;
}

int wgrad_planes_ok(int64_t M, int64_t N, int64_t K) {
  // the copying kernel pays where the loader-wave kernel is chosen (its cross-over, the same block shape)
  return M >= 8192 && M * wg_tiles(N, K) >= 160000;
}

int wgrad_planes(const void* Yp, int64_t ldyp, int64_t ystride, int64_t y_off, const void* Ap, int64_t ldap, int64_t astride,
                 int64_t a_off, int64_t M, int64_t N, int64_t K, float* G, int64_t ldg, float alpha, float beta, float* workspace,
                 int64_t workspace_floats, hipStream_t stream) {
  if (!Yp || !Ap || !G || !workspace || M <= 0 || N <= 0 || K <= 0 || ldg < K || y_off < 0 || a_off < 0 || ldyp < y_off + N ||
      ldap < a_off + K) {
    set_error("usf_wgrad_planes_f32: bad arguments");
    return -1;
  }
  if ((ldyp & 7) || (ldap & 7) || (y_off & 7) || (a_off & 7) || (ystride & 7) || (astride & 7) || !aligned16(Yp) || !aligned16(Ap)) {
    set_error("usf_wgrad_planes_f32: row strides, plane strides and column offsets must be multiples of 8 elements, planes 16-byte aligned");
    return -3;
  }
  const int64_t rows_pad = (M + WB_S - 1) / WB_S * WB_S;
  if (ystride < rows_pad * ldyp || astride < rows_pad * ldap) { set_error("usf_wgrad_planes_f32: plane stride < ceil32(M) * ld"); return -2; }
  const int64_t yb = (2 * ystride + rows_pad * ldyp) * 2, ab = (2 * astride + rows_pad * ldap) * 2;
  if (M > 0x7fffffff - 4096 || N > (1 << 20) || K > (1 << 20) || yb >= (1LL << 32) || ab >= (1LL << 32)) {
    set_error("usf_wgrad_planes_f32: size out of range (planes of one operand must stay below 4 GiB)");
    return -2;
  }
  const int64_t tiles = wg_tiles(N, K);
  const int splits = pick_splits(M, tiles, true);
  int rows = (int)((M + splits - 1) / splits);
  rows = (rows + WB_S - 1) / WB_S * WB_S;
  WpArgs a{(const __bf16*)Yp, ldyp, ystride, (const __bf16*)Ap, ldap, astride, (int)y_off, (int)a_off, (unsigned)yb, (unsigned)ab,
           workspace, (int)M, (int)N, (int)K, rows > 0 ? rows : WB_S, G, ldg, alpha, beta, splits == 1 ? 1 : 0, (int)tiles, splits, nullptr};
  memset(&a.sched, 0, sizeof(a.sched));
  a.trim = (yb < (1LL << 31) && ab < (1LL << 31) && wg_env_pct("USF_WGRADP_TRIM", 1) == 1) ? 1 : 0;
#ifdef USF_STAMP
  a.dbg = g_wdbg;
#endif
  static int use_sched = -1;
  if (use_sched < 0) { const char* e = getenv("USF_WGRAD_SCHED"); use_sched = e ? atoi(e) : 1; }
  if (use_sched && wg_schedule(M, N, K, wg_env_pct("USF_WGRADP_EDGE_PCT", 100), wg_env_pct("USF_WGRADP_CORNER_PCT", 100), a.sched) &&
      (int64_t)wg_sched_max_parts(a.sched) * N * K <= workspace_floats) {
    a.direct = 0;
    wgrad_planes_kernel<<<(unsigned)(a.sched.per_xcd * 8), 768, 0, stream>>>(a);
    int64_t rb = (N * K + 255) / 256;
    if (rb > 4096) rb = 4096;
    reduce_partials_cls_kernel<<<(unsigned)rb, 256, 0, stream>>>(workspace, a.sched, N, K, G, ldg, alpha, beta);
    return check_launch("usf_wgrad_planes_f32");
  }
  a.sched.items = 0;
  if (workspace_floats < (int64_t)splits * N * K) {
    set_error("usf_wgrad_planes_f32: workspace too small (%lld < %lld floats)", (long long)workspace_floats, (long long)splits * N * K);
    return -4;
  }
  const unsigned grid = (unsigned)(tiles * ((splits + 7) / 8) * 8);
  wgrad_planes_kernel<<<grid, 768, 0, stream>>>(a);
  if (splits > 1) {
    int64_t rb = (N * K + 255) / 256;
    if (rb > 4096) rb = 4096;
    reduce_partials_kernel<<<(unsigned)rb, 256, 0, stream>>>(workspace, splits, N, K, G, ldg, alpha, beta);
  }
  return check_launch("usf_wgrad_planes_f32");
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
