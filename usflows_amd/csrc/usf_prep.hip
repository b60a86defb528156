// Parameter prep on the device: everything the reference re-derives from the raw
// parameters on every call with ATen CPU ops --
//   LUTransform.L / .U / .matrix / .inverse_matrix / .log_abs_det_jacobian   transforms.py:1271-1320
//   HouseholderTransform._construct_householder_permutation                  transforms.py:795-809
//   SequentialAffineTransform.matrix / .inverse_matrix / .bias               transforms.py:1457-1476
// -- as a handful of batched fp64 launches over ALL affine blocks of a flow at once:
//
//   lu_unpack      raw fp32 [D,D] x 2n  ->  fp64 triangles  T[2i] = tril(L_raw,-1)+I,  T[2i+1] = triu(U_raw)^T
//                  (U is kept transposed so that all 2n matrices are lower-triangular: one code path)
//   tri_diag_inv   exact substitution on the 32x32 diagonal blocks
//   tri_level<1|2> recursive doubling: for [A 0; C B], X = -B^-1 (C A^-1); one pair of launches per level,
//                  all pairs of all 2n matrices in one grid
//   gemm_f64       M = L (U^T)^T (k <= min(i,j)),  M^-1 = ((U^T)^-1)^T L^-1 (k >= max(i,j)); also the generic
//                  batched C = alpha op(A) op(B) + beta C behind usf_gemm_f64 (Sequential composition, the
//                  matrix gradients of the training path)
//   householder    one wave per row of w_0: row <- row - 2 (row.v) v^T / (v.v), all nvs reflections in one launch
//   pack_weight    fp64 -> fp32 with the engine's row/column permutation + zero padding, and the three bf16
//                  planes of the bf16x3 kernels, in one pass
//   fold_bias      c = -(Minv b) in fp64 (bias folding)
//
// All matrix products run on v_mfma_f64_16x16x4_f64 (78.6 TFLOP/s dense peak on MI355X); the work is tiny
// (cfg2: ~40 GFLOP for 33 blocks), so the kernel is a plain LDS-tiled 64x64 design -- what matters is that the
// whole prep is ~25 launches instead of ~1000 host-bound ATen calls.
#include "usf_common.h"

typedef double f64x4 __attribute__((ext_vector_type(4)));

namespace usf {

constexpr int NB = 32;         // diagonal block of the triangular inverse
constexpr int PTRS_PER_LAUNCH = 48;

struct PtrTab {
  const float* L[PTRS_PER_LAUNCH];
  const float* U[PTRS_PER_LAUNCH];
};

// ---------------------------------------------------------------------------------------------------------
// unpack: grid (blocks, matrices of this chunk)
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lu_unpack_kernel(PtrTab t, int64_t first, int64_t D, double* __restrict__ tri) {
  const float* __restrict__ L = t.L[blockIdx.y];
  const float* __restrict__ U = t.U[blockIdx.y];
  double* outL = tri + 2 * (first + blockIdx.y) * D * D;
  double* outU = outL + D * D;
  const int64_t total = D * D;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t i = e / D, j = e - i * D;
    outL[e] = (j < i) ? (double)L[e] : (j == i ? 1.0 : 0.0);          // transforms.py:1271-1274
    outU[e] = (j <= i) ? (double)U[j * D + i] : 0.0;                   // (triu(U_raw))^T, :1276-1279
  }
}

// sum log|diag U| per matrix (transforms.py:1303-1320); one wave per matrix
__global__ __launch_bounds__(64) void lu_ladj_kernel(PtrTab t, int64_t first, int64_t D, double* __restrict__ ladj) {
  const float* __restrict__ U = t.U[blockIdx.x];
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < D; i += 64) s += log(fabs((double)U[i * D + i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (threadIdx.x == 0) ladj[first + blockIdx.x] = s;
}

// ---------------------------------------------------------------------------------------------------------
// inverse of the NB x NB diagonal blocks of lower-triangular matrices; grid (blocks per matrix, matrices)
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void tri_diag_inv_kernel(const double* __restrict__ tri, double* __restrict__ inv,
                                                          int64_t D) {
  __shared__ double T[NB][NB + 1];
  __shared__ double X[NB][NB + 1];
  const int lane = threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.x * NB;
  const int nb = (int)((D - r0) < NB ? (D - r0) : NB);
  const double* src = tri + (int64_t)blockIdx.y * D * D;
  double* dst = inv + (int64_t)blockIdx.y * D * D;
  for (int e = lane; e < NB * NB; e += 64) {
    const int i = e / NB, j = e % NB;
    T[i][j] = (i < nb && j < nb) ? src[(r0 + i) * D + r0 + j] : (i == j ? 1.0 : 0.0);
    X[i][j] = 0.0;
  }
  __syncthreads();
  if (lane < NB) {                       // lane j: column j of the inverse by forward substitution
    const int j = lane;
    X[j][j] = 1.0 / T[j][j];
    for (int i = j + 1; i < NB; ++i) {
      double s = 0.0;
      for (int k = j; k < i; ++k) s += T[i][k] * X[k][j];
      X[i][j] = -s / T[i][i];
    }
  }
  __syncthreads();
  for (int e = lane; e < NB * NB; e += 64) {
    const int i = e / NB, j = e % NB;
    if (i < nb && j < nb) dst[(r0 + i) * D + r0 + j] = X[i][j];
  }
}

// ---------------------------------------------------------------------------------------------------------
// 64x64 output tile of C = alpha op(A) op(B) + beta C on the f64 MFMA; 256 threads, K slabs of 16
// op(A)(i,k): A[i*lda+k] or (TA) A[k*lda+i]; op(B)(k,j): B[k*ldb+j] or (TB) B[j*ldb+k]
// only k in [kLo, kHi) is visited (callers pass the triangular support; everything outside is exact zero)
// ---------------------------------------------------------------------------------------------------------
template <bool TA, bool TB>
__device__ __forceinline__ void gemm_tile_f64(const double* __restrict__ A, int64_t lda, const double* __restrict__ B,
                                              int64_t ldb, double* __restrict__ C, int64_t ldc, int M, int N, int K,
                                              int r0, int c0, int kLo, int kHi, double alpha, double beta) {
  __shared__ double As[64][17];
  __shared__ double Bs[16][66];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  f64x4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = (f64x4){0.0, 0.0, 0.0, 0.0};
  if (kLo < 0) kLo = 0;
  kLo &= ~15;
  if (kHi > K) kHi = K;

  double ra[4], rb[4];
  // tiles that lie inside the matrices with 16-byte aligned rows (all but the last tile row / column at D = 784) fetch their
  // four consecutive values as two 16-byte loads without per-element bounds checks
  typedef double f64x2 __attribute__((ext_vector_type(2)));
  const bool inner = r0 + 64 <= M && c0 + 64 <= N && !(lda & 1) && !(ldb & 1) &&
                     !(reinterpret_cast<uintptr_t>(A) & 15) && !(reinterpret_cast<uintptr_t>(B) & 15);
  auto load4 = [&](const double* __restrict__ p, double (&r)[4]) {
    const f64x2 v0 = *reinterpret_cast<const f64x2*>(p), v1 = *reinterpret_cast<const f64x2*>(p + 2);
    r[0] = v0[0]; r[1] = v0[1]; r[2] = v1[0]; r[3] = v1[1];
  };
  auto load = [&](int k0) {
    if (inner && k0 + 16 <= K) {                // block-uniform
      if (!TA) load4(A + (int64_t)(r0 + (tid >> 2)) * lda + k0 + (tid & 3) * 4, ra);
      else load4(A + (int64_t)(k0 + (tid >> 4)) * lda + r0 + (tid & 15) * 4, ra);
      if (!TB) load4(B + (int64_t)(k0 + (tid >> 4)) * ldb + c0 + (tid & 15) * 4, rb);
      else load4(B + (int64_t)(c0 + (tid >> 2)) * ldb + k0 + (tid & 3) * 4, rb);
      return;
    }
    if (!TA) {
      const int row = tid >> 2, kq = (tid & 3) * 4;
      const int gr = r0 + row;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int gk = k0 + kq + e;
        ra[e] = (gr < M && gk < K) ? A[(int64_t)gr * lda + gk] : 0.0;
      }
    } else {
      const int k = tid >> 4, iq = (tid & 15) * 4;
      const int gk = k0 + k;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int gr = r0 + iq + e;
        ra[e] = (gr < M && gk < K) ? A[(int64_t)gk * lda + gr] : 0.0;
      }
    }
    if (!TB) {
      const int k = tid >> 4, jq = (tid & 15) * 4;
      const int gk = k0 + k;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int gc = c0 + jq + e;
        rb[e] = (gc < N && gk < K) ? B[(int64_t)gk * ldb + gc] : 0.0;
      }
    } else {
      const int col = tid >> 2, kq = (tid & 3) * 4;
      const int gc = c0 + col;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int gk = k0 + kq + e;
        rb[e] = (gc < N && gk < K) ? B[(int64_t)gc * ldb + gk] : 0.0;
      }
    }
  };
  auto stage = [&]() {
    if (!TA) {
      const int row = tid >> 2, kq = (tid & 3) * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) As[row][kq + e] = ra[e];
    } else {
      const int k = tid >> 4, iq = (tid & 15) * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) As[iq + e][k] = ra[e];
    }
    if (!TB) {
      const int k = tid >> 4, jq = (tid & 15) * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) Bs[k][jq + e] = rb[e];
    } else {
      const int col = tid >> 2, kq = (tid & 3) * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) Bs[kq + e][col] = rb[e];
    }
  };

  if (kLo < kHi) load(kLo);
  for (int k0 = kLo; k0 < kHi; k0 += 16) {
    __syncthreads();                 // previous slab's fragment reads are done
    stage();
    __syncthreads();
    if (k0 + 16 < kHi) load(k0 + 16);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      double a[2], b[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        a[t] = As[wm * 32 + t * 16 + (lane & 15)][kk * 4 + (lane >> 4)];
        b[t] = Bs[kk * 4 + (lane >> 4)][wn * 32 + t * 16 + (lane & 15)];
      }
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
          acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ti], b[tj], acc[ti][tj], 0, 0, 0);
    }
  }
  // C/D layout of the f64 MFMA: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gr = r0 + wm * 32 + ti * 16 + (lane >> 4) + 4 * r;
        const int gc = c0 + wn * 32 + tj * 16 + (lane & 15);
        if (gr < M && gc < N) {
          double v = alpha * acc[ti][tj][r];
          if (beta != 0.0) v += beta * C[(int64_t)gr * ldc + gc];
          C[(int64_t)gr * ldc + gc] = v;
        }
      }
}

struct GemmArgs {
  const double* A; int64_t lda, sA;
  const double* B; int64_t ldb, sB;
  double* C; int64_t ldc, sC;
  int M, N, K;
  int tri;            // low 3 bits: k-range hint (0 full; 1: k <= min(i,j); 2: k >= max(i,j); 3: k <= i; 4: k <= j);
                      // bits 3-4: output mask (1: only tiles touching the upper triangle, 2: ... the lower triangle)
  double alpha, beta;
};

// XCD-aware block map of a batched launch, grid (gx, gy, nz rounded up to a multiple of 8): the hardware deals consecutive
// blocks (x fastest) round-robin over the 8 XCDs, so in the plain map the tiles of ONE matrix are spread over all eight L2s and
// every L2 streams every matrix.  Here block L = 8 slot + xcd works on matrix 8 (slot / tiles) + xcd, tile slot % tiles: all
// tiles of a matrix share one XCD's L2 (its operands are re-read by the tiles of its row / column).  false: padding block.
__device__ __forceinline__ bool xcd_batch_map(int nz, int& bx, int& by, int& bz) {
  const int gx = gridDim.x, per = gridDim.x * gridDim.y;
  const int L = blockIdx.x + gx * (blockIdx.y + gridDim.y * blockIdx.z);
  const int xcd = L & 7, slot = L >> 3;
  const int grp = slot / per, t = slot - grp * per;
  bz = grp * 8 + xcd;
  by = t / gx;
  bx = t - by * gx;
  return bz < nz;
}

// grid (tilesN, tilesM, batch).  (The XCD-aware maps of tri_level_kernel were measured here too: whole matrices per XCD leave 33
// matrices unevenly over 8 XCDs, 10-15 % slower; one row of tiles per XCD: no different from the plain map.)
template <bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_f64_kernel(GemmArgs g) {
  const int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  const int r0 = by * 64, c0 = bx * 64;
  const int kmode = g.tri & 7, omask = g.tri >> 3;
  if (omask == 1 && r0 >= c0 + 64) return;       // tile strictly below the diagonal: not wanted, left untouched
  if (omask == 2 && c0 >= r0 + 64) return;       // ... strictly above
  int kLo = 0, kHi = g.K;
  if (kmode == 1) kHi = (r0 < c0 ? r0 : c0) + 64;
  if (kmode == 2) kLo = (r0 > c0 ? r0 : c0);
  if (kmode == 3) kHi = r0 + 64;
  if (kmode == 4) kHi = c0 + 64;
  const int64_t b = bz;
  gemm_tile_f64<TA, TB>(g.A + b * g.sA, g.lda, g.B + b * g.sB, g.ldb, g.C + b * g.sC, g.ldc, g.M, g.N, g.K, r0, c0,
                        kLo, kHi, g.alpha, g.beta);
}

// one level of the recursive doubling on lower-triangular matrices; grid (tiles, pairs, matrices)
//   pair j covers rows/cols [p0, p0+s) (A, inverse known) and [p0+s, p0+s+sb) (B, inverse known), C = tri[B rows, A cols]
//   PHASE 1: work[C block] = C A^-1          PHASE 2: inv[C block] = -B^-1 work[C block]
template <int PHASE>
__global__ __launch_bounds__(256) void tri_level_kernel(const double* __restrict__ tri, double* __restrict__ inv,
                                                        double* __restrict__ work, int64_t D, int s, int nmat) {
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (nmat >= 8 && !xcd_batch_map(nmat, bx, by, bz)) return;       // (grid z: nmat rounded up to a multiple of 8)
  const int64_t p0 = 2 * (int64_t)by * s;
  const int sa = s;
  const int64_t rest = D - p0 - s;
  const int sb = (int)(rest < s ? rest : s);
  const int tilesN = (sa + 63) / 64;
  const int r0 = (bx / tilesN) * 64, c0 = (bx % tilesN) * 64;
  if (r0 >= sb) return;
  const int64_t mo = (int64_t)bz * D * D;
  const int64_t offC = (p0 + s) * D + p0;
  if (PHASE == 1) {
    // (C A^-1)[i][j] = sum_{k >= j} C[i][k] A^-1[k][j]
    gemm_tile_f64<false, false>(tri + mo + offC, D, inv + mo + p0 * D + p0, D, work + mo + offC, D, sb, sa, sa, r0, c0,
                                c0, sa, 1.0, 0.0);
  } else {
    // (B^-1 T)[i][j] = sum_{k <= i} B^-1[i][k] T[k][j]
    gemm_tile_f64<false, false>(inv + mo + (p0 + s) * D + (p0 + s), D, work + mo + offC, D, inv + mo + offC, D, sb, sa,
                                sb, r0, c0, 0, r0 + 64, -1.0, 0.0);
  }
}

// ---------------------------------------------------------------------------------------------------------
// Householder product: one wave per row of w_0 (row-local: row <- row - 2 (row.v_k) v_k^T / (v_k.v_k))
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void householder_kernel(const float* __restrict__ w0, const float* __restrict__ vk,
                                                         int64_t nvs, int64_t D, double* __restrict__ out) {
  extern __shared__ double row[];
  const int lane = threadIdx.x;
  const int64_t i = blockIdx.x;
  for (int64_t j = lane; j < D; j += 64) row[j] = (double)w0[i * D + j];
  for (int64_t k = 0; k < nvs; ++k) {
    const float* v = vk + k * D;
    double dot = 0.0, vv = 0.0;
    for (int64_t j = lane; j < D; j += 64) {
      const double vj = (double)v[j];
      dot += row[j] * vj;
      vv += vj * vj;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      dot += __shfl_xor(dot, o, 64);
      vv += __shfl_xor(vv, o, 64);
    }
    const double f = 2.0 * dot / vv;
    for (int64_t j = lane; j < D; j += 64) row[j] -= f * (double)v[j];
  }
  for (int64_t j = lane; j < D; j += 64) out[i * D + j] = row[j];
}

// ---------------------------------------------------------------------------------------------------------
// fp64 -> permuted / padded fp32 weight + bf16x3 planes; grid (column blocks, n_out)
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint16_t bf16_rne(float x) {
  uint32_t u = __float_as_uint(x);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }

template <typename S>
__global__ __launch_bounds__(256) void pack_weight_kernel(const S* __restrict__ src, int64_t lds_, int transpose,
                                                          const int32_t* __restrict__ oidx, const int32_t* __restrict__ iidx,
                                                          int64_t n_in, float* __restrict__ W, int64_t ldw,
                                                          uint16_t* __restrict__ planes, int64_t ldp, int64_t plane_stride) {
  const int64_t o = blockIdx.y;
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int32_t so = oidx ? oidx[o] : (int32_t)o;
  float w = 0.0f;
  if (c < n_in) {
    const int32_t si = iidx ? iidx[c] : (int32_t)c;
    if (so >= 0 && si >= 0) w = (float)((transpose & 1) ? src[(int64_t)si * lds_ + so] : src[(int64_t)so * lds_ + si]);
    if (W) W[o * ldw + c] = w;
  }
  if (planes && c < ldp && (transpose & 2)) {          // two fp16 planes (USF_PLANES_F16X2)
    const _Float16 hi = (_Float16)w;
    const _Float16 lo = (_Float16)(w - (float)hi);
    planes[o * ldp + c] = __builtin_bit_cast(uint16_t, hi);
    planes[plane_stride + o * ldp + c] = __builtin_bit_cast(uint16_t, lo);
  } else if (planes && c < ldp) {
    const uint16_t hi = bf16_rne(w);
    const float r = w - bf16_to_f32(hi);
    const uint16_t mid = bf16_rne(r);
    const uint16_t lo = bf16_rne(r - bf16_to_f32(mid));
    planes[o * ldp + c] = hi;
    planes[plane_stride + o * ldp + c] = mid;
    planes[2 * plane_stride + o * ldp + c] = lo;
  }
}

// many pack jobs in ONE launch (grid z = job): a training step refreshes ~600 weight images, and what that costs
// on the device is not their bytes but 600 dependent dispatches (~7.5 us each)
template <typename S>
__device__ __forceinline__ void pack_job_body(const usf_pack_job& j, int64_t o, int64_t c) {
  const S* src = reinterpret_cast<const S*>(j.src);
  const int32_t so = j.out_idx ? j.out_idx[o] : (int32_t)o;
  float w = 0.0f;
  if (c < j.n_in) {
    const int32_t si = j.in_idx ? j.in_idx[c] : (int32_t)c;
    if (so >= 0 && si >= 0) w = (float)((j.transpose & 1) ? src[(int64_t)si * j.ld_src + so] : src[(int64_t)so * j.ld_src + si]);
    if (j.W) j.W[o * j.ldw + c] = w;
  }
  if (j.planes && c < j.ld_planes && (j.transpose & 2)) {   // two fp16 planes (USF_PLANES_F16X2)
    uint16_t* planes = reinterpret_cast<uint16_t*>(j.planes);
    const _Float16 hi = (_Float16)w;
    const _Float16 lo = (_Float16)(w - (float)hi);
    planes[o * j.ld_planes + c] = __builtin_bit_cast(uint16_t, hi);
    planes[j.plane_stride + o * j.ld_planes + c] = __builtin_bit_cast(uint16_t, lo);
  } else if (j.planes && c < j.ld_planes) {
    uint16_t* planes = reinterpret_cast<uint16_t*>(j.planes);
    const uint16_t hi = bf16_rne(w);
    const float r = w - bf16_to_f32(hi);
    const uint16_t mid = bf16_rne(r);
    const uint16_t lo = bf16_rne(r - bf16_to_f32(mid));
    planes[o * j.ld_planes + c] = hi;
    planes[j.plane_stride + o * j.ld_planes + c] = mid;
    planes[2 * j.plane_stride + o * j.ld_planes + c] = lo;
  }
}

// Transposed jobs (W[o][c] = src[in_idx[c]][out_idx[o]]) through a 32 x 32 LDS tile: the source is read along its rows
// (consecutive o: coalesced), the image written along its rows (consecutive c).  The one-element-per-thread kernel below reads
// a transposed source with one cache line per lane -- 585 us for the 128 transposed images of a cfg2 training step (236 MB),
// ten times the time of the same bytes read along rows.
__global__ __launch_bounds__(256) void pack_jobs_t_kernel(const usf_pack_job* __restrict__ jobs) {
  __shared__ float tile[32][33];
  const usf_pack_job j = jobs[blockIdx.z];
  const int64_t c0 = (int64_t)blockIdx.x * 32, o0 = (int64_t)blockIdx.y * 32;
  const int64_t cols = j.planes ? j.ld_planes : j.n_in;
  if (o0 >= j.n_out || c0 >= cols) return;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  {
    const int64_t o = o0 + tx;
    int32_t so = -1;
    if (o < j.n_out) so = j.out_idx ? j.out_idx[o] : (int32_t)o;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t c = c0 + ty + 8 * q;
      float w = 0.0f;
      if (c < j.n_in && so >= 0) {
        const int32_t si = j.in_idx ? j.in_idx[c] : (int32_t)c;
        if (si >= 0)
          w = j.src_is_f32 ? reinterpret_cast<const float*>(j.src)[(int64_t)si * j.ld_src + so]
                           : (float)reinterpret_cast<const double*>(j.src)[(int64_t)si * j.ld_src + so];
      }
      tile[ty + 8 * q][tx] = w;
    }
  }
  __syncthreads();
  const int64_t c = c0 + tx;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int64_t o = o0 + ty + 8 * q;
    if (o >= j.n_out) break;
    const float w = tile[tx][ty + 8 * q];
    if (c < j.n_in && j.W) j.W[o * j.ldw + c] = w;
    if (j.planes && c < j.ld_planes) {
      uint16_t* planes = reinterpret_cast<uint16_t*>(j.planes);
      if (j.transpose & 2) {                                   // two fp16 planes
        const _Float16 hi = (_Float16)w;
        const _Float16 lo = (_Float16)(w - (float)hi);
        planes[o * j.ld_planes + c] = __builtin_bit_cast(uint16_t, hi);
        planes[j.plane_stride + o * j.ld_planes + c] = __builtin_bit_cast(uint16_t, lo);
      } else {
        const uint16_t hi = bf16_rne(w);
        const float r = w - bf16_to_f32(hi);
        const uint16_t mid = bf16_rne(r);
        const uint16_t lo = bf16_rne(r - bf16_to_f32(mid));
        planes[o * j.ld_planes + c] = hi;
        planes[j.plane_stride + o * j.ld_planes + c] = mid;
        planes[2 * j.plane_stride + o * j.ld_planes + c] = lo;
      }
    }
  }
}

constexpr int PJ_ROWS = 8;     // rows per block: keeps the grid of (mostly empty) blocks of a mixed-size batch small
constexpr int PJ_COLS = 4;     // consecutive columns per thread: 16-byte stores of the fp32 image, 8-byte stores of each plane
// One thread = PJ_COLS consecutive columns of PJ_ROWS rows: the gather indices of its columns are read once, the source elements of
// a row are PJ_COLS independent loads, the image leaves as one 16-byte store and each plane as one 8-byte store (round 5: the
// one-element-per-thread form with 2-byte plane stores moved the 33 images of a cfg2 parameter refresh at 1.3 TB/s).  Same
// arithmetic, same bits.
template <typename S>
__device__ __forceinline__ void pack_job_rows(const usf_pack_job& j, int64_t o0, int64_t c0) {
  const S* src = reinterpret_cast<const S*>(j.src);
  int32_t si[PJ_COLS];
#pragma unroll
  for (int e = 0; e < PJ_COLS; ++e) {
    const int64_t c = c0 + e;
    si[e] = (c < j.n_in) ? (j.in_idx ? j.in_idx[c] : (int32_t)c) : -1;
  }
  const bool w_vec = j.W && c0 + PJ_COLS <= j.n_in && (j.ldw & 3) == 0 && (reinterpret_cast<uintptr_t>(j.W) & 15u) == 0;
  const bool p_ok = j.planes && c0 < j.ld_planes;       // (ld_planes % 4 == 0 on this path: all four columns or none)
  uint16_t* planes = reinterpret_cast<uint16_t*>(j.planes);
  for (int r = 0; r < PJ_ROWS; ++r) {
    const int64_t o = o0 + r;
    if (o >= j.n_out) break;
    const int32_t so = j.out_idx ? j.out_idx[o] : (int32_t)o;
    float w[PJ_COLS];
#pragma unroll
    for (int e = 0; e < PJ_COLS; ++e) {
      w[e] = 0.0f;
      if (so >= 0 && si[e] >= 0)
        w[e] = (float)((j.transpose & 1) ? src[(int64_t)si[e] * j.ld_src + so] : src[(int64_t)so * j.ld_src + si[e]]);
    }
    if (w_vec) {
      *reinterpret_cast<f32x4*>(j.W + o * j.ldw + c0) = (f32x4){w[0], w[1], w[2], w[3]};
    } else if (j.W) {
#pragma unroll
      for (int e = 0; e < PJ_COLS; ++e)
        if (c0 + e < j.n_in) j.W[o * j.ldw + c0 + e] = w[e];
    }
    if (p_ok) {
      typedef uint16_t u16x4 __attribute__((ext_vector_type(4)));
      u16x4 p0, p1, p2;
      if (j.transpose & 2) {                                   // two fp16 planes (USF_PLANES_F16X2)
#pragma unroll
        for (int e = 0; e < PJ_COLS; ++e) {
          const _Float16 hi = (_Float16)w[e];
          const _Float16 lo = (_Float16)(w[e] - (float)hi);
          p0[e] = __builtin_bit_cast(uint16_t, hi);
          p1[e] = __builtin_bit_cast(uint16_t, lo);
        }
        *reinterpret_cast<u16x4*>(planes + o * j.ld_planes + c0) = p0;
        *reinterpret_cast<u16x4*>(planes + j.plane_stride + o * j.ld_planes + c0) = p1;
      } else {
#pragma unroll
        for (int e = 0; e < PJ_COLS; ++e) {
          const uint16_t hi = bf16_rne(w[e]);
          const float rr = w[e] - bf16_to_f32(hi);
          const uint16_t mid = bf16_rne(rr);
          p0[e] = hi;
          p1[e] = mid;
          p2[e] = bf16_rne(rr - bf16_to_f32(mid));
        }
        *reinterpret_cast<u16x4*>(planes + o * j.ld_planes + c0) = p0;
        *reinterpret_cast<u16x4*>(planes + j.plane_stride + o * j.ld_planes + c0) = p1;
        *reinterpret_cast<u16x4*>(planes + 2 * j.plane_stride + o * j.ld_planes + c0) = p2;
      }
    }
  }
}

__global__ __launch_bounds__(256) void pack_jobs_kernel(const usf_pack_job* __restrict__ jobs) {
  const usf_pack_job j = jobs[blockIdx.z];
  const int64_t cols = j.planes ? j.ld_planes : j.n_in;
  if ((int64_t)blockIdx.y * PJ_ROWS >= j.n_out || (int64_t)blockIdx.x * (256 * PJ_COLS) >= cols) return;
  // the 4-column form needs whole groups of planes columns and aligned plane rows (block-uniform: a property of the job)
  const bool grouped = !j.planes || ((j.ld_planes & 3) == 0 && (j.plane_stride & 3) == 0 && (reinterpret_cast<uintptr_t>(j.planes) & 7u) == 0);
  if (grouped) {
    const int64_t c0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * PJ_COLS;
    if (c0 >= cols) return;
    if (j.src_is_f32) pack_job_rows<float>(j, (int64_t)blockIdx.y * PJ_ROWS, c0);
    else pack_job_rows<double>(j, (int64_t)blockIdx.y * PJ_ROWS, c0);
    return;
  }
  for (int e = 0; e < PJ_COLS; ++e) {
    const int64_t c = ((int64_t)blockIdx.x * PJ_COLS + e) * 256 + threadIdx.x;
    if (c >= cols) break;
    for (int r = 0; r < PJ_ROWS; ++r) {
      const int64_t o = (int64_t)blockIdx.y * PJ_ROWS + r;
      if (o >= j.n_out) break;
      if (j.src_is_f32) pack_job_body<float>(j, o, c);
      else pack_job_body<double>(j, o, c);
    }
  }
}

// c[o] = alpha * sum_k src[idx[o], k] * b[k]  (0 where idx[o] < 0); one wave per output
__global__ __launch_bounds__(64) void matvec_rows_kernel(const double* __restrict__ src, int64_t lds_, int64_t K,
                                                         const int32_t* __restrict__ idx, const double* __restrict__ b,
                                                         double alpha, float* __restrict__ out32,
                                                         double* __restrict__ out64) {
  const int64_t o = blockIdx.x;
  const int32_t so = idx ? idx[o] : (int32_t)o;
  double s = 0.0;
  if (so >= 0)
    for (int64_t k = threadIdx.x; k < K; k += 64) s += src[(int64_t)so * lds_ + k] * b[k];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if (threadIdx.x == 0) {
    if (out32) out32[o] = (float)(alpha * s);
    if (out64) out64[o] = alpha * s;
  }
}

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
static int launch_gemm(const GemmArgs& g, int transA, int transB, int64_t batch, hipStream_t stream) {
  if (g.M <= 0 || g.N <= 0 || batch <= 0) return 0;
  dim3 grid((g.N + 63) / 64, (g.M + 63) / 64, (unsigned)batch);
  if (!transA && !transB) gemm_f64_kernel<false, false><<<grid, 256, 0, stream>>>(g);
  else if (transA && !transB) gemm_f64_kernel<true, false><<<grid, 256, 0, stream>>>(g);
  else if (!transA && transB) gemm_f64_kernel<false, true><<<grid, 256, 0, stream>>>(g);
  else gemm_f64_kernel<true, true><<<grid, 256, 0, stream>>>(g);
  return check_launch("usf_gemm_f64");
}

int gemm_f64(const double* A, int64_t lda, int64_t sA, int transA, const double* B, int64_t ldb, int64_t sB, int transB,
             double* C, int64_t ldc, int64_t sC, int64_t M, int64_t N, int64_t K, int64_t batch, double alpha,
             double beta, int32_t tri, hipStream_t stream) {
  if (!A || !B || !C || M < 0 || N < 0 || K < 0 || batch < 0 || tri < 0 || (tri & 7) > 4 || (tri >> 3) > 2) {
    set_error("usf_gemm_f64: bad arguments");
    return -1;
  }
  if (M > (1 << 24) || N > (1 << 24) || K > (1 << 24) || batch > 65535) {
    set_error("usf_gemm_f64: size out of range");
    return -2;
  }
  GemmArgs g{A, lda, sA, B, ldb, sB, C, ldc, sC, (int)M, (int)N, (int)K, tri, alpha, beta};
  return launch_gemm(g, transA, transB, batch, stream);
}

int lu_prepare(const usf_lu_prep_desc* d, hipStream_t stream) {
  if (!d || d->n < 0 || d->D <= 0 || (d->n > 0 && (!d->L_raw || !d->U_raw || !d->tri || !d->tri_inv || !d->work))) {
    set_error("usf_lu_prepare_f64: bad descriptor");
    return -1;
  }
  const int64_t n = d->n, D = d->D;
  if (n == 0) return 0;
  if (D > (1 << 15) || 2 * n > 65535) {
    set_error("usf_lu_prepare_f64: D (%lld) or n (%lld) out of range", (long long)D, (long long)n);
    return -2;
  }
  for (int64_t i = 0; i < n; ++i)
    if (!d->L_raw[i] || !d->U_raw[i]) {
      set_error("usf_lu_prepare_f64: null parameter pointer at %lld", (long long)i);
      return -3;
    }
  // 1. unpack (chunks of PTRS_PER_LAUNCH: the pointer table travels as a kernel argument)
  for (int64_t first = 0; first < n; first += PTRS_PER_LAUNCH) {
    const int cnt = (int)((n - first) < PTRS_PER_LAUNCH ? (n - first) : PTRS_PER_LAUNCH);
    PtrTab t;
    for (int i = 0; i < PTRS_PER_LAUNCH; ++i) {
      t.L[i] = d->L_raw[first + (i < cnt ? i : 0)];
      t.U[i] = d->U_raw[first + (i < cnt ? i : 0)];
    }
    int bx = (int)((D * D + 255) / 256);
    if (bx > 1024) bx = 1024;
    lu_unpack_kernel<<<dim3(bx, cnt), 256, 0, stream>>>(t, first, D, d->tri);
    if (d->ladj) lu_ladj_kernel<<<cnt, 64, 0, stream>>>(t, first, D, d->ladj);
  }
  int rc = check_launch("lu_unpack");
  if (rc) return rc;
  // 2. zero the inverse (its upper triangle and not-yet-written blocks must read as exact zeros), diagonal blocks
  hipError_t e = hipMemsetAsync(d->tri_inv, 0, sizeof(double) * 2 * n * D * D, stream);
  if (e != hipSuccess) { set_error("usf_lu_prepare_f64: memset failed: %s", hipGetErrorString(e)); return (int)e; }
  const int nblk = (int)((D + NB - 1) / NB);
  tri_diag_inv_kernel<<<dim3(nblk, (unsigned)(2 * n)), 64, 0, stream>>>(d->tri, d->tri_inv, D);
  // 3. recursive doubling
  for (int64_t s = NB; s < D; s *= 2) {
    const int pairs = (int)((D - s + 2 * s - 1) / (2 * s));
    const int tiles = (int)(((s + 63) / 64) * ((s + 63) / 64));
    const int nmat = (int)(2 * n);
    dim3 grid(tiles, pairs, (unsigned)(nmat >= 8 ? (nmat + 7) / 8 * 8 : nmat));
    tri_level_kernel<1><<<grid, 256, 0, stream>>>(d->tri, d->tri_inv, d->work, D, (int)s, nmat);
    tri_level_kernel<2><<<grid, 256, 0, stream>>>(d->tri, d->tri_inv, d->work, D, (int)s, nmat);
  }
  rc = check_launch("tri_level");
  if (rc) return rc;
  // 4. M = L U = L (U^T)^T ; M^-1 = U^-1 L^-1 = ((U^T)^-1)^T L^-1      (transforms.py:1281-1293)
  const int64_t DD = D * D;
  if (d->M) {
    GemmArgs g{d->tri, D, 2 * DD, d->tri + DD, D, 2 * DD, d->M, D, DD, (int)D, (int)D, (int)D, 1, 1.0, 0.0};
    rc = launch_gemm(g, 0, 1, n, stream);
    if (rc) return rc;
  }
  if (d->Minv) {
    GemmArgs g{d->tri_inv + DD, D, 2 * DD, d->tri_inv, D, 2 * DD, d->Minv, D, DD, (int)D, (int)D, (int)D, 2, 1.0, 0.0};
    rc = launch_gemm(g, 1, 0, n, stream);
    if (rc) return rc;
  }
  return 0;
}

// dL_out = tril(dL + TL, -1), dU_out = triu(dU + TU) + diag(c / diag U): one thread per element pair (usf_lu_grad_finish_f64)
__global__ __launch_bounds__(256) void lu_grad_finish_kernel(const double* __restrict__ dL, const double* __restrict__ dU,
                                                            const double* __restrict__ TL, const double* __restrict__ TU,
                                                            const double* __restrict__ c, const double* __restrict__ tri, int64_t D,
                                                            float* __restrict__ oL, float* __restrict__ oU) {
  const int64_t i = blockIdx.y;
  const int64_t DD = D * D;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < DD; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e / D, cn = e - r * D;
    const int64_t o = i * DD + e;
    float vl = 0.f, vu = 0.f;
    if (r > cn) {
      double v = dL[o];
      if (TL) v += TL[o];
      vl = (float)v;
    } else {
      double v = dU[o];
      if (r == cn) v += c[i] / tri[(2 * i + 1) * DD + e];       // (the addition order of the torch formulation: triu, + diagonal term, + TU)
      if (TU) v += TU[o];
      vu = (float)v;
    }
    oL[o] = vl;
    oU[o] = vu;
  }
}

int lu_grad_finish(const double* dL, const double* dU, const double* TL, const double* TU, const double* c, const double* tri,
                   int64_t n, int64_t D, float* oL, float* oU, hipStream_t stream) {
  if (n < 0 || D <= 0 || n > 65535 || D > 46340) { set_error("usf_lu_grad_finish_f64: bad sizes"); return -2; }
  if (n == 0) return 0;
  if (!dL || !dU || !c || !tri || !oL || !oU || (!TL) != (!TU)) { set_error("usf_lu_grad_finish_f64: null pointer (TL and TU: both or neither)"); return -1; }
  int64_t bx = (D * D + 255) / 256;
  if (bx > 4096) bx = 4096;
  lu_grad_finish_kernel<<<dim3((unsigned)bx, (unsigned)n), 256, 0, stream>>>(dL, dU, TL, TU, c, tri, D, oL, oU);
  return check_launch("usf_lu_grad_finish_f64");
}

int householder(const float* w0, const float* vk, int64_t nvs, int64_t D, double* out, hipStream_t stream) {
  if (!w0 || !out || D <= 0 || nvs < 0 || (nvs > 0 && !vk) || D > 8000) {
    set_error("usf_householder_f64: bad arguments");
    return -1;
  }
  householder_kernel<<<(unsigned)D, 64, sizeof(double) * D, stream>>>(w0, vk, nvs, D, out);
  return check_launch("usf_householder_f64");
}

int pack_weight(const void* src, int32_t src_is_f32, int64_t lds_, int32_t transpose, const int32_t* out_idx, int64_t n_out,
                const int32_t* in_idx, int64_t n_in, float* W, int64_t ldw, void* planes, int64_t ldp,
                int64_t plane_stride, hipStream_t stream) {
  if (!src || n_out < 0 || n_in < 0 || (!W && !planes) || (W && ldw < n_in) || (planes && ldp < n_in) || n_out > 65535) {
    set_error("usf_pack_weight_f32: bad arguments");
    return -1;
  }
  if (n_out == 0 || n_in == 0) return 0;
  const int64_t cols = planes ? ldp : n_in;
  const dim3 grid((unsigned)((cols + 255) / 256), (unsigned)n_out);
  if (src_is_f32)
    pack_weight_kernel<float><<<grid, 256, 0, stream>>>((const float*)src, lds_, transpose, out_idx, in_idx, n_in, W, ldw,
                                                        (uint16_t*)planes, ldp, plane_stride);
  else
    pack_weight_kernel<double><<<grid, 256, 0, stream>>>((const double*)src, lds_, transpose, out_idx, in_idx, n_in, W,
                                                         ldw, (uint16_t*)planes, ldp, plane_stride);
  return check_launch("usf_pack_weight_f32");
}

int pack_jobs(const usf_pack_job* jobs, int64_t n_jobs, int64_t max_rows, int64_t max_cols, hipStream_t stream) {
  if (n_jobs < 0 || max_rows < 0 || max_cols < 0 || (n_jobs > 0 && !jobs) || n_jobs > 65535 || max_rows > 65535) {
    set_error("usf_pack_weights_f32: bad arguments");
    return -1;
  }
  if (n_jobs == 0 || max_rows == 0 || max_cols == 0) return 0;
  pack_jobs_kernel<<<dim3((unsigned)((max_cols + 256 * PJ_COLS - 1) / (256 * PJ_COLS)), (unsigned)((max_rows + PJ_ROWS - 1) / PJ_ROWS),
                          (unsigned)n_jobs), 256, 0, stream>>>(jobs);
  return check_launch("usf_pack_weights_f32");
}

int pack_jobs_t(const usf_pack_job* jobs, int64_t n_jobs, int64_t max_rows, int64_t max_cols, hipStream_t stream) {
  if (n_jobs < 0 || max_rows < 0 || max_cols < 0 || (n_jobs > 0 && !jobs) || n_jobs > 65535 || max_rows > 65535 * 32) {
    set_error("usf_pack_weights_t_f32: bad arguments");
    return -1;
  }
  if (n_jobs == 0 || max_rows == 0 || max_cols == 0) return 0;
  pack_jobs_t_kernel<<<dim3((unsigned)((max_cols + 31) / 32), (unsigned)((max_rows + 31) / 32), (unsigned)n_jobs), 256, 0,
                       stream>>>(jobs);
  return check_launch("usf_pack_weights_t_f32");
}

int matvec_rows(const double* src, int64_t lds_, int64_t K, const int32_t* idx, int64_t n_out, const double* b,
                double alpha, float* out32, double* out64, hipStream_t stream) {
  if (!src || !b || (!out32 && !out64) || n_out < 0 || K < 0) {
    set_error("usf_matvec_f64: bad arguments");
    return -1;
  }
  if (n_out == 0) return 0;
  matvec_rows_kernel<<<(unsigned)n_out, 64, 0, stream>>>(src, lds_, K, idx, b, alpha, out32, out64);
  return check_launch("usf_matvec_f64");
}

}  // namespace usf
