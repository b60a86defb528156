// HBM-bound kernels of the path: base-density tail (per-sample wave reduction), base sampling
// head (Philox), standalone scale layer, column gather.  One wave per row, 16-B lane accesses.
#include "usf_common.h"

namespace usf {

// ------------------------------------------------------------------------------------------
// tail: logp[m] = sum_d f(z[m,d]) + c    |   r[m] = ||z[m,:]-loc||_p
// ------------------------------------------------------------------------------------------
// cst = the per-feature constant of the density (hoisted out of the row loop: one logf per feature per
// block instead of one per element): Laplace -log(2 b), Normal -log(sigma) - log(sqrt(2 pi))
__device__ __forceinline__ float base_const(float scale, int base) {
  if (base == USF_BASE_LAPLACE) return -logf(2.0f * scale);
  if (base == USF_BASE_NORMAL) return -logf(scale) - 0.91893853320467274178f;
  return 0.f;
}
__device__ __forceinline__ float base_term(float z, float loc, float scale, float cst, int base) {
  switch (base) {
    case USF_BASE_LAPLACE:   // torch Laplace.log_prob: -log(2*scale) - |v-loc|/scale
      return cst - fabsf(z - loc) / scale;
    case USF_BASE_NORMAL: {  // torch Normal.log_prob: -((v-loc)^2)/(2 var) - log(scale) - log(sqrt(2 pi))
      const float d = z - loc;
      return -(d * d) / (2.0f * (scale * scale)) + cst;
    }
    case USF_BASE_LPNORM1:   return fabsf(z - loc);
    case USF_BASE_LPNORM2: { const float d = z - loc; return d * d; }
    default:                 return fabsf(z - loc);   // LPNORMINF (max-reduced)
  }
}

template <int BASE>
__global__ __launch_bounds__(256) void base_logprob_kernel(const float* __restrict__ z, int64_t ldz, int M, int D,
                                                           const float* __restrict__ loc,
                                                           const float* __restrict__ scale, float logdet_const,
                                                           const double* __restrict__ logdet_dev,
                                                           float* __restrict__ logp, double* __restrict__ sum_out) {
  if (logdet_dev) logdet_const += (float)*logdet_dev;      // the constant as a device scalar: no host round trip
  // per-feature tables in LDS, built once per block: loc | scale | the density's constant (the last two: Laplace / Normal only).
  // Only the rows of z come from HBM in the row loop (round 5: with loc / scale re-read from global memory next to every 16 bytes
  // of z the kernel streamed 2.9 TB/s)
  extern __shared__ __attribute__((aligned(16))) float base_tab[];
  const int lane = threadIdx.x & 63;
  const int wave_in_block = threadIdx.x >> 6;
  const int waves_per_block = blockDim.x >> 6;
  double block_sum = 0.0;
  constexpr bool HAS_CST = (BASE == USF_BASE_LAPLACE || BASE == USF_BASE_NORMAL);
  const int Dp = (D + 3) & ~3;
  float* const loc_t = base_tab;
  float* const scale_t = base_tab + Dp;
  float* const cst_t = base_tab + 2 * Dp;
  const bool vec = ((ldz & 3) == 0) && ((reinterpret_cast<uintptr_t>(z) & 15u) == 0);
  const int D4 = vec ? (D & ~3) : 0;
  for (int d = threadIdx.x; d < D; d += blockDim.x) {
    loc_t[d] = loc[d];
    if (HAS_CST) {
      scale_t[d] = scale[d];
      cst_t[d] = base_const(scale[d], BASE);
    }
  }
  __syncthreads();
  for (int64_t row = (int64_t)blockIdx.x * waves_per_block + wave_in_block; row < M;
       row += (int64_t)gridDim.x * waves_per_block) {
    const float* zr = z + row * ldz;
    float acc = 0.f;
#pragma unroll 4
    for (int d = lane * 4; d < D4; d += 256) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(zr + d);
      const f32x4 l = *reinterpret_cast<const f32x4*>(loc_t + d);
      f32x4 s = {1.f, 1.f, 1.f, 1.f}, c = {0.f, 0.f, 0.f, 0.f};
      if (HAS_CST) {
        s = *reinterpret_cast<const f32x4*>(scale_t + d);
        c = *reinterpret_cast<const f32x4*>(cst_t + d);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float t = base_term(v[j], l[j], s[j], c[j], BASE);
        acc = (BASE == USF_BASE_LPNORMINF) ? fmaxf(acc, t) : acc + t;
      }
    }
    for (int d = D4 + lane; d < D; d += 64) {
      const float s = HAS_CST ? scale_t[d] : 1.f;
      const float t = base_term(zr[d], loc_t[d], s, HAS_CST ? cst_t[d] : 0.f, BASE);
      acc = (BASE == USF_BASE_LPNORMINF) ? fmaxf(acc, t) : acc + t;
    }
    acc = (BASE == USF_BASE_LPNORMINF) ? wave_max(acc) : wave_sum(acc);
    float out;
    if (BASE == USF_BASE_LPNORM2) out = sqrtf(acc);
    else if (HAS_CST) out = acc + logdet_const;
    else out = acc;
    if (lane == 0) {
      logp[row] = out;
      block_sum += (double)out;
    }
  }
  if (sum_out != nullptr) {
    __shared__ double part[4];
    if (lane == 0) part[wave_in_block] = block_sum;
    __syncthreads();
    if (threadIdx.x == 0) {
      double t = 0.0;
      for (int w = 0; w < waves_per_block; ++w) t += part[w];
      if (t != 0.0) atomicAdd(&sum_out[0], t);
    }
  }
}

__global__ void add_count_kernel(double* sum_out, double n) { sum_out[1] += n; }

// USF_BASE_ROWSUM: the row's partial sums (one per column block of the last GEMM, usf_planes.hip) -> logp; one thread per row
__global__ __launch_bounds__(256) void base_rowsum_kernel(const float* __restrict__ z, int64_t ldz, int M, int D, float logdet_const,
                                                          const double* __restrict__ logdet_dev, float* __restrict__ logp,
                                                          double* __restrict__ sum_out) {
  if (logdet_dev) logdet_const += (float)*logdet_dev;
  double mine = 0.0;
  for (int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; row < M; row += (int64_t)gridDim.x * blockDim.x) {
    const float* zr = z + row * ldz;
    float acc = 0.f;
    for (int d = 0; d < D; ++d) acc += zr[d];
    const float out = acc + logdet_const;
    logp[row] = out;
    mine += (double)out;
  }
  if (sum_out != nullptr) {
    __shared__ double part[4];
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
      const double t = part[0] + part[1] + part[2] + part[3];
      if (t != 0.0) atomicAdd(&sum_out[0], t);
    }
  }
}

__global__ void base_tables_kernel(int base, const float* __restrict__ loc, const float* __restrict__ scale, int D, float* __restrict__ tab,
                                   int stride) {
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= stride) return;
  const bool in = d < D;
  tab[d] = in ? loc[d] : 0.f;
  tab[stride + d] = in ? 1.0f / scale[d] : 0.f;
  tab[2 * stride + d] = in ? base_const(scale[d], base) : 0.f;
}

int base_tables(int32_t base, const float* loc, const float* scale, int64_t D, float* tab, int64_t stride, hipStream_t stream) {
  if (D <= 0 || stride < D || (stride & 3) || stride > 0x7fffffff) { set_error("usf_base_tables_f32: bad sizes"); return -2; }
  if (!loc || !scale || !tab) { set_error("usf_base_tables_f32: null pointer"); return -1; }
  if (base != USF_BASE_LAPLACE && base != USF_BASE_NORMAL) { set_error("usf_base_tables_f32: base %d has no tables", base); return -2; }
  hipLaunchKernelGGL(base_tables_kernel, dim3((unsigned)((stride + 255) / 256)), dim3(256), 0, stream, (int)base, loc, scale, (int)D,
                     tab, (int)stride);
  return check_launch("usf_base_tables_f32");
}

int base_logprob(const float* z, int64_t ldz, int64_t M, int64_t D, int32_t base, const float* loc,
                 const float* scale, float logdet_const, const double* logdet_dev, float* logp, double* sum_out,
                 hipStream_t stream) {
  if (M < 0 || D <= 0 || M > 0x7fffffff || D > 0x7fffffff || ldz < D) { set_error("usf_base_logprob_f32: bad sizes"); return -2; }
  if (M == 0) return 0;
  if (base == USF_BASE_ROWSUM) {
    if (!z || !logp) { set_error("usf_base_logprob_f32: null pointer"); return -1; }
    if (D > 8) { set_error("usf_base_logprob_f32: USF_BASE_ROWSUM sums at most 8 partial sums per row"); return -2; }
    int64_t blocks = (M + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(base_rowsum_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, z, ldz, (int)M, (int)D, logdet_const,
                       logdet_dev, logp, sum_out);
    int rc = check_launch("usf_base_logprob_f32(rowsum)");
    if (rc == 0 && sum_out) {
      hipLaunchKernelGGL(add_count_kernel, dim3(1), dim3(1), 0, stream, sum_out, (double)M);
      rc = check_launch("usf_base_logprob_f32(count)");
    }
    return rc;
  }
  if (!z || !loc || !logp) { set_error("usf_base_logprob_f32: null pointer"); return -1; }
  if ((base == USF_BASE_LAPLACE || base == USF_BASE_NORMAL) && !scale) { set_error("usf_base_logprob_f32: scale required"); return -1; }
  const int wpb = 4;
  int64_t blocks = (M + wpb - 1) / wpb;
  if (blocks > 256 * 8) blocks = 256 * 8;        // grid-stride: the per-feature constant table is built once per block
  dim3 g((unsigned)blocks), b(256);
  const size_t tab = (size_t)((D + 3) / 4 * 4) * sizeof(float) * ((base == USF_BASE_LAPLACE || base == USF_BASE_NORMAL) ? 3 : 1);
  if (tab > 96 * 1024) { set_error("usf_base_logprob_f32: D = %lld too large for the per-feature tables", (long long)D); return -2; }
  // (dynamic LDS above HIP's 64 KB default needs the kernel's limit raised -- once per instantiation)
#define USF_LAUNCH_BASE(B)                                                                                            \
  do {                                                                                                                \
    if (tab > 64 * 1024) {                                                                                            \
      static bool attr_done_dev[USF_MAX_DEVICES] = {false};      /* (the attribute belongs to the device) */             \
      bool& attr_done = attr_done_dev[current_device_slot()];                                                         \
      if (!attr_done) {                                                                                               \
        const hipError_t attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(&base_logprob_kernel<B>),        \
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);        \
        if (attr_rc != hipSuccess) { set_error("usf_base_logprob_f32: cannot raise the LDS limit"); return (int)attr_rc; } \
        attr_done = true;                                                                                             \
      }                                                                                                               \
    }                                                                                                                 \
    hipLaunchKernelGGL((base_logprob_kernel<B>), g, b, tab, stream, z, ldz, (int)M, (int)D, loc, scale, logdet_const, \
                       logdet_dev, logp, sum_out);                                                                                \
  } while (0)
  switch (base) {
    case USF_BASE_LAPLACE: USF_LAUNCH_BASE(USF_BASE_LAPLACE); break;
    case USF_BASE_NORMAL: USF_LAUNCH_BASE(USF_BASE_NORMAL); break;
    case USF_BASE_LPNORM1: USF_LAUNCH_BASE(USF_BASE_LPNORM1); break;
    case USF_BASE_LPNORM2: USF_LAUNCH_BASE(USF_BASE_LPNORM2); break;
    case USF_BASE_LPNORMINF: USF_LAUNCH_BASE(USF_BASE_LPNORMINF); break;
    default: set_error("usf_base_logprob_f32: unknown base %d", base); return -2;
  }
#undef USF_LAUNCH_BASE
  int rc = check_launch("usf_base_logprob_f32");
  if (rc) return rc;
  if (sum_out) {
    hipLaunchKernelGGL(add_count_kernel, dim3(1), dim3(1), 0, stream, sum_out, (double)M);
    rc = check_launch("usf_base_logprob_f32(count)");
  }
  return rc;
}

// ------------------------------------------------------------------------------------------
// head: Philox4x32-10 counter RNG -> Laplace / Normal
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
  const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
  c[0] = n0; c[1] = (uint32_t)p1; c[2] = n2; c[3] = (uint32_t)p0;
}
__device__ __forceinline__ void philox4x32_10(uint64_t ctr, uint64_t stream_id, uint64_t seed, uint32_t (&out)[4]) {
  uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), (uint32_t)stream_id, (uint32_t)(stream_id >> 32)};
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}
// uniform in (0,1): 23 random bits at the cell centres (k + 1/2) * 2^-23 -- never 0 or 1.  Every step is EXACT in
// fp32: k + 0.5 <= 8388607.5 needs 24 significant bits (with 24 random bits the top value 16777215.5 is a
// round-to-even tie and became 2^24, i.e. u = 1 and log1p(-1) = -inf in the Laplace transform, once per 2^24
// draws); 2u - 1 = (2k + 1 - 2^23) * 2^-23 is exact too, |2u - 1| <= 1 - 2^-23, and never 0.
__device__ __forceinline__ float u01(uint32_t x) { return ((float)(x >> 9) + 0.5f) * (1.0f / 8388608.0f); }
// torch Laplace.rsample's transform of u in (-1, 1): -sign(u) * log1p(-|u|)
__device__ __forceinline__ float laplace_icdf(float u) {
  const float sgn = (u > 0.f) ? 1.f : ((u < 0.f) ? -1.f : 0.f);
  return -sgn * log1pf(-fabsf(u));
}

__global__ __launch_bounds__(256) void base_sample_kernel(float* __restrict__ z, int64_t ldz, int64_t M, int D,
                                                          int base, const float* __restrict__ loc,
                                                          const float* __restrict__ scale, uint64_t seed,
                                                          uint64_t offset, int64_t row_offset) {
  const int64_t groups_per_row = (D + 3) / 4;
  const int64_t total = M * groups_per_row;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = g / groups_per_row;
    const int d0 = (int)(g % groups_per_row) * 4;
    uint32_t rnd[4];
    philox4x32_10((uint64_t)((m + row_offset) * groups_per_row + d0 / 4), offset, seed, rnd);
    float v[4];
    if (base == USF_BASE_LAPLACE) {
      // torch Laplace.rsample: u ~ U(eps-1, 1); loc - scale*sign(u)*log1p(-|u|)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[j] = laplace_icdf(2.0f * u01(rnd[j]) - 1.0f);
      }
    } else {
      const float r0 = sqrtf(-2.0f * logf(u01(rnd[0]))), r1 = sqrtf(-2.0f * logf(u01(rnd[2])));
      float s0, c0, s1, c1;
      sincosf(6.28318530717958647692f * u01(rnd[1]), &s0, &c0);
      sincosf(6.28318530717958647692f * u01(rnd[3]), &s1, &c1);
      v[0] = r0 * c0; v[1] = r0 * s0; v[2] = r1 * c1; v[3] = r1 * s1;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int d = d0 + j;
      if (d < D) z[m * ldz + d] = loc[d] + scale[d] * v[j];
    }
  }
}

int base_sample(float* z, int64_t ldz, int64_t M, int64_t D, int32_t base, const float* loc, const float* scale,
                uint64_t seed, uint64_t offset, int64_t row_offset, hipStream_t stream) {
  if (M < 0 || D <= 0 || D > 0x7fffffff || ldz < D) { set_error("usf_base_sample_f32: bad sizes"); return -2; }
  if (M == 0) return 0;
  if (!z || !loc || !scale) { set_error("usf_base_sample_f32: null pointer"); return -1; }
  if (base != USF_BASE_LAPLACE && base != USF_BASE_NORMAL) { set_error("usf_base_sample_f32: base %d not sampled on device", base); return -2; }
  const int64_t total = M * ((D + 3) / 4);
  int64_t blocks = (total + 255) / 256;
  if (blocks > 256 * 8) blocks = 256 * 8;
  hipLaunchKernelGGL(base_sample_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, z, ldz, M, (int)D, base, loc,
                     scale, seed, offset, row_offset);
  return check_launch("usf_base_sample_f32");
}

// the word -> variate maps of the head kernels applied to caller-supplied random words (tests feed the extreme
// words 0 and 0xFFFFFFFF; a caller with its own generator can use it as the inverse-CDF stage)
__global__ void variates_from_bits_kernel(const uint32_t* __restrict__ bits, int64_t n, float* __restrict__ u,
                                          float* __restrict__ laplace, float* __restrict__ exponential) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = u01(bits[i]);
    if (u) u[i] = v;
    if (laplace) laplace[i] = laplace_icdf(2.0f * v - 1.0f);
    if (exponential) exponential[i] = -logf(v);
  }
}

int variates_from_bits(const uint32_t* bits, int64_t n, float* u, float* laplace, float* exponential, hipStream_t stream) {
  if (n < 0) { set_error("usf_variates_from_bits_f32: bad size"); return -2; }
  if (n == 0) return 0;
  if (!bits) { set_error("usf_variates_from_bits_f32: null pointer"); return -1; }
  int64_t blocks = (n + 255) / 256;
  if (blocks > 256 * 8) blocks = 256 * 8;
  hipLaunchKernelGGL(variates_from_bits_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, bits, n, u, laplace, exponential);
  return check_launch("usf_variates_from_bits_f32");
}

// ------------------------------------------------------------------------------------------
// RadialDistribution.sample (distributions.py:474-499): x = loc + r * u,  u uniform on the unit Lp sphere
// (UniformUnitLpBall.sample, distributions.py:283-319); r [M] comes from the caller (norm_distribution.sample).
//   p = 1  : Dirichlet(1,..,1) (= normalised Exp(1) variates) times random signs
//   p = 2  : normalised standard normals
//   p = inf: Uniform(-1,1) coordinates, one coordinate (uniformly chosen) set to +1.0 (the reference sets 1.0, not +-1)
// One wave per row: variates from Philox counter (row, d/4) of stream `offset`, wave reduction for the norm.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void radial_sample_kernel(float* __restrict__ z, int64_t ldz, int64_t M, int D, int base,
                                                            const float* __restrict__ loc, const float* __restrict__ r,
                                                            uint64_t seed, uint64_t offset, int64_t row_offset) {
  const int lane = threadIdx.x & 63;
  const int64_t groups_per_row = (D + 3) / 4;
  for (int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); m < M; m += (int64_t)gridDim.x * 4) {
    float* zr = z + m * ldz;
    float acc = 0.f;
    // pass 1: raw variates into the row, reduction of the normaliser
    for (int64_t g = lane; g < groups_per_row; g += 64) {
      uint32_t rnd[4], rnd2[4];
      philox4x32_10((uint64_t)((m + row_offset) * groups_per_row + g), offset, seed, rnd);
      float v[4];
      if (base == USF_BASE_LPNORM1) {
        philox4x32_10((uint64_t)((m + row_offset) * groups_per_row + g), offset ^ 0x5bd1e995u, seed, rnd2);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float e = -logf(u01(rnd[j]));
          acc += ((int)(g * 4 + j) < D) ? e : 0.f;
          v[j] = (rnd2[j] & 1u) ? e : -e;
        }
      } else if (base == USF_BASE_LPNORM2) {
        const float r0 = sqrtf(-2.0f * logf(u01(rnd[0]))), r1 = sqrtf(-2.0f * logf(u01(rnd[2])));
        float s0, c0, s1, c1;
        sincosf(6.28318530717958647692f * u01(rnd[1]), &s0, &c0);
        sincosf(6.28318530717958647692f * u01(rnd[3]), &s1, &c1);
        v[0] = r0 * c0; v[1] = r0 * s0; v[2] = r1 * c1; v[3] = r1 * s1;
#pragma unroll
        for (int j = 0; j < 4; ++j) acc += ((int)(g * 4 + j) < D) ? v[j] * v[j] : 0.f;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = 2.0f * u01(rnd[j]) - 1.0f;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if ((int)(g * 4 + j) < D) zr[g * 4 + j] = v[j];
    }
    acc = wave_sum(acc);
    float mul = r[m];
    int extremal = -1;
    if (base == USF_BASE_LPNORM1) mul = mul / acc;
    else if (base == USF_BASE_LPNORM2) mul = mul / sqrtf(acc);
    else {
      uint32_t rnd[4];
      philox4x32_10((uint64_t)(m + row_offset), offset ^ 0x9e3779b9u, seed, rnd);
      extremal = (int)(((uint64_t)rnd[0] * (uint64_t)D) >> 32);     // uniform index in [0, D)
    }
    // pass 2: each lane rescales the elements it wrote itself (no cross-lane dependency on memory)
    for (int64_t g = lane; g < groups_per_row; g += 64)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int d = (int)(g * 4 + j);
        if (d < D) zr[d] = loc[d] + mul * (d == extremal ? 1.0f : zr[d]);
      }
  }
}

int radial_sample(float* z, int64_t ldz, int64_t M, int64_t D, int32_t base, const float* loc, const float* r,
                  uint64_t seed, uint64_t offset, int64_t row_offset, hipStream_t stream) {
  if (M < 0 || D <= 0 || D > 0x7fffffff || ldz < D) { set_error("usf_radial_sample_f32: bad sizes"); return -2; }
  if (M == 0) return 0;
  if (!z || !loc || !r) { set_error("usf_radial_sample_f32: null pointer"); return -1; }
  if (base != USF_BASE_LPNORM1 && base != USF_BASE_LPNORM2 && base != USF_BASE_LPNORMINF) {
    set_error("usf_radial_sample_f32: base %d is not an Lp-radial id", base);
    return -2;
  }
  int64_t blocks = (M + 3) / 4;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(radial_sample_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, z, ldz, M, (int)D, base, loc, r,
                     seed, offset, row_offset);
  return check_launch("usf_radial_sample_f32");
}

// ------------------------------------------------------------------------------------------
// standalone scale layer and column gather
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void scale_kernel(const float* __restrict__ x, int64_t ldx, float* __restrict__ y,
                                                    int64_t ldy, int64_t M, int D, const float* __restrict__ s,
                                                    int divide) {
  const int64_t total = M * D;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = i / D;
    const int d = (int)(i % D);
    const float v = x[m * ldx + d];
    y[m * ldy + d] = divide ? v / s[d] : v * s[d];
  }
}

int scale(const float* x, int64_t ldx, float* y, int64_t ldy, int64_t M, int64_t D, const float* s, int32_t divide,
          hipStream_t stream) {
  if (M < 0 || D <= 0 || D > 0x7fffffff || ldx < D || ldy < D) { set_error("usf_scale_f32: bad sizes"); return -2; }
  if (M == 0) return 0;
  if (!x || !y || !s) { set_error("usf_scale_f32: null pointer"); return -1; }
  int64_t blocks = (M * D + 255) / 256;
  if (blocks > 256 * 8) blocks = 256 * 8;
  hipLaunchKernelGGL(scale_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, ldx, y, ldy, M, (int)D, s, divide);
  return check_launch("usf_scale_f32");
}

__global__ __launch_bounds__(256) void gather_cols_kernel(const float* __restrict__ src, int64_t lds_, float* __restrict__ dst,
                                                          int64_t ldd, int64_t M, int n, const int32_t* __restrict__ idx) {
  const int64_t total = M * n;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = i / n;
    const int j = (int)(i % n);
    const int c = idx[j];
    dst[m * ldd + j] = (c >= 0) ? src[m * lds_ + c] : 0.f;
  }
}

int gather_cols(const float* src, int64_t lds_, float* dst, int64_t ldd, int64_t M, int64_t n, const int32_t* idx,
                hipStream_t stream) {
  if (M < 0 || n <= 0 || n > 0x7fffffff || ldd < n) { set_error("usf_gather_cols_f32: bad sizes"); return -2; }
  if (M == 0) return 0;
  if (!src || !dst || !idx) { set_error("usf_gather_cols_f32: null pointer"); return -1; }
  int64_t blocks = (M * n + 255) / 256;
  if (blocks > 256 * 8) blocks = 256 * 8;
  hipLaunchKernelGGL(gather_cols_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, src, lds_, dst, ldd, M, (int)n, idx);
  return check_launch("usf_gather_cols_f32");
}

// ------------------------------------------------------------------------------------------
// Affine (scale-and-shift) coupling, the extension BASELINE.json's north_star names (the reference has additive
// coupling only, transforms.py:254-347: parity unpinned, opt-in, not uniformly scaling):
//   forward : z[m, j] = z[m, j] * exp(s[m, j]) + t[m, j]      logdet[m] += sum_j s[m, j]
//   inverse : z[m, j] = (z[m, j] - t[m, j]) * exp(-s[m, j])   logdet[m] -= sum_j s[m, j]
// for the transformed columns j < n of row m (z, t, s row-major with their own strides); s = bound * tanh(raw / bound)
// when bound > 0, else raw.  One wave per row: 16-byte lane accesses where alignment allows, the per-sample log-det
// by a 64-lane shuffle reduction -- the only per-sample reduction an affine coupling adds.  HBM-bound.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void affine_coupling_apply_kernel(float* __restrict__ z, int64_t ldz,
                                                                    const float* __restrict__ t, int64_t ldt,
                                                                    const float* __restrict__ sraw, int64_t lds_,
                                                                    int64_t M, int n, float bound, int inverse,
                                                                    float* __restrict__ logdet) {
  const int lane = threadIdx.x & 63;
  for (int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); m < M; m += (int64_t)gridDim.x * 4) {
    float* zr = z + m * ldz;
    const float* tr = t + m * ldt;
    const float* sr = sraw + m * lds_;
    float acc = 0.f;
    for (int j = lane; j < n; j += 64) {
      float s = sr[j];
      if (bound > 0.f) s = bound * tanhf(s / bound);
      acc += s;
      const float v = zr[j];
      zr[j] = inverse ? (v - tr[j]) * expf(-s) : v * expf(s) + tr[j];
    }
    acc = wave_sum(acc);
    if (lane == 0 && logdet) logdet[m] += inverse ? -acc : acc;
  }
}

int affine_coupling_apply(float* z, int64_t ldz, const float* t, int64_t ldt, const float* s, int64_t lds_, int64_t M,
                          int64_t n, float bound, int32_t inverse, float* logdet, hipStream_t stream) {
  if (M < 0 || n <= 0 || n > 0x7fffffff || ldz < n || ldt < n || lds_ < n) { set_error("usf_affine_coupling_apply_f32: bad sizes"); return -2; }
  if (M == 0) return 0;
  if (!z || !t || !s) { set_error("usf_affine_coupling_apply_f32: null pointer"); return -1; }
  int64_t blocks = (M + 3) / 4;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(affine_coupling_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, z, ldz, t, ldt, s, lds_, M,
                     (int)n, bound, inverse, logdet);
  return check_launch("usf_affine_coupling_apply_f32");
}

// ------------------------------------------------------------------------------------------
// BlockAffineTransform for image-shaped inputs: the 1 x 1 convolution
//   y[b, c, p] = sum_c' W[c, c'] * (x[b, c', p] - pre_sub[c']) + bias[c]        (x: [B, C, P] contiguous, P = H * W)
// (transforms.py:904-962: F.conv2d with the C x C block matrix viewed [C, C, 1, 1]).  C is the channel count (16 in
// the reference's MNIST configs): 2 C flops per 8 bytes -- HBM-bound.  One thread per pixel: every x element is read
// once, coalesced along p; the weights are wave-uniform (scalar loads); CMAX accumulators per thread.
// ------------------------------------------------------------------------------------------
template <int CMAX, bool FULL>
__global__ __launch_bounds__(256) void channel_affine_kernel(const float* __restrict__ x, float* __restrict__ y, int C,
                                                             int64_t P, const float* __restrict__ W,
                                                             const float* __restrict__ pre_sub,
                                                             const float* __restrict__ bias, int64_t BP, int co_per_block) {
  // pixels of all samples in one index space: with one block row per sample a 7 x 7 image left 207 of 256 threads idle
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= BP) return;
  // few pixels (a training batch of 32 samples): the output channels are dealt over blockIdx.y -- a thread then walks
  // co_per_block weight rows instead of all C (48 x 48 FMAs in one thread are 20 us whatever the batch); same sums
  const int co0 = blockIdx.y * co_per_block, co1 = co0 + co_per_block;
  const int64_t b = i / P, p = i - b * P;
  const float* xb = x + b * C * P + p;
  const int Cn = FULL ? CMAX : C;
  // all input channels first (CMAX independent loads in flight), then one output channel at a time: its weight row is
  // contiguous and wave-uniform (wide scalar loads), the sum runs over c' in ascending order
  float v[CMAX];
#pragma unroll
  for (int ci = 0; ci < CMAX; ++ci) {
    v[ci] = (FULL || ci < Cn) ? xb[(int64_t)ci * P] : 0.f;
    if (pre_sub && (FULL || ci < Cn)) v[ci] -= pre_sub[ci];
  }
  float* yb = y + b * C * P + p;
#pragma unroll
  for (int co = 0; co < CMAX; ++co) {
    if ((FULL || co < Cn) && co >= co0 && co < co1) {          // (wave-uniform)
      const float* wr = W + co * Cn;
      float acc = 0.f;
#pragma unroll
      for (int ci = 0; ci < CMAX; ++ci) {
        const float w = (FULL || ci < Cn) ? wr[ci] : 0.f;      // wave-uniform address
        acc = fmaf(w, v[ci], acc);
      }
      yb[(int64_t)co * P] = acc + (bias ? bias[co] : 0.f);
    }
  }
}

int channel_affine(const float* x, float* y, int64_t B, int64_t C, int64_t P, const float* W, const float* pre_sub,
                   const float* bias, hipStream_t stream) {
  if (B < 0 || C <= 0 || P <= 0 || C > 64) { set_error("usf_channel_affine_f32: bad sizes (C must be 1..64)"); return -2; }
  if (B == 0) return 0;
  if (!x || !y || !W) { set_error("usf_channel_affine_f32: null pointer"); return -1; }
  if (x == y) { set_error("usf_channel_affine_f32: in-place operation is not supported"); return -2; }
  const int64_t BP = B * P, blocks = (BP + 255) / 256;
  if (blocks > 0x7fffffffLL) { set_error("usf_channel_affine_f32: grid too large"); return -3; }
  // up to 64 pixel blocks (16 384 pixels: a quarter of the chip): 8 output channels per block row
  const int co_per_block = (blocks <= 64 && C > 8) ? 8 : 64;
  const dim3 b(256), g((unsigned)blocks, (unsigned)((C + co_per_block - 1) / co_per_block));
#define USF_CA(CM)                                                                                                       \
  do {                                                                                                                  \
    if (C == CM) hipLaunchKernelGGL((channel_affine_kernel<CM, true>), g, b, 0, stream, x, y, (int)C, P, W, pre_sub, bias, BP, co_per_block);  \
    else hipLaunchKernelGGL((channel_affine_kernel<CM, false>), g, b, 0, stream, x, y, (int)C, P, W, pre_sub, bias, BP, co_per_block);         \
  } while (0)
  if (C <= 8) USF_CA(8); else if (C <= 16) USF_CA(16); else if (C <= 24) USF_CA(24); else if (C <= 32) USF_CA(32);
  else if (C <= 48) USF_CA(48); else USF_CA(64);
#undef USF_CA
  return check_launch("usf_channel_affine_f32");
}

// ------------------------------------------------------------------------------------------
// Pointwise (1 x 1) convolution with few channels on the vector ALUs -- GatedConv's second convolution with its gate
// (networks.py:108-122: `x + val * sigmoid(gate)`, [val, gate] = conv1x1(f(t))) and plain 1 x 1 convolutions:
//     a = in_act(x);  plain:  y[b, co, p] = out_act(bias[co] + sum_ci W[co, ci] a[b, ci, p])
//                     gated:  y[b, c, p]  = gate_x[b, c, p] + (bias[c] + W[c] . a) * sigmoid(bias[C + c] + W[C + c] . a),  cout = 2 C
// 2 cin flops per 4-byte output element next to 4 (cin + cout) bytes per pixel: HBM-bound work that the matrix-core kernel
// (usf_conv.hip: LDS image, bf16x3 split, per-group barriers) serves at a quarter of the HBM rate; here one thread owns a
// pixel, loads its cin channel values once (coalesced along p), and walks the output channels with wave-uniform weight
// rows (wide scalar loads) -- exact fp32 FMAs (plain form: sums over ci in ascending order; gated forms: pw_dot2 below).
// ------------------------------------------------------------------------------------------
struct PwLn { const float* gamma; const float* beta; float eps; };
typedef float f32x2 __attribute__((ext_vector_type(2)));

// two dot products of wave-uniform weight rows with the pixel's channel values, on PACKED FMAs (v_pk_fma_f32: two fp32 FMAs per
// lane and instruction): input channels in pairs -- the pair of weights is one 64-bit scalar operand, the pair of values two
// neighbouring registers -- so each row keeps an even and an odd partial sum (ascending ci within each), added at the end
template <int CIN>
__device__ __forceinline__ void pw_dot2(const float* __restrict__ wa, const float* __restrict__ wb, const float (&v)[CIN],
                                        float& ra, float& rb) {
  f32x2 a2 = {0.f, 0.f}, b2 = {0.f, 0.f};
#pragma unroll
  for (int ci = 0; ci < CIN; ci += 2) {
    const f32x2 vv = {v[ci], v[ci + 1]};
    a2 = __builtin_elementwise_fma(*reinterpret_cast<const f32x2*>(wa + ci), vv, a2);
    b2 = __builtin_elementwise_fma(*reinterpret_cast<const f32x2*>(wb + ci), vv, b2);
  }
  ra = a2[0] + a2[1];
  rb = b2[0] + b2[1];
}

template <int CIN, bool GATED, bool LN>
__global__ __launch_bounds__(256) void pointwise_conv_kernel(const float* __restrict__ x, float* __restrict__ y, int cout,
                                                             int64_t P, const float* __restrict__ W,
                                                             const float* __restrict__ bias, int in_act, float in_slope,
                                                             int out_act, float out_slope,
                                                             const float* __restrict__ gate_x, int64_t BP, PwLn ln) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= BP) return;
  const int64_t b = i / P, p = i - b * P;
  const float* xb = x + b * CIN * P + p;
  float v[CIN];
#pragma unroll
  for (int ci = 0; ci < CIN; ++ci) v[ci] = act_apply(xb[(int64_t)ci * P], in_act, in_slope);
  if (LN) {
    // gated convolution + out_act + LayerNormChannels over the C == CIN output channels of this pixel (GatedConv, the
    // nonlinearity and the layer norm that follow it in ConvNet2D, networks.py:480-493): the pixel's outputs stay in
    // registers, mean / biased variance / normalisation with the arithmetic of layernorm_channels_kernel
    constexpr int C = CIN;
    const float* gb = gate_x + b * C * P + p;
    float r[C];
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const float* wv = W + c * CIN;
      const float* wg = W + (C + c) * CIN;
      float av, ag;
      pw_dot2<CIN>(wv, wg, v, av, ag);
      if (bias) { av += bias[c]; ag += bias[C + c]; }
      r[c] = act_apply(gb[(int64_t)c * P] + av * __builtin_amdgcn_rcpf(1.f + __expf(-ag)), out_act, out_slope);
      sum += r[c];
    }
    const float mean = sum / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const float d = r[c] - mean;
      sq += d * d;
    }
    const float den = sqrtf(sq / (float)C + ln.eps);
    float* yb = y + b * C * P + p;
#pragma unroll
    for (int c = 0; c < C; ++c) yb[(int64_t)c * P] = (r[c] - mean) / den * ln.gamma[c] + ln.beta[c];
    return;
  }
  // output channels in groups of 8 (unrolled): the group's weight rows, gate values and stores are independent of each
  // other, so scalar loads, global loads and FMAs of neighbouring channels overlap
  if (GATED) {
    const int C = cout >> 1;
    const float* gb = gate_x + b * C * P + p;
    float* yb = y + b * C * P + p;
    for (int c0 = 0; c0 < C; c0 += 8) {
      float gx[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) gx[j] = (c0 + j < C) ? gb[(int64_t)(c0 + j) * P] : 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = c0 + j;
        if (c < C) {
          const float* wv = W + c * CIN;
          const float* wg = W + (C + c) * CIN;
          float av, ag;
          pw_dot2<CIN>(wv, wg, v, av, ag);
          if (bias) { av += bias[c]; ag += bias[C + c]; }
          // (v_exp_f32 / v_rcp_f32: 1 ulp each)
          yb[(int64_t)c * P] = gx[j] + av * __builtin_amdgcn_rcpf(1.f + __expf(-ag));
        }
      }
    }
  } else {
    float* yb = y + b * cout * P + p;
    // (small batches: gridDim.y > 1 deals the groups of 8 output channels over blocks -- a handful of pixels per CU would
    //  otherwise walk all cout dot products in one thread; large batches keep the loop: the inputs are loaded once)
    const int c_lo = gridDim.y > 1 ? (int)blockIdx.y * 8 : 0;
    const int c_hi = gridDim.y > 1 ? min(c_lo + 8, cout) : cout;
    for (int c0 = c_lo; c0 < c_hi; c0 += 8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int co = c0 + j;
        if (co < cout) {
          const float* wr = W + co * CIN;
          float acc = 0.f;
#pragma unroll
          for (int ci = 0; ci < CIN; ++ci) acc = fmaf(wr[ci], v[ci], acc);
          if (bias) acc += bias[co];
          // USF_ACT_GATE (a data gradient): times the derivative of the (Leaky)ReLU in front of the forward layer, read
          // from that layer's input gate_x [B, cout, P]
          if (out_act == USF_ACT_GATE) acc = gate_apply(acc, gate_x[(b * cout + co) * P + p], out_slope);
          yb[(int64_t)co * P] = act_apply(acc, out_act, out_slope);
        }
      }
    }
  }
}

int pointwise_conv_supported(int64_t cin, int64_t cout, int32_t gated) {
  const bool cin_ok = cin == 8 || cin == 16 || cin == 24 || cin == 32 || cin == 48 || cin == 64;
  return cin_ok && cout >= 1 && cout <= 256 && (!gated || (cout & 1) == 0);
}

int pointwise_conv(const float* x, float* y, int64_t B, int64_t cin, int64_t cout, int64_t P, const float* W, const float* bias,
                   int32_t in_act, float in_slope, int32_t out_act, float out_slope, const float* gate_x,
                   const float* ln_gamma, const float* ln_beta, float ln_eps, hipStream_t stream) {
  if (B == 0 && P > 0 && cin > 0 && cout > 0) return 0;      // (an empty batch: its tensors have no storage to point to)
  const int gated = gate_x != nullptr && out_act != USF_ACT_GATE;
  const bool lnorm = ln_gamma != nullptr;
  if (lnorm && (!ln_beta || !gated || cout != 2 * cin || cin > 32)) {
    set_error("usf_pointwise_conv_f32: the layer-norm form needs gamma and beta, the gated mode, cout == 2 cin and cin <= 32");
    return -2;
  }
  if (B < 0 || P <= 0 || !pointwise_conv_supported(cin, cout, gated)) {
    set_error("usf_pointwise_conv_f32: unsupported sizes (cin in {8, 16, 24, 32, 48, 64}, 1 <= cout <= 256, even cout when gated)");
    return -2;
  }
  if (B == 0) return 0;
  if (!x || !y || !W) { set_error("usf_pointwise_conv_f32: null pointer"); return -1; }
  if (x == y || gate_x == y) { set_error("usf_pointwise_conv_f32: in-place operation is not supported"); return -2; }
  if ((in_act != USF_ACT_NONE && in_act != USF_ACT_LEAKY_RELU) ||
      (out_act != USF_ACT_NONE && out_act != USF_ACT_LEAKY_RELU && !(out_act == USF_ACT_GATE && gate_x && !lnorm))) {
    set_error("usf_pointwise_conv_f32: bad act");
    return -2;
  }
  const int64_t BP = B * P, blocks = (BP + 255) / 256;
  if (blocks > 0x7fffffffLL) { set_error("usf_pointwise_conv_f32: grid too large"); return -3; }
  // plain mode at small batches: fewer than two blocks of pixels per CU -> the output channels are dealt over gridDim.y
  const bool split_c = !gated && !lnorm && blocks < 2 * (int64_t)device_cu_count() && cout > 8;
  const dim3 g((unsigned)blocks, split_c ? (unsigned)((cout + 7) / 8) : 1u), bl(256);
  const PwLn ln{ln_gamma, ln_beta, ln_eps};
#define USF_PW(CI)                                                                                                      \
  do {                                                                                                                  \
    if (gated) hipLaunchKernelGGL((pointwise_conv_kernel<CI, true, false>), g, bl, 0, stream, x, y, (int)cout, P, W, bias,       \
                                  in_act, in_slope, out_act, out_slope, gate_x, BP, ln);                                \
    else hipLaunchKernelGGL((pointwise_conv_kernel<CI, false, false>), g, bl, 0, stream, x, y, (int)cout, P, W, bias, in_act,    \
                            in_slope, out_act, out_slope, gate_x, BP, ln);                                              \
  } while (0)
#define USF_PWLN(CI) hipLaunchKernelGGL((pointwise_conv_kernel<CI, true, true>), g, bl, 0, stream, x, y, (int)cout, P, W, bias, \
                                        in_act, in_slope, out_act, out_slope, gate_x, BP, ln)
  if (lnorm) {
    switch ((int)cin) {
      case 8: USF_PWLN(8); break;
      case 16: USF_PWLN(16); break;
      case 24: USF_PWLN(24); break;
      default: USF_PWLN(32); break;
    }
    return check_launch("usf_pointwise_conv_f32");
  }
  switch ((int)cin) {
    case 8: USF_PW(8); break;
    case 16: USF_PW(16); break;
    case 24: USF_PW(24); break;
    case 32: USF_PW(32); break;
    case 48: USF_PW(48); break;
    default: USF_PW(64); break;
  }
#undef USF_PW
#undef USF_PWLN
  return check_launch("usf_pointwise_conv_f32");
}

// ------------------------------------------------------------------------------------------
// Elementwise pieces of the image-shaped coupling layer, each ONE pass over [B, C, P] (P = H * W):
//   layernorm_channels  y = (a - mean_c a) / sqrt(var_c a + eps) * gamma + beta,  a = act(x)        LayerNormChannels.forward
//                       (networks.py:40-58: mean / biased variance over the channel axis) with the (Leaky)ReLU that
//                       precedes it in ConvNet2D (networks.py:480-493) folded in; the reference chain is 8 passes
//   gated_residual      y = x + val * sigmoid(gate),  (val, gate) = chunk(vg, 2, dim=1)              GatedConv.forward
//                       (networks.py:108-122), 4 passes there
//   masked_residual     y = x + sign * om * t,  om = 1 - mask broadcast over the batch                MaskedCoupling
//                       (transforms.py:277-306), 2 passes there
// One thread per pixel (b, p): channel values strided by P (a wave reads 64 consecutive pixels of one channel plane:
// coalesced within a sample); HBM-bound, 8 / 12 / 12 bytes per element.
// ------------------------------------------------------------------------------------------
template <int CMAX>
__global__ __launch_bounds__(256) void layernorm_channels_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t BP,
                                                                 int C, int64_t P, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, float eps, int act,
                                                                 float slope) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= BP) return;
  const int64_t b = i / P, p = i - b * P;
  const float* xb = x + b * C * P + p;
  float v[CMAX];
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < CMAX; ++c) {
    v[c] = (c < C) ? act_apply(xb[(int64_t)c * P], act, slope) : 0.f;
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
  float* yb = y + b * C * P + p;
#pragma unroll
  for (int c = 0; c < CMAX; ++c)
    if (c < C) yb[(int64_t)c * P] = (v[c] - mean) / den * gamma[c] + beta[c];
}

// Few pixels (a training batch of 32 rows: 1 568 pixels): one thread per pixel fills seven blocks and every thread walks its
// 32 channels alone -- 13.8 us per launch, three times the launch floor, on the chain of dependent launches that bounds such a
// step.  Here EIGHT lanes share a pixel (lane = 8 * channel group + pixel of the wave's 8: channels cg, cg + 8, ...), the two
// sums cross the 8 lanes by shuffles: 8 x the blocks, an eighth of the serial work per thread.  (Sums are added in another
// order than in the one-thread kernel: the two kernels agree to rounding, not bit for bit; a launch uses one of them for all rows.)
template <int CMAX>
__global__ __launch_bounds__(256) void layernorm_channels_small_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t BP,
                                                                       int C, int64_t P, const float* __restrict__ gamma,
                                                                       const float* __restrict__ beta, float eps, int act,
                                                                       float slope) {
  constexpr int LP = 8, NC = CMAX / LP;
  const int lane = threadIdx.x & 63;
  const int pi = lane & 7, cg = lane >> 3;
  const int64_t i = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + pi;
  const bool on = i < BP;
  const int64_t ic = on ? i : BP - 1;
  const int64_t b = ic / P, p = ic - b * P;
  const float* xb = x + b * C * P + p;
  float v[NC];
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < NC; ++k) {
    const int c = cg + LP * k;
    v[k] = (c < C) ? act_apply(xb[(int64_t)c * P], act, slope) : 0.f;
    sum += v[k];
  }
#pragma unroll
  for (int o = 8; o < 64; o <<= 1) sum += __shfl_xor(sum, o, 64);
  const float mean = sum / (float)C;
  float sq = 0.f;
#pragma unroll
  for (int k = 0; k < NC; ++k) {
    const float d = (cg + LP * k < C) ? v[k] - mean : 0.f;
    sq += d * d;
  }
#pragma unroll
  for (int o = 8; o < 64; o <<= 1) sq += __shfl_xor(sq, o, 64);
  const float den = sqrtf(sq / (float)C + eps);
  float* yb = y + b * C * P + p;
#pragma unroll
  for (int k = 0; k < NC; ++k) {
    const int c = cg + LP * k;
    if (on && c < C) yb[(int64_t)c * P] = (v[k] - mean) / den * gamma[c] + beta[c];
  }
}

__global__ __launch_bounds__(256) void gated_residual_kernel(const float* __restrict__ x, const float* __restrict__ vg,
                                                             float* __restrict__ y, int64_t total, int64_t CP) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t b = e / CP, r = e - b * CP;
    const float val = vg[b * 2 * CP + r], gate = vg[b * 2 * CP + CP + r];
    y[e] = x[e] + val * (1.f / (1.f + expf(-gate)));
  }
}

__global__ __launch_bounds__(256) void masked_residual_kernel(const float* __restrict__ x, const float* __restrict__ t,
                                                              const float* __restrict__ om, float sign,
                                                              float* __restrict__ y, int64_t total, int64_t CP) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256)
    y[e] = (x ? x[e] : 0.f) + sign * (om[e % CP] * t[e]);
}

int layernorm_channels(const float* x, float* y, int64_t B, int64_t C, int64_t P, const float* gamma, const float* beta,
                       float eps, int32_t act, float slope, hipStream_t stream) {
  if (B < 0 || C <= 0 || P <= 0 || C > 64) { set_error("usf_layernorm_channels_f32: bad sizes (C must be 1..64)"); return -2; }
  if (B == 0) return 0;
  if (!x || !y || !gamma || !beta) { set_error("usf_layernorm_channels_f32: null pointer"); return -1; }
  if (act != USF_ACT_NONE && act != USF_ACT_LEAKY_RELU) { set_error("usf_layernorm_channels_f32: bad act"); return -2; }
  const int64_t BP = B * P, blocks = (BP + 255) / 256;
  if (blocks > 0x7fffffffLL) { set_error("usf_layernorm_channels_f32: grid too large"); return -3; }
  if (BP <= 16384 && C > 8) {                  // few pixels: eight lanes per pixel (see layernorm_channels_small_kernel)
    const dim3 gs((unsigned)((BP + 31) / 32)), bs(256);
    if (C <= 16) hipLaunchKernelGGL(layernorm_channels_small_kernel<16>, gs, bs, 0, stream, x, y, BP, (int)C, P, gamma, beta, eps, act, slope);
    else if (C <= 32) hipLaunchKernelGGL(layernorm_channels_small_kernel<32>, gs, bs, 0, stream, x, y, BP, (int)C, P, gamma, beta, eps, act, slope);
    else hipLaunchKernelGGL(layernorm_channels_small_kernel<64>, gs, bs, 0, stream, x, y, BP, (int)C, P, gamma, beta, eps, act, slope);
    return check_launch("usf_layernorm_channels_f32");
  }
  const dim3 g((unsigned)blocks), b(256);
  if (C <= 8) hipLaunchKernelGGL(layernorm_channels_kernel<8>, g, b, 0, stream, x, y, BP, (int)C, P, gamma, beta, eps, act, slope);
  else if (C <= 16) hipLaunchKernelGGL(layernorm_channels_kernel<16>, g, b, 0, stream, x, y, BP, (int)C, P, gamma, beta, eps, act, slope);
  else if (C <= 32) hipLaunchKernelGGL(layernorm_channels_kernel<32>, g, b, 0, stream, x, y, BP, (int)C, P, gamma, beta, eps, act, slope);
  else hipLaunchKernelGGL(layernorm_channels_kernel<64>, g, b, 0, stream, x, y, BP, (int)C, P, gamma, beta, eps, act, slope);
  return check_launch("usf_layernorm_channels_f32");
}

int gated_residual(const float* x, const float* vg, float* y, int64_t B, int64_t CP, hipStream_t stream) {
  if (B < 0 || CP <= 0) { set_error("usf_gated_residual_f32: bad sizes"); return -2; }
  if (B == 0) return 0;
  if (!x || !vg || !y) { set_error("usf_gated_residual_f32: null pointer"); return -1; }
  int64_t blocks = (B * CP + 255) / 256;
  if (blocks > 256 * 64) blocks = 256 * 64;
  hipLaunchKernelGGL(gated_residual_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, vg, y, B * CP, CP);
  return check_launch("usf_gated_residual_f32");
}

int masked_residual(const float* x, const float* t, const float* om, float sign, float* y, int64_t B, int64_t CP,
                    hipStream_t stream) {
  if (B < 0 || CP <= 0) { set_error("usf_masked_residual_f32: bad sizes"); return -2; }
  if (B == 0) return 0;
  if (!t || !om || !y) { set_error("usf_masked_residual_f32: null pointer"); return -1; }
  int64_t blocks = (B * CP + 255) / 256;
  if (blocks > 256 * 64) blocks = 256 * 64;
  hipLaunchKernelGGL(masked_residual_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, t, om, sign, y, B * CP, CP);
  return check_launch("usf_masked_residual_f32");
}

// ------------------------------------------------------------------------------------------
// Row pass of the vector ConvNet conditioner's blocks (networks.py:206-245: GatedMLP, LayerNormVector; ConvNet.__init__
// vector branch :287-308): for each row of [M, C]
//     r = skip + vg[:, :C] * sigmoid(vg[:, gate_off : gate_off + C])      (vg == NULL: r = skip)
//     y = (r - mean r) / sqrt(var r + eps) * gamma + beta                 (gamma == NULL: y = r)
//     out = y,  out_act = f(y)          (either may be NULL; f = the (Leaky)ReLU in front of the next block's Linear)
// columns [C, c_pad) of out / out_act are written as zeros (the operand padding of the linear kernels).
// One wave per row, the row in registers (C <= 64 * NIT); mean and biased variance by two wave reductions, as
// torch.nn.LayerNorm computes them.  HBM-bound: 4 * (3..5) * C bytes per row.
// ------------------------------------------------------------------------------------------
template <int NIT>
__global__ __launch_bounds__(256) void gated_norm_rows_kernel(usf_gated_norm_desc d) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= d.M) return;
  const int C = (int)d.C, CP = (int)d.c_pad;
  const float* sk = d.skip + row * d.ld_skip;
  const float* vg = d.vg ? d.vg + row * d.ld_vg : nullptr;
  float r[NIT];
  float sum = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int c = it * 64 + lane;
    float v = 0.f;
    if (c < C) {
      v = sk[c];
      if (vg) v += vg[c] * (1.f / (1.f + expf(-vg[d.gate_off + c])));
    }
    r[it] = v;
    sum += v;
  }
  if (d.gamma) {
    const float mean = wave_sum(sum) / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const float dv = (it * 64 + lane < C) ? r[it] - mean : 0.f;
      sq += dv * dv;
    }
    const float rstd = 1.f / sqrtf(wave_sum(sq) / (float)C + d.eps);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = it * 64 + lane;
      if (c < C) r[it] = (r[it] - mean) * rstd * d.gamma[c] + d.beta[c];
    }
  }
  float* o = d.out ? d.out + row * d.ld_out : nullptr;
  float* oa = d.out_act ? d.out_act + row * d.ld_act : nullptr;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int c = it * 64 + lane;
    if (c < CP) {
      const float v = (c < C) ? r[it] : 0.f;
      if (o) o[c] = v;
      if (oa) oa[c] = act_apply(v, d.act, d.slope);
    }
  }
}

int gated_norm_rows(const usf_gated_norm_desc* d, hipStream_t stream) {
  if (!d) { set_error("usf_gated_norm_rows_f32: null descriptor"); return -1; }
  if (d->M < 0 || d->C <= 0 || d->c_pad < d->C || d->c_pad > 4096) {
    set_error("usf_gated_norm_rows_f32: bad sizes (1 <= C <= c_pad <= 4096)");
    return -2;
  }
  if (d->M == 0) return 0;
  if (!d->skip || (!d->out && !d->out_act) || ((d->gamma == nullptr) != (d->beta == nullptr))) {
    set_error("usf_gated_norm_rows_f32: null pointer (skip, one of out / out_act, gamma and beta together)");
    return -1;
  }
  if (d->ld_skip < d->C || (d->out && d->ld_out < d->c_pad) || (d->out_act && d->ld_act < d->c_pad) ||
      (d->vg && (d->gate_off < 0 || d->ld_vg < d->gate_off + d->C))) {
    set_error("usf_gated_norm_rows_f32: row stride shorter than the row");
    return -2;
  }
  if (d->act != USF_ACT_NONE && d->act != USF_ACT_LEAKY_RELU) { set_error("usf_gated_norm_rows_f32: bad act"); return -2; }
  const int64_t blocks = (d->M + 3) / 4;
  if (blocks > 0x7fffffffLL) { set_error("usf_gated_norm_rows_f32: grid too large"); return -3; }
  const dim3 g((unsigned)blocks), b(256);
  if (d->c_pad <= 256) hipLaunchKernelGGL(gated_norm_rows_kernel<4>, g, b, 0, stream, *d);
  else if (d->c_pad <= 1024) hipLaunchKernelGGL(gated_norm_rows_kernel<16>, g, b, 0, stream, *d);
  else hipLaunchKernelGGL(gated_norm_rows_kernel<64>, g, b, 0, stream, *d);
  return check_launch("usf_gated_norm_rows_f32");
}

// ------------------------------------------------------------------------------------------
// usf_gated_norm_rows_bwd_f32: the backward twin.  With r, mean, rstd, xh = (r - mean) rstd recomputed from (skip, vg):
//   g  = dy * gamma                                         (no layer norm: dr = dy)
//   dr = (g - mean_c g - xh * mean_c (g xh)) * rstd         torch.nn.LayerNorm's backward over the C real columns
//   d_skip = dr;   d_val = dr * s,  d_gate = dr * val * s (1 - s),  s = sigmoid(gate)
//   dy_xh  = dy * xh    (optional: its column sums are dgamma; dbeta = the column sums of dy)
// Padding columns [C, c_pad) of d_skip / d_vg (both halves) / dy_xh are written as zeros: they are operands of the
// weight-gradient and data-gradient GEMMs.  One wave per row, the row in registers.
// ------------------------------------------------------------------------------------------
template <int NIT>
__global__ __launch_bounds__(256) void gated_norm_rows_bwd_kernel(usf_gated_norm_bwd_desc d) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= d.M) return;
  const int C = (int)d.C, CP = (int)d.c_pad;
  const float* sk = d.skip + row * d.ld_skip;
  const float* vg = d.vg ? d.vg + row * d.ld_vg : nullptr;
  const float* dy = d.dy + row * d.ld_dy;
  float r[NIT], val[NIT], sg[NIT], g[NIT];
  float sum = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int c = it * 64 + lane;
    float v = 0.f;
    val[it] = 0.f;
    sg[it] = 0.f;
    g[it] = 0.f;
    if (c < C) {
      v = sk[c];
      if (vg) {
        val[it] = vg[c];
        sg[it] = 1.f / (1.f + expf(-vg[d.gate_off + c]));
        v += val[it] * sg[it];
      }
      g[it] = dy[c];
    }
    r[it] = v;
    sum += v;
  }
  float* dyx = d.dy_xh ? d.dy_xh + row * d.ld_dy_xh : nullptr;
  if (d.gamma) {
    const float mean = wave_sum(sum) / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const float dv = (it * 64 + lane < C) ? r[it] - mean : 0.f;
      r[it] = dv;
      sq += dv * dv;
    }
    const float rstd = 1.f / sqrtf(wave_sum(sq) / (float)C + d.eps);
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = it * 64 + lane;
      r[it] *= rstd;                                            // xh
      if (dyx && c < CP) dyx[c] = g[it] * r[it];
      g[it] = (c < C) ? g[it] * d.gamma[c] : 0.f;
      m1 += g[it];
      m2 += g[it] * r[it];
    }
    m1 = wave_sum(m1) / (float)C;
    m2 = wave_sum(m2) / (float)C;
#pragma unroll
    for (int it = 0; it < NIT; ++it) g[it] = (it * 64 + lane < C) ? (g[it] - m1 - r[it] * m2) * rstd : 0.f;   // dr
  }
  float* ds = d.d_skip + row * d.ld_d_skip;
  float* dv = d.d_vg ? d.d_vg + row * d.ld_d_vg : nullptr;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int c = it * 64 + lane;
    if (c < CP) {
      ds[c] = g[it];
      if (dv) {
        dv[c] = g[it] * sg[it];
        dv[d.gate_off + c] = g[it] * val[it] * (sg[it] * (1.f - sg[it]));
      }
    }
  }
}

int gated_norm_rows_bwd(const usf_gated_norm_bwd_desc* d, hipStream_t stream) {
  if (!d) { set_error("usf_gated_norm_rows_bwd_f32: null descriptor"); return -1; }
  if (d->M < 0 || d->C <= 0 || d->c_pad < d->C || d->c_pad > 4096) {
    set_error("usf_gated_norm_rows_bwd_f32: bad sizes (1 <= C <= c_pad <= 4096)");
    return -2;
  }
  if (d->M == 0) return 0;
  if (!d->skip || !d->dy || !d->d_skip || ((d->vg == nullptr) != (d->d_vg == nullptr)) || (d->dy_xh && !d->gamma)) {
    set_error("usf_gated_norm_rows_bwd_f32: null pointer (skip, dy, d_skip; vg and d_vg together; dy_xh only with gamma)");
    return -1;
  }
  if (d->ld_skip < d->C || d->ld_dy < d->C || d->ld_d_skip < d->c_pad || (d->dy_xh && d->ld_dy_xh < d->c_pad) ||
      (d->vg && (d->gate_off < d->c_pad || d->ld_vg < d->gate_off + d->C || d->ld_d_vg < d->gate_off + d->c_pad))) {
    set_error("usf_gated_norm_rows_bwd_f32: row stride shorter than the row (gate_off >= c_pad)");
    return -2;
  }
  const int64_t blocks = (d->M + 3) / 4;
  if (blocks > 0x7fffffffLL) { set_error("usf_gated_norm_rows_bwd_f32: grid too large"); return -3; }
  const dim3 g((unsigned)blocks), b(256);
  if (d->c_pad <= 256) hipLaunchKernelGGL(gated_norm_rows_bwd_kernel<4>, g, b, 0, stream, *d);
  else if (d->c_pad <= 1024) hipLaunchKernelGGL(gated_norm_rows_bwd_kernel<16>, g, b, 0, stream, *d);
  else hipLaunchKernelGGL(gated_norm_rows_bwd_kernel<64>, g, b, 0, stream, *d);
  return check_launch("usf_gated_norm_rows_bwd_f32");
}

}  // namespace usf
