// usf_mfma_probe: what this part SUSTAINS on the bf16 matrix cores for the planes GEMM's instruction mix -- the ceiling the
// bench prints beside the nominal roof (bench.py: roofline.sustained_peak).  A register-only loop: no LDS, no global memory
// inside it, the occupancy (512 threads, two waves per SIMD) and the accumulator tiling (10 feature tiles x 2 batch tiles of
// v_mfma_f32_16x16x32_bf16, six products per fp32-equivalent product, 120 MFMAs per "slab") of gemm_planes_kernel<3, 5>.
// Under dense MFMA load the chip is power-bound (its clock drops below the 2.4 GHz the nominal 2.5 PFLOP/s assume), so no
// schedule of the real kernel can beat this number; how close it comes is the figure of merit (tools/exp_mfma_peak.hip
// holds the longer study: with LDS fragment reads, staging stores, operand loads added one by one).
#include "usf_common.h"

namespace usf {

typedef __bf16 pb_bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(512, 2) void mfma_probe_kernel(const float* __restrict__ src, float* __restrict__ sink, int iters,
                                                            unsigned long long* __restrict__ clk) {
  const int tid = threadIdx.x;
  unsigned long long clk_c0 = 0, clk_r0 = 0;
  if (clk) { clk_c0 = __builtin_amdgcn_s_memtime(); clk_r0 = __builtin_amdgcn_s_memrealtime(); }
  pb_bf16x8 a[2][3], w[2][3];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        a[b][q][e] = (__bf16)src[(tid + 8 * (3 * b + q) + e) & 1023];
        w[b][q][e] = (__bf16)src[(tid * 3 + 8 * (3 * b + q) + e + 77) & 1023];
      }
  f32x4 acc[10][2];
#pragma unroll
  for (int t = 0; t < 10; ++t)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[t][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 10; ++t) {
      const int f = t & 1;
#define USF_PB(P, Q)                                                                              \
  acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[f][P], a[0][Q], acc[t][0], 0, 0, 0);      \
  acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[f][P], a[1][Q], acc[t][1], 0, 0, 0)
      USF_PB(2, 0); USF_PB(1, 1); USF_PB(0, 2); USF_PB(1, 0); USF_PB(0, 1); USF_PB(0, 0);
#undef USF_PB
    }
  }
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < 10; ++t)
#pragma unroll
    for (int b = 0; b < 2; ++b) s += acc[t][b][0] + acc[t][b][1] + acc[t][b][2] + acc[t][b][3];
  if (s == 12345.678f) sink[0] = s;             // (never true: keeps the loop alive)
  if (clk && tid == 0) {                         // usf_set_clock_buffer: the clock this loop runs at
    atomicAdd(clk, __builtin_amdgcn_s_memtime() - clk_c0);
    atomicAdd(clk + 1, __builtin_amdgcn_s_memrealtime() - clk_r0);
  }
}

// one launch of `blocks` blocks (0: two per CU); returns the bf16 MFMA flops it performs through *flops_out
int mfma_probe(const float* src1024, float* sink, int64_t iters, int64_t blocks, double* flops_out, hipStream_t stream) {
  if (!src1024 || !sink || iters <= 0 || iters > (1 << 24) || blocks < 0 || blocks > (1 << 20)) { set_error("usf_mfma_probe: bad arguments"); return -1; }
  if (blocks == 0) blocks = 2 * (int64_t)device_cu_count();
  mfma_probe_kernel<<<(unsigned)blocks, 512, 0, stream>>>(src1024, sink, (int)iters, clock_buffer());
  if (flops_out) *flops_out = (double)blocks * 8.0 * (double)iters * 120.0 * (2.0 * 16 * 16 * 32);
  return check_launch("usf_mfma_probe");
}

}  // namespace usf
