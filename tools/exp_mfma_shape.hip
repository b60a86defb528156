// Power-bound sustained rate of the matrix cores by instruction SHAPE (tuning aid): register-only loops, random
// non-zero operands (switching activity matters on a power-bound chip), 2 waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/exp_mfma_shape.hip -o tools/exp_mfma_shape && tools/exp_mfma_shape
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// NOPS distinct operand pairs cycle through the loop so that operand buses toggle as in a real K loop
template <int SHAPE, bool F16>
__global__ __launch_bounds__(512, 2) void k(const float* __restrict__ src, float* __restrict__ out, int iters) {
  const int tid = threadIdx.x;
  typedef typename std::conditional<F16, f16x8, bf16x8>::type v8;
  v8 a[6], b[6];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      a[i][j] = (decltype(a[i][j] + a[i][j]))src[(tid * 8 + j + 61 * i + blockIdx.x) & 0xffff];
      b[i][j] = (decltype(a[i][j] + a[i][j]))src[(tid * 8 + j + 977 * i + 3 * blockIdx.x + 5) & 0xffff];
    }
  float s = 0.f;
  if (SHAPE == 32) {
    f32x16 acc[5];
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int t = 0; t < 5; ++t) {
          if constexpr (F16) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + t) % 6], b[i], acc[t], 0, 0, 0);
          else acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i + t) % 6], b[i], acc[t], 0, 0, 0);
        }
    }
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
      for (int j = 0; j < 16; ++j) s += acc[t][j];
  } else {
    f32x4 acc[20];
#pragma unroll
    for (int t = 0; t < 20; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int t = 0; t < 20; ++t) {
          if constexpr (F16) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(i + t) % 6], b[(i + (t >> 2)) % 6], acc[t], 0, 0, 0);
          else acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(i + t) % 6], b[(i + (t >> 2)) % 6], acc[t], 0, 0, 0);
        }
    }
#pragma unroll
    for (int t = 0; t < 20; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) s += acc[t][j];
  }
  out[blockIdx.x * 512 + tid] = s;
}

template <int SHAPE, bool F16>
static void run(const char* name, const float* src, float* out) {
  // flops per iteration and wave: SHAPE 32: 30 MFMAs x 32*32*16*2; SHAPE 16: 120 x 16*16*32*2  (the same)
  const int iters = 4000, blocks = 256, reps = 20;
  const double flops = (double)blocks * 8 * iters * 30.0 * 32 * 32 * 16 * 2;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k<SHAPE, F16>), dim3(blocks), dim3(512), 0, 0, src, out, iters);
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<SHAPE, F16>), dim3(blocks), dim3(512), 0, 0, src, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  // pipe-bound time at clock f: iters*30 MFMAs x 32 cycles (SHAPE 32: 16 passes x ... ) per wave, 2 waves per SIMD
  printf("%-22s %.3f ms  %.0f TFLOP/s  (implied clock at 100%% pipe: %.0f MHz)\n", name, ms, flops / ms / 1e9,
         flops / ms / 1e9 / 2500.0 * 2400.0);
}
int main() {
  float *src, *out;
  hipMalloc(&src, 65536 * 4); hipMalloc(&out, 256 * 512 * 4);
  std::vector<float> h(65536);
  unsigned s = 99;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.0f - 0.5f; }
  hipMemcpy(src, h.data(), 65536 * 4, hipMemcpyHostToDevice);
  for (int r = 0; r < 2; ++r) {
    run<32, false>("bf16 32x32x16", src, out);
    run<16, false>("bf16 16x16x32", src, out);
    run<32, true>("f16  32x32x16", src, out);
    run<16, true>("f16  16x16x32", src, out);
  }
  return 0;
}
