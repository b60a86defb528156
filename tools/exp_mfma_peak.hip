// What does the chip sustain on the bf16 matrix cores for the bf16x3 instruction mix?  Register-only loops
// (no LDS, no global memory inside the loop) at the occupancy of linear_bf16x3_kernel:
//   mode 0: 6 MFMA per accumulator tile per step (the split-precision product), operands fixed in registers
//   mode 1: mode 0 + the fp32 -> 3 x bf16 operand split of 16 floats per lane per 2 steps (VALU beside MFMA)
//   mode 2: mode 1 + 3 ds_read_b128 per tile per step from a 60 KB LDS image
//   mode 3: mode 2 + one __syncthreads per slab (double-buffered image)
//   mode 4: mode 3 + the weight-slab ds_write_b128 staging (4 per thread per slab, from registers)
//   mode 5: mode 4 + the weight-slab global loads (L2-resident planes) that feed the staging
//   mode 6: mode 5 + the activation fragment global loads (4 x 16 B per lane per slab) that feed the split
//   mode 7: mode 6 with line-coalesced activation loads (8 lanes = one 128-B row piece); values land in the
//           wrong lanes (no transpose) -- instruction-count-equivalent upper bound
//   mode 8: mode 7 + the wave-private LDS transpose (4 ds_write_b128 + 4 ds_read_b128 per lane per slab)
// Prints sustained TFLOP/s (bf16 MFMA flops), the fp32-equivalent rate (/6) and the shader clock estimate.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split3(const f32x4 x0, const f32x4 x1, bf16x8& p1, bf16x8& p2, bf16x8& p3) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float x = (j < 4) ? x0[j & 3] : x1[j & 3];
    const __bf16 h = (__bf16)x;
    const float r = x - (float)h;
    const __bf16 m = (__bf16)r;
    const float r2 = r - (float)m;
    p1[j] = h; p2[j] = m; p3[j] = (__bf16)r2;
  }
}

template <int TN, int MODE, bool ILV>
__global__ __launch_bounds__(512, 2) void mfma_mix(const float* __restrict__ src, float* __restrict__ out, int iters,
                                                   unsigned long long* clk, const float* __restrict__ Abig,
                                                   const __bf16* __restrict__ Wp, int flags) {
  __shared__ __attribute__((aligned(16))) float lds2[2][8192];   // 2 x 32 KB (30 KB image + surplus slots)
  __shared__ __attribute__((aligned(16))) float atr[MODE >= 8 ? 8 * 32 * 36 : 4];   // per wave 32 rows x (32 + 4) floats
  float* lds = &lds2[0][0];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 15360; i += 512) lds[i] = src[(i * 7 + blockIdx.x) & 0xffff];
  __syncthreads();
  constexpr int BN = 160, NSLOT = 3 * 4 * BN, NT = 512, NWV = (NSLOT + NT - 1) / NT, NRG = BN / 8;
  const int bid = blockIdx.x, xcd = bid & 7, seq = bid >> 3;
  const int bn = seq % 5;
  const int panel = (flags & 1) ? 0 : (flags & 4) ? bid % 104 : (seq / 5) * 8 + xcd;
  const int n0 = bn * BN;
  const int64_t ldwp = 800, plane_stride = 800 * 800, lda = 800;
  const float* arow = Abig + (int64_t)(panel * 256 + (tid >> 6) * 32 + (lane & 31)) * lda;
  auto slot_of = [&](int idx, int& plane, int& chunk, int& row) {
    const int g32 = idx >> 5;
    plane = g32 / NRG; row = (g32 % NRG) * 8 + (idx & 7); chunk = (idx >> 3) & 3;
  };
  const float* arow8 = Abig + (int64_t)(panel * 256 + (tid >> 6) * 32 + (lane >> 3)) * lda + 4 * (lane & 7);
  f32x4 wst[NWV];
#pragma unroll
  for (int i = 0; i < NWV; ++i) wst[i] = *reinterpret_cast<const f32x4*>(src + ((tid * 4 + i * 2048) & 0xfff0));
  f32x4 a[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const f32x4*>(src + ((tid * 16 + i * 4 + blockIdx.x * 64) & 0xfff0) + 4 * 0);
  bf16x8 pc[2][3], w[3];
  split3(a[0], a[1], pc[0][0], pc[0][1], pc[0][2]);
  split3(a[2], a[3], pc[1][0], pc[1][1], pc[1][2]);
  split3(a[1], a[2], w[0], w[1], w[2]);
  f32x16 acc[TN];
#pragma unroll
  for (int t = 0; t < TN; ++t)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;
  const unsigned long long c0 = __builtin_readcyclecounter();
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    const int buf = (MODE >= 3) ? (it & 1) : 0;
    const int k1 = ((it + 1) % 24) * 32;
    f32x4 an[4];
    if (MODE >= 5) {
#pragma unroll
      for (int i = 0; i < NWV; ++i) {
        int pl, ch, r;
        slot_of(min(tid + NT * i, NSLOT - 1), pl, ch, r);
        wst[i] = *reinterpret_cast<const f32x4*>(Wp + pl * plane_stride + (int64_t)(n0 + r) * ldwp + k1 + 8 * ch);
      }
    }
    if (MODE == 6) {
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int u = 0; u < 2; ++u) an[2 * s + u] = *reinterpret_cast<const f32x4*>(arow + k1 + 16 * s + 8 * (lane >> 5) + 4 * u);
    }
    if (MODE >= 7) {
#pragma unroll
      for (int i = 0; i < 4; ++i) an[i] = *reinterpret_cast<const f32x4*>(arow8 + (int64_t)(8 * i) * lda + k1);
    }
    __builtin_amdgcn_sched_barrier(0);
    lds = &lds2[buf][0];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int t = 0; t < TN; ++t) {
        bf16x8 w1 = w[0], w2 = w[1], w3 = w[2];
        if (MODE >= 2) {
          const float* wl = lds + 4 * ((lane >> 5) * 160 + (lane & 31));
          w1 = *reinterpret_cast<const bf16x8*>(wl + 4 * ((0 * 4 + 2 * s) * 160 + t * 32));
          w2 = *reinterpret_cast<const bf16x8*>(wl + 4 * ((1 * 4 + 2 * s) * 160 + t * 32));
          w3 = *reinterpret_cast<const bf16x8*>(wl + 4 * ((2 * 4 + 2 * s) * 160 + t * 32));
        }
        if (ILV && t + 1 < TN && (t & 1) == 0) {
          bf16x8 v1 = w[0], v2 = w[1], v3 = w[2];
          if (MODE >= 2) {
            const float* wl = lds + 4 * ((lane >> 5) * 160 + (lane & 31));
            v1 = *reinterpret_cast<const bf16x8*>(wl + 4 * ((0 * 4 + 2 * s) * 160 + (t + 1) * 32));
            v2 = *reinterpret_cast<const bf16x8*>(wl + 4 * ((1 * 4 + 2 * s) * 160 + (t + 1) * 32));
            v3 = *reinterpret_cast<const bf16x8*>(wl + 4 * ((2 * 4 + 2 * s) * 160 + (t + 1) * 32));
          }
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w3, pc[s][0], acc[t], 0, 0, 0);
          acc[t + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v3, pc[s][0], acc[t + 1], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2, pc[s][1], acc[t], 0, 0, 0);
          acc[t + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v2, pc[s][1], acc[t + 1], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, pc[s][2], acc[t], 0, 0, 0);
          acc[t + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, pc[s][2], acc[t + 1], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2, pc[s][0], acc[t], 0, 0, 0);
          acc[t + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v2, pc[s][0], acc[t + 1], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, pc[s][1], acc[t], 0, 0, 0);
          acc[t + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, pc[s][1], acc[t + 1], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, pc[s][0], acc[t], 0, 0, 0);
          acc[t + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, pc[s][0], acc[t + 1], 0, 0, 0);
          ++t;
          continue;
        }
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w3, pc[s][0], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2, pc[s][1], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, pc[s][2], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2, pc[s][0], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, pc[s][1], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, pc[s][0], acc[t], 0, 0, 0);
      }
    }
    if (MODE >= 6 && MODE < 8) {
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = an[i];
    }
    if (MODE >= 8) {
      float* tw = atr + (tid >> 6) * (32 * 36);
#pragma unroll
      for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(tw + ((lane >> 3) + 8 * i) * 36 + 4 * (lane & 7)) = an[i];
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int u = 0; u < 2; ++u) a[2 * s + u] = *reinterpret_cast<const f32x4*>(tw + (lane & 31) * 36 + 16 * s + 8 * (lane >> 5) + 4 * u);
    }
    if (MODE >= 1) {
      // perturb the fp32 source so the split cannot be hoisted (xor of a mantissa bit with the counter)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) a[i][j] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, a[i][j]) ^ (unsigned)(it & 255));
      split3(a[0], a[1], pc[0][0], pc[0][1], pc[0][2]);
      split3(a[2], a[3], pc[1][0], pc[1][1], pc[1][2]);
    }
    if (MODE >= 4) {
#pragma unroll
      for (int i = 0; i < NWV; ++i) {
        int pl, ch, r;
        slot_of(tid + NT * i, pl, ch, r);
        const int sl = (tid + NT * i < NSLOT) ? (pl * 4 + ch) * BN + r : tid + NT * i;
        *reinterpret_cast<f32x4*>(&lds2[buf ^ 1][4 * sl]) = wst[i];
      }
    }
    if (MODE >= 3 && !(flags & 2)) __syncthreads();
  }
  const unsigned long long c1 = __builtin_readcyclecounter();
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < TN; ++t)
#pragma unroll
    for (int j = 0; j < 16; ++j) s += acc[t][j];
  out[blockIdx.x * 512 + tid] = s;
  if (tid == 0 && blockIdx.x < 512) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = t1 - t0; }
}

template <int TN, int MODE, bool ILV = false>
static void run(const float* src, float* out, unsigned long long* clk, int iters, const float* Abig, const __bf16* Wp, int flags = 0, int dyn = 0) {
  hipFuncSetAttribute((const void*)mfma_mix<TN, MODE, ILV>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 512;     // 2 per CU
  for (int i = 0; i < 30; ++i) hipLaunchKernelGGL((mfma_mix<TN, MODE, ILV>), dim3(blocks), dim3(512), dyn, 0, src, out, iters, clk, Abig, Wp, flags);
  hipEventRecord(e0);
  const int reps = 30;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((mfma_mix<TN, MODE, ILV>), dim3(blocks), dim3(512), dyn, 0, src, out, iters, clk, Abig, Wp, flags);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  std::vector<unsigned long long> h(1024);
  hipMemcpy(h.data(), clk, 1024 * 8, hipMemcpyDeviceToHost);
  double cyc = 0, rt = 0;
  for (int i = 0; i < 512; ++i) { cyc += h[2 * i]; rt += h[2 * i + 1]; }
  const double flops = (double)blocks * 8 * iters * 2 * TN * 6 * 2.0 * 32 * 32 * 16;
  const double tf = flops / (ms * 1e-3) / 1e12;
  printf("TN=%d mode=%d ilv=%d dynLDS=%d flags=%d  %.3f ms  %.1f TF bf16 (%.1f fp32-equiv)  clock ~ %.0f MHz (cycles / 100 MHz realtime)\n", TN, MODE, (int)ILV, dyn, flags, ms, tf,
         tf / 6, cyc / rt * 100.0);
}

// same output tile per wave (32 batch rows x 160 features), v_mfma_f32_16x16x32_bf16: 10 x 2 tiles, 6 terms each
template <int MODE>
__global__ __launch_bounds__(512, 2) void mfma_mix16(const float* __restrict__ src, float* __restrict__ out, int iters,
                                                     unsigned long long* clk) {
  __shared__ __attribute__((aligned(16))) float lds[7680];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 7680; i += 512) lds[i] = src[(i * 7 + blockIdx.x) & 0xffff];
  __syncthreads();
  f32x4 a[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const f32x4*>(src + ((tid * 16 + i * 4 + blockIdx.x * 64) & 0xfff0));
  bf16x8 pc[2][3], w[3];
  split3(a[0], a[1], pc[0][0], pc[0][1], pc[0][2]);
  split3(a[2], a[3], pc[1][0], pc[1][1], pc[1][2]);
  split3(a[1], a[2], w[0], w[1], w[2]);
  f32x4 acc[10][2];
#pragma unroll
  for (int t = 0; t < 10; ++t)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[t][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  const unsigned long long c0 = __builtin_readcyclecounter();
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 10; ++t) {
      bf16x8 w1 = w[0], w2 = w[1], w3 = w[2];
      if (MODE >= 2) {
        const float* wl = lds + 4 * ((lane >> 4) * 160 + (lane & 15));
        w1 = *reinterpret_cast<const bf16x8*>(wl + 4 * (0 * 640 + t * 16));
        w2 = *reinterpret_cast<const bf16x8*>(wl + 4 * (1 * 640 + t * 16));
        w3 = *reinterpret_cast<const bf16x8*>(wl + 4 * (2 * 640 + t * 16));
      }
#define M16(W, P) acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W, pc[0][P], acc[t][0], 0, 0, 0); acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W, pc[1][P], acc[t][1], 0, 0, 0)
      M16(w3, 0); M16(w2, 1); M16(w1, 2); M16(w2, 0); M16(w1, 1); M16(w1, 0);
    }
    if (MODE >= 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) a[i][j] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, a[i][j]) ^ (unsigned)(it & 255));
      split3(a[0], a[1], pc[0][0], pc[0][1], pc[0][2]);
      split3(a[2], a[3], pc[1][0], pc[1][1], pc[1][2]);
    }
  }
  const unsigned long long c1 = __builtin_readcyclecounter();
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 10; ++t) s4 = s4 + acc[t][0] + acc[t][1];
  out[blockIdx.x * 512 + tid] = s4[0] + s4[1] + s4[2] + s4[3];
  if (tid == 0 && blockIdx.x < 512) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = t1 - t0; }
}

template <int MODE>
static void run16(const float* src, float* out, unsigned long long* clk, int iters, int dyn) {
  hipFuncSetAttribute((const void*)mfma_mix16<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 512;
  for (int i = 0; i < 30; ++i) hipLaunchKernelGGL((mfma_mix16<MODE>), dim3(blocks), dim3(512), dyn, 0, src, out, iters, clk);
  hipEventRecord(e0);
  const int reps = 30;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((mfma_mix16<MODE>), dim3(blocks), dim3(512), dyn, 0, src, out, iters, clk);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  std::vector<unsigned long long> h(1024);
  hipMemcpy(h.data(), clk, 1024 * 8, hipMemcpyDeviceToHost);
  double cyc = 0, rt = 0;
  for (int i = 0; i < 512; ++i) { cyc += h[2 * i]; rt += h[2 * i + 1]; }
  const double flops = (double)blocks * 8 * iters * 20 * 6 * 2.0 * 16 * 16 * 32;
  const double tf = flops / (ms * 1e-3) / 1e12;
  printf("16x16x32 mode=%d dynLDS=%d  %.3f ms  %.1f TF bf16 (%.1f fp32-equiv)  clock ~ %.0f MHz\n", MODE, dyn, ms, tf, tf / 6, cyc / rt * 100.0);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 400;
  float *src, *out; unsigned long long* clk;
  hipMalloc(&src, 65536 * 4 + 64); hipMalloc(&out, 512 * 512 * 4); hipMalloc(&clk, 1024 * 8);
  std::vector<float> h(65536 + 16);
  srand(1);
  for (auto& v : h) v = ((rand() % 20001) - 10000) * 1e-3f * (1.f + (rand() % 1000) * 1e-6f);
  hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  float* Abig; __bf16* Wp;
  const size_t abytes = (size_t)512 / 5 * 8 * 256 * 800 * 4 + (1 << 20), wbytes = (size_t)3 * 800 * 800 * 2 + 4096;
  hipMalloc(&Abig, abytes); hipMalloc(&Wp, wbytes);
  for (size_t o = 0; o < abytes; o += 65536 * 4) hipMemcpy((char*)Abig + o, h.data(), std::min((size_t)65536 * 4, abytes - o), hipMemcpyHostToDevice);
  for (size_t o = 0; o < wbytes; o += 65536 * 4) hipMemcpy((char*)Wp + o, h.data(), std::min((size_t)65536 * 4, wbytes - o), hipMemcpyHostToDevice);
  const int big = 36000;   // dynamic LDS that leaves room for one block per CU only
  for (int rep = 0; rep < 2; ++rep) {
    run<5, 0>(src, out, clk, iters, Abig, Wp, 0, big);
    run16<0>(src, out, clk, iters, big);
    run<5, 2>(src, out, clk, iters, Abig, Wp, 0, big);
    run16<2>(src, out, clk, iters, big);
    run<5, 2>(src, out, clk, iters, Abig, Wp, 0, 0);
    run16<2>(src, out, clk, iters, 0);
  }
  return 0;
}
