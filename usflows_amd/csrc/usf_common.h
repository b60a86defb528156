// Shared device/host helpers for libusflows_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/usflows_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define USF_WAVE 64

namespace usf {

void set_error(const char* fmt, ...);
// a named tuning knob (usf_api.hip: the one table, preset from USFLOWS_AMD_TUNE, changed with usf_set_tuning)
long long tuning(const char* name, long long dflt);
// measurement aid (usf_set_clock_buffer): device buffer [2] that the planes GEMM and the MFMA probe add their blocks' lifetimes
// to -- [0] shader-clock cycles (s_memtime), [1] ticks of the constant 100 MHz counter (s_memrealtime); nullptr: off
unsigned long long* clock_buffer();

static inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Per-device launch state (a process may drive several GPUs: function attributes and the CU count belong to a device)
#define USF_MAX_DEVICES 64
static inline int current_device_slot() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= USF_MAX_DEVICES) dev = 0;
  return dev;
}
// compute units of the current device (cached per device; 256 when the query fails)
static inline int device_cu_count() {
  static int cus[USF_MAX_DEVICES] = {0};
  const int dev = current_device_slot();
  if (cus[dev] <= 0) {
    hipDeviceProp_t prop;
    cus[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }
  return cus[dev];
}

// leaky_relu exactly as ATen: x > 0 ? x : x * slope
__device__ __forceinline__ float act_apply(float v, int act, float slope) {
  return (act == USF_ACT_LEAKY_RELU) ? (v > 0.0f ? v : v * slope) : v;
}
// USF_ACT_GATE: leaky_relu_backward from the saved OUTPUT h (as usf_act_grad_f32: h > 0 ? v : v * slope)
__device__ __forceinline__ float gate_apply(float v, float h, float slope) { return (h > 0.0f) ? v : v * slope; }

// 64-lane sum via DPP-friendly shuffles (wavefront = 64 on gfx950)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

}  // namespace usf
