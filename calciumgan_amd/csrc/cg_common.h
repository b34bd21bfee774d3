// Shared device helpers for the gfx950 CalciumGAN kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "../../include/calciumgan_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define CG_LAUNCH_CHECK()                        \
  do {                                           \
    hipError_t e__ = hipGetLastError();          \
    return (int)e__;                             \
  } while (0)

// Kernel-precise timing of the MFMA launches (cg_profile_enable /
// cg_profile_collect in the C ABI): while enabled, a launch is issued through
// hipExtLaunchKernelGGL with an event pair that carries the kernel's own begin /
// end timestamps (what rocprofv3 --kernel-trace reports), instead of events
// recorded around the launch.  Defined in swconv.hip.
#define CG_FAMILY_SWCONV 0
#define CG_FAMILY_WGRAD 1
bool cg_prof_next(int family, hipEvent_t* start, hipEvent_t* stop);
#define CG_LAUNCH_PROF(family, kernel, grid, block, lds, stream, ...)          \
  do {                                                                         \
    hipEvent_t ps__, pe__;                                                     \
    if (cg_prof_next(family, &ps__, &pe__))                                    \
      hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, ps__, pe__, 0,   \
                            __VA_ARGS__);                                      \
    else                                                                       \
      hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);       \
  } while (0)

__device__ __forceinline__ float bf2f(uint16_t v) {
  return __uint_as_float(((uint32_t)v) << 16);
}

// round-to-nearest-even f32 -> bf16 (plain cast keeps NaN a NaN on gfx950)
__device__ __forceinline__ uint16_t f2bf(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(uint16_t, b);
}

__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
}

// PhaseShuffle source row (reference gan/models/calciumgan.py:117-138):
// out[t] = x[shuffle_src(t, shift, w)], tf.pad 'reflect' semantics.
__device__ __forceinline__ int shuffle_src(int t, int s, int w) {
  if (s > 0) {
    int u = t + s;
    return u < w ? u : 2 * (w - 1) - u;
  }
  int a = -s;
  return t < a ? a - t : t - a;
}

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
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}
