// Shared device helpers for the gfx950 CalciumGAN kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include <type_traits>
#include <utility>

#include "../../include/calciumgan_hip.h"

// Storage type of activations / activation gradients / packed MFMA operands:
// bf16 (default build, libcalciumgan_hip.so) or IEEE fp16 (-DCG_ACT_F16=1,
// libcalciumgan_hip_f16.so: the reference's mixed_float16 policy, main.py:22-30).
// Same sources, same C ABI; accumulation, statistics, losses and the master
// weights are f32 in both.
#ifndef CG_ACT_F16
#define CG_ACT_F16 0
#endif
#if CG_ACT_F16
typedef _Float16 act_scalar;
typedef __attribute__((ext_vector_type(8))) _Float16 act8;
#define cg_mfma_16x16x32 __builtin_amdgcn_mfma_f32_16x16x32_f16
#define cg_mfma_32x32x16 __builtin_amdgcn_mfma_f32_32x32x16_f16
#else
typedef __bf16 act_scalar;
typedef __attribute__((ext_vector_type(8))) __bf16 act8;
#define cg_mfma_16x16x32 __builtin_amdgcn_mfma_f32_16x16x32_bf16
#define cg_mfma_32x32x16 __builtin_amdgcn_mfma_f32_32x32x16_bf16
#endif
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

// Launch-path caches are per DEVICE (hipFuncSetAttribute(MaxDynamicShared...)
// applies to the current device only; a CU count or an occupancy answer belongs
// to one device) and may be touched from several host threads: one atomic flag /
// value per device id, checked and set without a lock (setting an attribute
// twice is harmless).  ADVICE r3.
#include <atomic>
constexpr int kCgMaxDevices = 32;
inline int cg_device_index() {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= kCgMaxDevices) d = 0;
  return d;
}
struct CgPerDeviceFlag {
  std::atomic<bool> set[kCgMaxDevices];
  bool test() const { return set[cg_device_index()].load(std::memory_order_acquire); }
  void mark() { set[cg_device_index()].store(true, std::memory_order_release); }
};

__device__ __forceinline__ float act2f(uint16_t v) {
#if CG_ACT_F16
  return (float)__builtin_bit_cast(_Float16, v);
#else
  return __uint_as_float(((uint32_t)v) << 16);
#endif
}

// round-to-nearest-even f32 -> bf16 / fp16 (a plain cast keeps NaN a NaN on
// gfx950; fp16 overflows to infinity, which the loss scaler looks for)
__device__ __forceinline__ uint16_t f2act(float f) {
  act_scalar b = (act_scalar)f;
  return __builtin_bit_cast(uint16_t, b);
}

// the two activations of one 32-bit word
__device__ __forceinline__ float act_lo(uint32_t w) { return act2f((uint16_t)w); }
__device__ __forceinline__ float act_hi(uint32_t w) {
  return act2f((uint16_t)(w >> 16));
}

__device__ __forceinline__ uint32_t pack2act(float lo, float hi) {
  return (uint32_t)f2act(lo) | ((uint32_t)f2act(hi) << 16);
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

// Cross-lane sums without the LDS crossbar: __shfl_xor compiles to
// ds_bpermute_b32 (an LDS-pipe round trip of ~100 cycles per step, six dependent
// steps per wave sum -- the LayerNorm kernels spent most of their time there);
// DPP moves inside a 16-lane row cost one VALU op each and gfx950's
// v_permlane16_swap / v_permlane32_swap pair the rows.  After step k every lane
// holds the sum of its 2^k-lane group, so any pairing of the groups will do
// (quad swaps, half-row / row mirrors, row swaps).
template <int CTRL>
__device__ __forceinline__ float dpp_partner(float v) {
  return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(
      0, (int)__float_as_uint(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_sum_partner(float v) {
  const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(v),
                                                   __float_as_uint(v), false, false);
  return __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
}
__device__ __forceinline__ float half32_sum_partner(float v) {
  const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v),
                                                   __float_as_uint(v), false, false);
  return __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
}
// sum over aligned groups of `n` lanes (n a power of two, 1 .. 64); every lane
// of a group receives the group's sum
__device__ __forceinline__ float group_sum_n(float v, int n) {
  if (n > 1) v += dpp_partner<0xB1>(v);    // quad_perm [1,0,3,2]
  if (n > 2) v += dpp_partner<0x4E>(v);    // quad_perm [2,3,0,1]
  if (n > 4) v += dpp_partner<0x141>(v);   // row_half_mirror
  if (n > 8) v += dpp_partner<0x140>(v);   // row_mirror
  if (n > 16) v = row16_sum_partner(v);    // rows 0+1, 2+3
  if (n > 32) v = half32_sum_partner(v);   // halves
  return v;
}
__device__ __forceinline__ float wave_sum(float v) { return group_sum_n(v, 64); }
// min / max over aligned groups of n lanes, the same DPP / permlane pairings
// (__shfl_xor is ds_bpermute_b32: an LDS-pipe round trip per step)
__device__ __forceinline__ float group_max_n(float v, int n) {
  if (n > 1) v = fmaxf(v, dpp_partner<0xB1>(v));
  if (n > 2) v = fmaxf(v, dpp_partner<0x4E>(v));
  if (n > 4) v = fmaxf(v, dpp_partner<0x141>(v));
  if (n > 8) v = fmaxf(v, dpp_partner<0x140>(v));
  if (n > 16) {
    const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v),
                                                     false, false);
    v = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
  }
  if (n > 32) {
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v),
                                                     false, false);
    v = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
  }
  return v;
}
__device__ __forceinline__ float group_min_n(float v, int n) { return -group_max_n(-v, n); }
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

// compile-time loop: f(integral_constant<int, 0>), ..., f(integral_constant<int, N - 1>)
template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>,
                                                F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

// The K loop's LDS reads and MFMAs are inline assembly: hipcc, given the
// builtins and the unrolled pass, renames every accumulator per K-step and
// copies it back (562 v_mov, 90-170 spilled VGPRs); `+v` ties the MFMA's
// destination to its accumulator, the reads land in fixed fragment sets, and
// volatile asm keeps the stream in program order.  The compiler does not see
// these reads, so the waits are placed by hand (s_waitcnt lgkmcnt(0) after the
// MFMA block that hid their latency).
template <int OFF>
__device__ __forceinline__ void lds_read128(act8& d, int addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field");
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
__device__ __forceinline__ void mfma_acc(f32x4& c, const act8& a, const act8& b) {
#if CG_ACT_F16
  asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
#else
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
#endif
}
__device__ __forceinline__ void lds_wait() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

