// What the chip SUSTAINS on dense bf16 MFMA (round 4): a register-
// resident v_mfma_f32_16x16x32_bf16 loop (no LDS, no global traffic; 4 waves per
// SIMD, 8 independent accumulators per wave) launched back to back for a few
// seconds on random or zero operands; prints TFLOP/s per 0.25 s window and the
// in-kernel clock (s_memtime / s_memrealtime) of each window's last launch.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_sustained mfma_sustained.hip
//   ./mfma_sustained [seconds] [zero|rand] [wide]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ __launch_bounds__(256, 4) void mfma_loop(float* out, unsigned long long* clk,
                                                    int iters, int zero) {
  const unsigned tid = blockIdx.x * 256 + threadIdx.x;
  union { bf16x8 v; unsigned short u[8]; } a[2], b[4];
  for (int i = 0; i < 2; ++i)
    for (int e = 0; e < 8; ++e) {
      unsigned h = (tid * 8 + e + i * 977) * 2654435761u;
      a[i].u[e] = zero ? 0 : (unsigned short)(((h >> 9) & 0x807f) | 0x3f00 | ((h >> 3) & 0x0080));
    }
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 8; ++e) {
      unsigned h = (tid * 8 + e + i * 131 + 7) * 2246822519u;
      b[i].u[e] = zero ? 0 : (unsigned short)(((h >> 9) & 0x807f) | 0x3f00 | ((h >> 3) & 0x0080));
    }
  f32x4 acc[2][4];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i].v, b[j].v, acc[i][j], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
  if (s == 12345.f) out[0] = s;
  if (tid == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

typedef __attribute__((ext_vector_type(16))) float f32x16;
// the same loop on v_mfma_f32_32x32x16_bf16: 2 A x 2 B fragments, four 32 x 32
// accumulators (64 registers; the 16 x 16 form above holds 32): per K = 16 the same
// operand bytes feed twice the multiply-adds
__global__ __launch_bounds__(256, 4) void mfma_loop32(float* out, unsigned long long* clk,
                                                      int iters, int zero) {
  const unsigned tid = blockIdx.x * 256 + threadIdx.x;
  union { bf16x8 v; unsigned short u[8]; } a[2], b[2];
  for (int i = 0; i < 2; ++i)
    for (int e = 0; e < 8; ++e) {
      unsigned h = (tid * 8 + e + i * 977) * 2654435761u;
      a[i].u[e] = zero ? 0 : (unsigned short)(((h >> 9) & 0x807f) | 0x3f00 | ((h >> 3) & 0x0080));
      unsigned g = (tid * 8 + e + i * 131 + 7) * 2246822519u;
      b[i].u[e] = zero ? 0 : (unsigned short)(((g >> 9) & 0x807f) | 0x3f00 | ((g >> 3) & 0x0080));
    }
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j)
    for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i].v, b[j].v, acc[i][j], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) s += acc[i][j][0] + acc[i][j][15];
  if (s == 12345.f) out[0] = s;
  if (tid == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 3.0;
  const int zero = argc > 2 && argv[2][0] == 'z' ? 1 : 0;
  const int wide = argc > 3 ? 1 : 0;  // third argument: the 32 x 32 x 16 form
  float* out; unsigned long long* clk;
  hipMalloc(&out, 4); hipMalloc(&clk, 16);
  const int blocks = 256 * 4, iters = 20000;        // ~3 ms per launch
  const double flop = wide ? (double)blocks * 4 * iters * 4 * 2.0 * 32 * 32 * 16
                           : (double)blocks * 4 * iters * 8 * 2.0 * 16 * 16 * 32;
  auto kern = wide ? mfma_loop32 : mfma_loop;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, clk, iters, zero);
  hipDeviceSynchronize();
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double start = now();
  printf("dense bf16 MFMA %s from registers, %s operands, %d blocks x 256 threads:\n", wide ? "32x32x16" : "16x16x32", zero ? "ZERO" : "random", blocks);
  while (now() - start < seconds) {
    const double w0 = now();
    int n = 0;
    while (now() - w0 < 0.25) {
      for (int k = 0; k < 8; ++k)
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, clk, iters, zero);
      hipDeviceSynchronize();
      n += 8;
    }
    const double dt = now() - w0;
    unsigned long long c[2];
    hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
    printf("  t=%5.2f s  %7.1f TFLOP/s   in-kernel clock %.3f GHz\n", now() - start,
           flop * n / dt / 1e12, (double)c[0] / (double)c[1] * 0.1);
  }
  return 0;
}
