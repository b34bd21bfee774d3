// Ceiling probe for the swconv inner loop: operands resident in LDS, no global
// loads, optional barrier every `bar_every` K-steps.  Prints TFLOP/s for a
// given wave tile (MT x 4 subtiles of 16x16), workgroups per CU (via LDS size).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_probe mfma_probe.hip && ./mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int MT, int NT>
__global__ __launch_bounds__(256) void probe(float* out, int ksteps, int bar_every,
                                             int pitchA, int pitchB) {
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, g = lane >> 4;
  for (int i = tid; i < 20000; i += 256) {
    unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u;  // random bf16 in (-2, 2)
    lds[i] = (unsigned short)(((h >> 9) & 0x807f) | 0x3f00 | ((h >> 3) & 0x0080));
  }
  __syncthreads();
  unsigned short* ldsA = lds;
  unsigned short* ldsB = lds + 12000;
  int rowbase[MT];
  for (int mt = 0; mt < MT; ++mt) rowbase[mt] = ((wave * MT + mt) * 16 % 64 + r16) * pitchA + g * 8;
  f32x4 acc[MT][NT];
  for (int mt = 0; mt < MT; ++mt) for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0, 0, 0, 0};
  int tap = 0;
  for (int ks = 0; ks < ksteps; ++ks) {
    const int aoff = tap * pitchA;
    bf16x8 b[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      b[nt] = *reinterpret_cast<const bf16x8*>(ldsB + ((nt * 16 + r16) % 64) * pitchB + ((ks & 3) * 4 + g) * 8);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(ldsA + rowbase[mt] + aoff);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b[nt], acc[mt][nt], 0, 0, 0);
    }
    tap = (tap + 1) % 12;
    if (bar_every > 0 && (ks % bar_every) == bar_every - 1) __syncthreads();
  }
  float s = 0;
  for (int mt = 0; mt < MT; ++mt) for (int nt = 0; nt < NT; ++nt) s += acc[mt][nt][0] + acc[mt][nt][3];
  if (s == 12345.f) out[0] = s;
}

template <int MT, int NT>
void run(int lds_bytes, int bar_every, const char* tag) {
  float* out; hipMalloc(&out, 4);
  hipFuncSetAttribute((const void*)&probe<MT, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int ksteps = 2048, blocks = 256 * 12;
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  probe<MT, NT><<<blocks, 256, lds_bytes>>>(out, ksteps, bar_every, 80, 144);
  hipEventRecord(s);
  probe<MT, NT><<<blocks, 256, lds_bytes>>>(out, ksteps, bar_every, 80, 144);
  hipEventRecord(e); hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e);
  double fl = (double)blocks * 4 * ksteps * MT * NT * 16384.0;
  printf("%-28s MT=%d NT=%d lds=%6d bar_every=%2d : %7.1f TF/s\n", tag, MT, NT, lds_bytes, bar_every, fl / (ms * 1e-3) / 1e12);
  hipFree(out);
}

int main() {
  for (int bar : {0, 4, 2}) {
    run<4, 4>(160 * 1024, bar, "1 wg/cu");
    run<4, 4>(80 * 1024, bar, "2 wg/cu");
    run<4, 4>(53 * 1024, bar, "3 wg/cu");
    run<2, 4>(53 * 1024, bar, "3 wg/cu");
    run<2, 4>(40 * 1024, bar, "4 wg/cu");
    run<4, 8>(80 * 1024, bar, "2 wg/cu 64x128");
    run<8, 4>(80 * 1024, bar, "2 wg/cu 128x64");
  }
  return 0;
}
