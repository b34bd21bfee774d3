// Ceiling probe for the swconv inner loop: operands resident in LDS, no global
// loads, optional barrier every `bar_every` K-steps.  Prints TFLOP/s for a
// given wave tile (MT x 4 subtiles of 16x16), workgroups per CU (via LDS size).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_probe mfma_probe.hip && ./mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

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


// same loop on v_mfma_f32_32x32x16_bf16: MT x NT subtiles of 32x32, two
// 16-deep MFMAs per 32-deep K-step
template <int MT, int NT>
__global__ __launch_bounds__(256) void probe32(float* out, int ksteps, int bar_every,
                                               int pitchA, int pitchB) {
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r32 = lane & 31, g = lane >> 5;
  for (int i = tid; i < 20000; i += 256) {
    unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u;
    lds[i] = (unsigned short)(((h >> 9) & 0x807f) | 0x3f00 | ((h >> 3) & 0x0080));
  }
  __syncthreads();
  unsigned short* ldsA = lds;
  unsigned short* ldsB = lds + 12000;
  int rowbase[MT];
  for (int mt = 0; mt < MT; ++mt) rowbase[mt] = ((wave * MT + mt) * 32 % 64 + r32) * pitchA + g * 8;
  f32x16 acc[MT][NT];
  for (int mt = 0; mt < MT; ++mt) for (int nt = 0; nt < NT; ++nt)
    for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;
  int tap = 0;
  for (int ks = 0; ks < ksteps; ++ks) {
    const int aoff = tap * pitchA;
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
      bf16x8 b[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        b[nt] = *reinterpret_cast<const bf16x8*>(ldsB + ((nt * 32 + r32) % 64) * pitchB + ((ks & 3) * 4 + kh * 2 + g) * 8);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(ldsA + rowbase[mt] + aoff + kh * 16);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[nt], acc[mt][nt], 0, 0, 0);
      }
    }
    tap = (tap + 1) % 12;
    if (bar_every > 0 && (ks % bar_every) == bar_every - 1) __syncthreads();
  }
  float s = 0;
  for (int mt = 0; mt < MT; ++mt) for (int nt = 0; nt < NT; ++nt) s += acc[mt][nt][0] + acc[mt][nt][15];
  if (s == 12345.f) out[0] = s;
}

// software-pipelined stage loop: the fragments of K-step j+1 are read before
// the MFMAs of K-step j are issued (register double buffering inside a stage
// of KS K-steps; the first read of a stage follows its barrier)
template <int MF, int MT, int NT, int KS>
__global__ __launch_bounds__(256) void probe_pipe(float* out, int nstages, int pitchA, int pitchB) {
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  using acc_t = typename std::conditional<MF == 16, f32x4, f32x16>::type;
  constexpr int KH = MF / 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rM = lane & (MF - 1), g = lane / MF;
  for (int i = tid; i < 20000; i += 256) {
    unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u;
    lds[i] = (unsigned short)(((h >> 9) & 0x807f) | 0x3f00 | ((h >> 3) & 0x0080));
  }
  __syncthreads();
  unsigned short* ldsA = lds;
  unsigned short* ldsB = lds + 12000;
  int rowbase[MT];
  for (int mt = 0; mt < MT; ++mt) rowbase[mt] = ((wave * MT + mt) * MF % 64 + rM) * pitchA + g * 8;
  acc_t acc[MT][NT];
  for (int mt = 0; mt < MT; ++mt) for (int nt = 0; nt < NT; ++nt)
    for (int i = 0; i < MF * MF / 64; ++i) acc[mt][nt][i] = 0.f;
  int tap = 0;
  for (int st = 0; st < nstages; ++st) {
    __syncthreads();
    bf16x8 af[2][KH][MT], bf[2][KH][NT];
    auto rd = [&](int buf, int j) {
      const int aoff = ((tap + j) % 12) * pitchA;
#pragma unroll
      for (int kh = 0; kh < KH; ++kh) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          bf[buf][kh][nt] = *reinterpret_cast<const bf16x8*>(ldsB + ((nt * MF + rM) % 64) * pitchB + (j * 4 + kh * 2 + g) * 8);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          af[buf][kh][mt] = *reinterpret_cast<const bf16x8*>(ldsA + rowbase[mt] + aoff + kh * 16);
      }
    };
    rd(0, 0);
#pragma unroll
    for (int j = 0; j < KS; ++j) {
      if (j + 1 < KS) rd((j + 1) & 1, j + 1);
#pragma unroll
      for (int kh = 0; kh < KH; ++kh)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            if constexpr (MF == 16)
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[j & 1][kh][mt], bf[j & 1][kh][nt], acc[mt][nt], 0, 0, 0);
            else
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[j & 1][kh][mt], bf[j & 1][kh][nt], acc[mt][nt], 0, 0, 0);
          }
    }
    tap = (tap + KS) % 12;
  }
  float s = 0;
  for (int mt = 0; mt < MT; ++mt) for (int nt = 0; nt < NT; ++nt) s += acc[mt][nt][0] + acc[mt][nt][3];
  if (s == 12345.f) out[0] = s;
}

template <int MF, int MT, int NT, int KS>
void run_pipe(int lds_bytes, const char* tag) {
  float* out; hipMalloc(&out, 4);
  hipFuncSetAttribute((const void*)&probe_pipe<MF, MT, NT, KS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int nstages = 2048 / KS, blocks = 256 * 12;
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  probe_pipe<MF, MT, NT, KS><<<blocks, 256, lds_bytes>>>(out, nstages, 80, 144);
  hipEventRecord(s);
  probe_pipe<MF, MT, NT, KS><<<blocks, 256, lds_bytes>>>(out, nstages, 80, 144);
  hipEventRecord(e); hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e);
  double fl = (double)blocks * 4 * 2048 * MT * NT * (MF * MF * 32 * 2.0);
  printf("PIPE MF=%d %-18s MT=%d NT=%d KS=%d lds=%6d : %7.1f TF/s\n", MF, tag, MT, NT, KS, lds_bytes, fl / (ms * 1e-3) / 1e12);
  hipFree(out);
}

template <int MT, int NT>
void run32(int lds_bytes, int bar_every, const char* tag) {
  float* out; hipMalloc(&out, 4);
  hipFuncSetAttribute((const void*)&probe32<MT, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int ksteps = 2048, blocks = 256 * 12;
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  probe32<MT, NT><<<blocks, 256, lds_bytes>>>(out, ksteps, bar_every, 80, 144);
  hipEventRecord(s);
  probe32<MT, NT><<<blocks, 256, lds_bytes>>>(out, ksteps, bar_every, 80, 144);
  hipEventRecord(e); hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e);
  double fl = (double)blocks * 4 * ksteps * MT * NT * 2 * 32768.0;
  printf("32x32x16 %-19s MT=%d NT=%d lds=%6d bar_every=%2d : %7.1f TF/s\n", tag, MT, NT, lds_bytes, bar_every, fl / (ms * 1e-3) / 1e12);
  hipFree(out);
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
  run_pipe<16, 4, 4, 4>(80 * 1024, "2 wg/cu 64x64");
  run_pipe<16, 4, 4, 4>(53 * 1024, "3 wg/cu 64x64");
  run_pipe<16, 4, 4, 2>(53 * 1024, "3 wg/cu 64x64");
  run_pipe<16, 2, 4, 4>(40 * 1024, "4 wg/cu 32x64");
  run_pipe<32, 2, 2, 4>(80 * 1024, "2 wg/cu 64x64");
  run_pipe<32, 2, 2, 4>(53 * 1024, "3 wg/cu 64x64");
  run_pipe<32, 2, 2, 2>(53 * 1024, "3 wg/cu 64x64");
  run_pipe<32, 1, 2, 4>(40 * 1024, "4 wg/cu 32x64");
  run_pipe<32, 4, 2, 4>(80 * 1024, "2 wg/cu 128x64");
  run_pipe<32, 4, 2, 2>(80 * 1024, "2 wg/cu 128x64");
  for (int bar : {0, 4, 2}) {
    run<4, 4>(160 * 1024, bar, "1 wg/cu");
    run<4, 4>(80 * 1024, bar, "2 wg/cu");
    run<4, 4>(53 * 1024, bar, "3 wg/cu");
    run<2, 4>(53 * 1024, bar, "3 wg/cu");
    run<2, 4>(40 * 1024, bar, "4 wg/cu");
    run<4, 8>(80 * 1024, bar, "2 wg/cu 64x128");
    run<8, 4>(80 * 1024, bar, "2 wg/cu 128x64");
    run32<2, 2>(80 * 1024, bar, "2 wg/cu 64x64");
    run32<2, 2>(53 * 1024, bar, "3 wg/cu 64x64");
    run32<1, 2>(40 * 1024, bar, "4 wg/cu 32x64");
    run32<4, 2>(80 * 1024, bar, "2 wg/cu 128x64");
    run32<2, 4>(80 * 1024, bar, "2 wg/cu 64x128");
  }
  return 0;
}
