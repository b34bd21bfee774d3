// LDS read-rate probe (development tool): bytes per clock per CU of
// ds_read_b128 / ds_read_b64 / ds_read_b64_tr_b16 with 8 waves per workgroup,
// one workgroup per CU, conflict-free addresses.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe/lds_rate tools/probe/lds_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(2))) int i32x2;

template <int MODE>
__global__ __launch_bounds__(512) void probe(int iters, long long* cycles, int* sink) {
  extern __shared__ char smem[];
  const int lane = threadIdx.x & 63;
  int addr;
  if (MODE == 0) addr = lane * 16;                    // b128: lane-linear
  else if (MODE == 1) addr = lane * 8;                // b64: lane-linear
  else {                                              // tr: 8 rows x 32 B per half wave
    const int r16 = lane & 15, g4 = lane >> 4, q = r16 >> 2, p = r16 & 3;
    const int row = 4 * g4 + q;
    addr = row * 64 + (((row >> 2) & 1) * 32) + 8 * p;
  }
  addr += (threadIdx.x >> 6) * 2048;
  i32x4 a0 = {}, a1 = {}, a2 = {}, a3 = {};
  i32x2 b0 = {}, b1 = {}, b2 = {}, b3 = {};
  __syncthreads();
  const long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {
      asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:1024\n"
                   "ds_read_b128 %2, %4 offset:4096\n ds_read_b128 %3, %4 offset:5120\n"
                   "s_waitcnt lgkmcnt(0)"
                   : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"(addr));
    } else if (MODE == 1) {
      asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:1024\n"
                   "ds_read_b64 %2, %4 offset:4096\n ds_read_b64 %3, %4 offset:5120\n"
                   "s_waitcnt lgkmcnt(0)"
                   : "=v"(b0), "=v"(b1), "=v"(b2), "=v"(b3) : "v"(addr));
    } else {
      asm volatile("ds_read_b64_tr_b16 %0, %4\n ds_read_b64_tr_b16 %1, %4 offset:1024\n"
                   "ds_read_b64_tr_b16 %2, %4 offset:4096\n ds_read_b64_tr_b16 %3, %4 offset:5120\n"
                   "s_waitcnt lgkmcnt(0)"
                   : "=v"(b0), "=v"(b1), "=v"(b2), "=v"(b3) : "v"(addr));
    }
  }
  const long long t1 = clock64();
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
  if (a0[0] + a1[0] + a2[0] + a3[0] + b0[0] + b1[0] + b2[0] + b3[0] == 12345) sink[0] = 1;
}

int main() {
  long long* cyc; int* sink;
  hipMalloc(&cyc, 256 * 8); hipMalloc(&sink, 4);
  const int iters = 20000;
  const char* names[3] = {"ds_read_b128", "ds_read_b64", "ds_read_b64_tr_b16"};
  for (int mode = 0; mode < 3; ++mode) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) probe<0><<<256, 512, 65536>>>(iters, cyc, sink);
      if (mode == 1) probe<1><<<256, 512, 65536>>>(iters, cyc, sink);
      if (mode == 2) probe<2><<<256, 512, 65536>>>(iters, cyc, sink);
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double bytes = (double)iters * 4 * 512 * (mode == 0 ? 16 : 8);  // per CU
    printf("%-20s %8.3f ms  %6.1f B/ns/CU  (clock64 ticks %lld)\n", names[mode], ms,
           bytes / (ms * 1e6), h);
  }
  return 0;
}
