// Probe: does a hipMemsetAsync NODE of a captured hipGraph still write zeros
// when other work (eager memsets with other patterns, kernels) runs between
// two replays?  (the round-2 "stale graph" penalties.)
//   hipcc --offload-arch=gfx950 -O2 tools/probe/graph_memset.hip -o tools/probe/graph_memset
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { \
  printf("HIP error %d at %s:%d\n", (int)e, __FILE__, __LINE__); return 2; } } while (0)

__global__ void add_one(float* v, int n) {
  if (threadIdx.x < n) atomicAdd(v + threadIdx.x, 1.0f);
}
__global__ void copy_out(const float* v, float* out, int n) {
  if (threadIdx.x < n) out[threadIdx.x] = v[threadIdx.x];
}
__global__ void busy(float* p, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = p[i] * 1.0001f + 1.f;
}

int main(int argc, char** argv) {
  const int n = 8, blocks = 64;
  float *buf, *out, *other, *scratch;
  const size_t big = 64u << 20;
  CK(hipMalloc(&buf, 512)); CK(hipMalloc(&out, 512)); CK(hipMalloc(&other, 512));
  CK(hipMalloc(&scratch, big));
  hipStream_t s; CK(hipStreamCreate(&s));
  CK(hipMemset(buf, 0x7f, 512));
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  CK(hipMemsetAsync(buf, 0, sizeof(float) * n, s));
  hipLaunchKernelGGL(add_one, dim3(blocks), dim3(64), 0, s, buf, n);
  hipLaunchKernelGGL(copy_out, dim3(1), dim3(64), 0, s, buf, out, n);
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  float h[8];
  int bad_plain = 0, bad_mixed = 0;
  for (int it = 0; it < 50; ++it) {
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i) if (h[i] != (float)blocks) { ++bad_plain; if (bad_plain < 5) printf("plain it %d: out[%d] = %g\n", it, i, h[i]); break; }
  }
  for (int it = 0; it < 50; ++it) {
    // eager work between replays: memsets with other patterns, kernels
    CK(hipMemsetAsync(other, 0x7f, 512, s));
    CK(hipMemsetAsync(scratch, 0x55, big, s));
    hipLaunchKernelGGL(busy, dim3((unsigned)(big / 4 / 256)), dim3(256), 0, s, scratch, big / 4);
    CK(hipMemsetAsync(other, 0x3c, 32, s));
    // many small fills / copies: whatever staging the runtime keeps for a fill
    // pattern is recycled
    for (int j = 0; j < 400; ++j) {
      CK(hipMemsetAsync(other + 8 * (j % 8), 0x40 + (j & 31), 32, s));
      CK(hipMemcpyAsync(other + 64, other, 32, hipMemcpyDeviceToDevice, s));
    }
    if (it & 1) CK(hipStreamSynchronize(s));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i) if (h[i] != (float)blocks) { ++bad_mixed; if (bad_mixed < 5) printf("mixed it %d: out[%d] = %g\n", it, i, h[i]); break; }
  }
  printf("GRAPH_MEMSET plain_bad=%d/50 mixed_bad=%d/50\n", bad_plain, bad_mixed);
  return 0;
}
