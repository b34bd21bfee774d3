// Per-timestep Dense with the whole weight matrix held in MFMA registers.
//
// The generator ends with layers.Dense(C) on the last axis + sigmoid
// (gan/models/calciumgan.py:96-101): rows = B*L = 262 144 at cfg2, K = 128,
// N = 102 -- 8.6 GFLOP against 201 MB of traffic (bf16 in, f32 out), i.e.
// HBM-bound by a wide margin.  The general cg_swconv path stages the source
// through LDS in four chunks with barriers and reaches ~2.9 TB/s on it; this
// kernel is the streaming form: a wave keeps all of W (<= 128 x 128 bf16 = 128
// VGPRs of B fragments) resident, loads its A fragments straight from global
// memory in MFMA layout (16 rows x 64 bytes per instruction, the next block
// prefetched), and turns the accumulators around through a wave-private LDS
// tile so that every store instruction writes two whole 512-byte rows.
#include "cg_common.h"

namespace {

constexpr int kDrThreads = 256;
constexpr int kDrPitch = 132;  // f32 pitch of the 16-row transpose tile

struct DenseRowsArgs {
  const uint16_t* x;    // bf16 [rows][Cx]
  const uint16_t* w;    // packed operand (taps 1, CK 32): [128][Cx/32][16][8]
  const float* bias;    // f32 [N] or null
  float* y;             // f32 [rows][Cy]
  long long rows;
  int Cx, N, Cy, epilogue;
};

// (W takes 32 * KSTEPS registers per lane; the second launch-bound argument
// holds the rest so that two workgroups share a CU.)
template <int KSTEPS>
__global__ __launch_bounds__(kDrThreads, 2) void dense_rows_kernel(
    DenseRowsArgs a) {
  __shared__ float tile[4][16 * kDrPitch];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int r16 = lane & 15;
  const int g = lane >> 4;
  // all of W as B fragments: fragment (ks, nt) = 8 k-values of column
  // nt*16 + r16, k-group g of K-step ks
  act8 wf[KSTEPS][8];
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
    for (int nt = 0; nt < 8; ++nt)
      wf[ks][nt] = *reinterpret_cast<const act8*>(
          a.w + (((long long)(nt * 16 + r16) * KSTEPS + ks) * 16 + g) * 8);
  float* tl = tile[wave];
  // read-back map of the transpose tile: lane -> (row lane>>5 of a 2-row
  // pass, 4 consecutive columns (lane&31)*4)
  const int tcol = (lane & 31) * 4;
  const int trow = lane >> 5;
  f32x4 bv = {0.f, 0.f, 0.f, 0.f};   // bias of this lane's 4 columns
  f32x4 keep = {0.f, 0.f, 0.f, 0.f}; // 1 for real channels, 0 for padding
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (tcol + e < a.N) {
      keep[e] = 1.f;
      if (a.bias) bv[e] = a.bias[tcol + e];
    }
  const bool sig = a.epilogue == CG_EPI_SIGMOID;

  const long long nblk = (a.rows + 15) / 16;
  const long long stride = (long long)gridDim.x * 4;
  long long blk = (long long)blockIdx.x * 4 + wave;
  auto load_a = [&](long long b, act8 (&dst)[KSTEPS]) {
    const long long row = b * 16 + r16;
    const bool ok = b < nblk && row < a.rows;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      act8 v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (ok)
        v = *reinterpret_cast<const act8*>(a.x + row * a.Cx + ks * 32 + g * 8);
      dst[ks] = v;
    }
  };
  auto compute = [&](long long b, const act8 (&af)[KSTEPS]) {
    f32x4 acc[8];
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
      for (int nt = 0; nt < 8; ++nt)
        acc[nt] = cg_mfma_16x16x32(af[ks], wf[ks][nt],
                                                          acc[nt], 0, 0, 0);
    // accumulator (row 4g + r, column nt*16 + r16) -> LDS tile -> row-major
#pragma unroll
    for (int nt = 0; nt < 8; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        tl[(4 * g + r) * kDrPitch + nt * 16 + r16] = acc[nt][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll 2
    for (int pass = 0; pass < 8; ++pass) {
      const int row = pass * 2 + trow;
      const long long m = b * 16 + row;
      f32x4 v = *reinterpret_cast<const f32x4*>(tl + row * kDrPitch + tcol);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float t = v[e] + bv[e];
        const float sg = 1.f / (1.f + __expf(-t));
        v[e] = (sig ? sg : t) * keep[e];  // channel padding stays zero
      }
      if (m < a.rows && tcol < a.Cy)
        *reinterpret_cast<f32x4*>(a.y + m * a.Cy + tcol) = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  // two statically named fragment sets (a runtime-indexed register array
  // would live in scratch): the next block's loads fly during this block
  act8 af0[KSTEPS], af1[KSTEPS];
  load_a(blk, af0);
  while (blk < nblk) {
    load_a(blk + stride, af1);
    compute(blk, af0);
    blk += stride;
    if (blk >= nblk) break;
    load_a(blk + stride, af0);
    compute(blk, af1);
    blk += stride;
  }
}

}  // namespace

extern "C" int cg_dense_rows(const void* x, const void* w, const float* bias,
                             float* y, long long rows, int Cx, int N, int Cy,
                             int epilogue, void* stream) {
  if (!x || !w || !y || rows < 1) return CG_EINVAL;
  if (Cx % 32 || Cx < 32 || Cx > 128 || N < 1 || N > 128 || Cy < N || Cy > 128 ||
      Cy % 4)
    return CG_EINVAL;
  if (epilogue != CG_EPI_NONE && epilogue != CG_EPI_SIGMOID) return CG_EINVAL;
  DenseRowsArgs a;
  a.x = reinterpret_cast<const uint16_t*>(x);
  a.w = reinterpret_cast<const uint16_t*>(w);
  a.bias = bias;
  a.y = y;
  a.rows = rows;
  a.Cx = Cx; a.N = N; a.Cy = Cy; a.epilogue = epilogue;
  const long long nblk = (rows + 15) / 16;
  long long blocks = (nblk + 3) / 4;
  if (blocks > 512) blocks = 512;  // two 4-wave workgroups per CU, grid-stride
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)blocks), block(kDrThreads);
  switch (Cx / 32) {
    case 1: hipLaunchKernelGGL(dense_rows_kernel<1>, grid, block, 0, s, a); break;
    case 2: hipLaunchKernelGGL(dense_rows_kernel<2>, grid, block, 0, s, a); break;
    case 3: hipLaunchKernelGGL(dense_rows_kernel<3>, grid, block, 0, s, a); break;
    default: hipLaunchKernelGGL(dense_rows_kernel<4>, grid, block, 0, s, a); break;
  }
  CG_LAUNCH_CHECK();
}
