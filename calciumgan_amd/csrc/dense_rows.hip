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


// ---------------------------------------------------------------------------
// The same Dense + sigmoid for the fake batches of ALL critic updates of a step,
// with the interpolation and the packing of the critic's input fused into its
// epilogue (round 5).  On one rank the generator runs once per step over n x B
// samples (WGAN_GP._critic_generate_all); the f32 fake batch it used to write
// (545 MB at cfg2) was read back by n cg_interp_pack launches that formed
//     X0_k = [ real | fake_k | x^_k ],  x^ = a real + (1 - a) fake   (wgan_gp.py:38-41)
// in bf16.  Here a wave takes a 16-row block of the (sample, time) grid, loads its
// rows of `real` ONCE, and for k = 0 .. n - 1 computes update k's 16 output rows
// and stores the three bf16 segments of X0_k directly: no f32 fake batch in HBM,
// `real` read once instead of n times.  Same arithmetic, same order: the bytes of
// X0_k equal those of cg_dense_rows + cg_interp_pack (tests/test_hip_kernels.py).
// ---------------------------------------------------------------------------
constexpr int kMaxInterp = 8;
struct DenseInterpArgs {
  const uint16_t* x;    // bf16 [n * B * L][Cx]: the generator's last hidden layer
  const uint16_t* w;
  const float* bias;
  const float* real;    // f32 [B * L][Cr]
  const float* alpha;   // f32 [n * B]
  uint16_t* x0[kMaxInterp];  // per update: bf16 [3 B][L][Cp]
  int n, B, L, Cx, N, Cr, Cp, epilogue;
};

template <int KSTEPS>
__global__ __launch_bounds__(kDrThreads, 2) void dense_rows_interp_kernel(
    DenseInterpArgs a) {
  __shared__ float tile[4][16 * kDrPitch];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int r16 = lane & 15;
  const int g = lane >> 4;
  act8 wf[KSTEPS][8];
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
    for (int nt = 0; nt < 8; ++nt)
      wf[ks][nt] = *reinterpret_cast<const act8*>(
          a.w + (((long long)(nt * 16 + r16) * KSTEPS + ks) * 16 + g) * 8);
  float* tl = tile[wave];
  const int tcol = (lane & 31) * 4;
  const int trow = lane >> 5;
  f32x4 bv = {0.f, 0.f, 0.f, 0.f};
  f32x4 keep = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (tcol + e < a.N) {
      keep[e] = 1.f;
      if (a.bias) bv[e] = a.bias[tcol + e];
    }
  const bool sig = a.epilogue == CG_EPI_SIGMOID;
  const long long R = (long long)a.B * a.L;  // rows of one update (L % 16 == 0)
  const long long nblk = R / 16;
  const long long nitem = nblk * a.n;        // (block, update) pairs, update fastest
  const long long stride = (long long)gridDim.x * 4;
  const long long seg = R * a.Cp;            // elements of one segment of X0
  auto load_a = [&](long long blk, int k, bool ok, act8 (&dst)[KSTEPS]) {
    const long long row = (long long)k * R + blk * 16 + r16;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      act8 v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (ok) v = *reinterpret_cast<const act8*>(a.x + row * a.Cx + ks * 32 + g * 8);
      dst[ks] = v;
    }
  };
  auto store4 = [&](uint16_t* p, const f32x4& v) {
    *reinterpret_cast<uint2*>(p) = make_uint2(pack2act(v[0], v[1]), pack2act(v[2], v[3]));
  };
  auto compute = [&](long long blk, int k, const act8 (&af)[KSTEPS]) {
    f32x4 acc[8];
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
      for (int nt = 0; nt < 8; ++nt)
        acc[nt] = cg_mfma_16x16x32(af[ks], wf[ks][nt], acc[nt], 0, 0, 0);
#pragma unroll
    for (int nt = 0; nt < 8; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        tl[(4 * g + r) * kDrPitch + nt * 16 + r16] = acc[nt][r];
    // this lane's share of the block's `real` rows, into the registers the
    // accumulators just left (HBM once per block: updates 1 .. n - 1 of the block
    // follow at once and find the lines in L2; holding them across the MFMAs of
    // all n updates spilled 12 registers)
    f32x4 rr[8];
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
      const long long m = blk * 16 + pass * 2 + trow;
      const float* p = a.real + m * a.Cr + tcol;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (tcol + e < a.N) v[e] = p[e];
      rr[pass] = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int b = (int)((blk * 16) / a.L);  // the block's sample
    // (alpha == nullptr: the caller forms layer 1 on x^ from the layer's outputs on
    // real and fake, cg_lrelu_mix -- x^ itself is never read, so never written)
    const float al = a.alpha ? a.alpha[k * a.B + b] : 0.f;
    uint16_t* x0 = a.x0[k];
#pragma unroll 2
    for (int pass = 0; pass < 8; ++pass) {
      const int row = pass * 2 + trow;
      const long long m = blk * 16 + row;
      f32x4 f = *reinterpret_cast<const f32x4*>(tl + row * kDrPitch + tcol);
      f32x4 xh;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float t = f[e] + bv[e];
        const float sg = 1.f / (1.f + __expf(-t));
        f[e] = (sig ? sg : t) * keep[e];
        xh[e] = al * rr[pass][e] + (1.f - al) * f[e];
      }
      if (tcol < a.Cp) {
        uint16_t* q = x0 + m * a.Cp + tcol;
        store4(q, rr[pass]);
        store4(q + seg, f);
        if (a.alpha) store4(q + 2 * seg, xh);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  // items (block, update) of this wave: blocks blk0, blk0 + stride, ..., every
  // update of a block in turn; two statically named fragment sets
  long long blk = (long long)blockIdx.x * 4 + wave;
  int k = 0;
  auto advance = [&](long long& bb, int& kk) {
    if (++kk == a.n) { kk = 0; bb += stride; }
  };
  act8 af0[KSTEPS], af1[KSTEPS];
  load_a(blk, k, blk < nblk, af0);
  (void)nitem;
  while (blk < nblk) {
    long long nb = blk; int nk = k;
    advance(nb, nk);
    load_a(nb, nk, nb < nblk, af1);
    compute(blk, k, af0);
    blk = nb; k = nk;
    if (blk >= nblk) break;
    advance(nb, nk);
    load_a(nb, nk, nb < nblk, af0);
    compute(blk, k, af1);
    blk = nb; k = nk;
  }
}

// ---------------------------------------------------------------------------
// Wide form: K and / or N beyond 128 (BASELINE configs[4]: 512 -> 512 per
// timestep over 2 M rows -- as a cg_swconv launch it ran at 0.2 PFLOP/s, 19 % of
// that configuration's step).  W no longer fits a wave's registers, so a
// workgroup keeps ONE 128-column panel of it in LDS (K x 128 bf16 = 128 KB at K
// = 512, rows 16 bytes longer than a power of two: conflict-free ds_read_b128)
// for its whole life and streams 32-row blocks through it: the x fragments of a
// block (2 x K/32 16-byte loads per lane, MFMA layout, straight from global
// memory) are double-buffered in registers -- the next block's loads fly during
// this block's 16 K MFMA cycles --, every W fragment read from LDS feeds two
// MFMAs (the block's two 16-row halves: 128 B/clk/CU of LDS at full matrix rate).
// The weights are the MFMA's A operand, so a lane's accumulator registers are
// consecutive columns of one row and v_permlane16_swap makes them eight: the
// epilogue stores 32 (f32) or 16 (bf16) contiguous bytes per lane with no LDS.
// One workgroup per CU (4 waves, one per SIMD, ~340 registers each); the
// workgroups that walk the same rows through the N / 128 panels have ids
// congruent mod 8 (same XCD, same time: x comes from HBM once).
constexpr int kWideThreads = 256;
constexpr int kWideCols = 128;

struct DenseWideArgs {
  const uint16_t* x;    // bf16 [rows][Cx]
  const uint16_t* w;    // packed operand (taps 1, CK 32): [Npad][Cx/32][16][8]
  const float* bias;    // f32 [N] or null
  void* y;              // f32 or bf16 [rows][Cy]
  long long rows;
  int Cx, N, Cy, epilogue, out_act, panels;
  // interpolation epilogue (cg_dense_rows_interp at pitches beyond 128; real ==
  // nullptr: plain output into y).  Rows are [n_upd][seg_rows] (seg_rows = B * L, a
  // multiple of 32: a wave's 32-row block lies in one update); update k's rows go,
  // in bf16, to x0[k] = [real | fake_k | x^_k] (alpha == nullptr: no x^), pitch Cy
  const float* real;    // f32 [seg_rows][Cr]
  const float* alpha;   // f32 [n_upd * B] or null
  uint16_t* x0[kMaxInterp];
  long long seg_rows;
  int n_upd, L, Cr;
};

// (K = 128: the panel is 34 KB and the fragment sets 230 registers -- two
// workgroups per CU, twice the loads in flight: the cfg2 input gradient of the
// output Dense is latency-bound at one wave per SIMD, 2.8 TB/s)
template <int KSTEPS>
__global__ __launch_bounds__(kWideThreads, KSTEPS <= 4 ? 2 : 1) void dense_rows_wide_kernel(
    DenseWideArgs a) {
  constexpr int kRowB = KSTEPS * 64 + 16;  // LDS bytes per W column (k-major)
  extern __shared__ __attribute__((aligned(16))) unsigned char wl[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int r16 = lane & 15;
  const int g = lane >> 4;
  // workgroup -> (panel, row group): ids congruent mod 8 share an XCD
  const int xcd = blockIdx.x & 7;
  const int j = blockIdx.x >> 3;
  const int panel = j % a.panels;
  const int rgroup = (j / a.panels) * 8 + xcd;
  const int ngroups = (int)(gridDim.x / 8 / a.panels) * 8;
  const int n0 = panel * kWideCols;
  // the panel: column n0 + c, K-step ks, k-group gg -> 16 bytes
  for (int i = tid; i < kWideCols * KSTEPS * 4; i += kWideThreads) {
    const int c = i / (KSTEPS * 4);
    const int ks = (i / 4) % KSTEPS;
    const int gg = i & 3;
    const uint4 v = *reinterpret_cast<const uint4*>(
        a.w + (((long long)(n0 + c) * KSTEPS + ks) * 16 + gg) * 8);
    *reinterpret_cast<uint4*>(wl + c * kRowB + (ks * 4 + gg) * 16) = v;
  }
  __syncthreads();
  // lane's eight output columns per 32-column pair p (after the lane swap):
  // block 2p + (g & 1), columns 8 (g >> 1) .. + 7
  const int cq = (g & 1) * 16 + (g >> 1) * 8;
  float bv[4][8];
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int n = n0 + p * 32 + cq + e;
      bv[p][e] = (a.bias && n < a.N) ? a.bias[n] : 0.f;
    }
  const bool sig = a.epilogue == CG_EPI_SIGMOID;
  const int woff0 = r16 * kRowB + g * 16;

  const long long nblk = (a.rows + 31) / 32;
  const long long stride = (long long)ngroups * 4;
  long long blk = (long long)rgroup * 4 + wave;
  // (unconditional loads from a clamped row: with a predicate around them the
  // compiler cannot count them and waits for the NEXT block's loads -- vmcnt(0)
  // -- before computing this one; rows past the end are never stored)
  auto load_x = [&](long long b, act8 (&dst)[2][KSTEPS]) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      long long row = b * 32 + h * 16 + r16;
      row = row < a.rows ? row : a.rows - 1;
      const uint16_t* src = a.x + row * a.Cx + g * 8;
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks)
        dst[h][ks] = *reinterpret_cast<const act8*>(src + ks * 32);
    }
  };
  auto compute = [&](long long b, const act8 (&xf)[2][KSTEPS]) {
    f32x4 acc[2][8];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) acc[h][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // (opaque per block: the W fragments are the same for every block, and
    // hoisted out of the block loop all 8 K of them would sit in registers)
    // (the OFFSET is laundered, not the pointer: a laundered pointer is generic,
    // and its loads became flat_load + vmcnt(0) waits that also drained the x
    // prefetch)
    int woff = woff0;
    asm volatile("" : "+v"(woff));
    const unsigned char* wrow = wl + woff;
    // W fragments double-buffered by K-step: the eight reads of step ks + 1 go
    // out before the sixteen MFMAs of step ks, so their LDS latency sits behind
    // 256 matrix-pipe cycles (one wave per SIMD: nothing else would hide it).
    // The scheduling barriers keep the compiler from sinking the reads back to
    // their first use.
    act8 wa[8], wb[8];
    auto ldw = [&](act8 (&d)[8], int ks) {
#pragma unroll
      for (int nt = 0; nt < 8; ++nt)
        d[nt] = *reinterpret_cast<const act8*>(wrow + nt * 16 * kRowB + ks * 64);
    };
    auto mm = [&](const act8 (&w)[8], int ks) {
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
        acc[0][nt] = cg_mfma_16x16x32(w[nt], xf[0][ks], acc[0][nt], 0, 0, 0);
        acc[1][nt] = cg_mfma_16x16x32(w[nt], xf[1][ks], acc[1][nt], 0, 0, 0);
      }
    };
    static_assert(KSTEPS % 2 == 0, "K-steps are walked in pairs");
    ldw(wa, 0);
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ks += 2) {
      ldw(wb, ks + 1);
      __builtin_amdgcn_sched_barrier(0);
      mm(wa, ks);
      __builtin_amdgcn_sched_barrier(0);
      if (ks + 2 < KSTEPS) ldw(wa, ks + 2);
      __builtin_amdgcn_sched_barrier(0);
      mm(wb, ks + 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    // (interpolation epilogue: the block's update and its input buffer, uniform)
    const bool interp = a.n_upd > 0;
    long long kupd = 0;
    uint16_t* x0k = nullptr;
    if (interp) {
      kupd = (b * 32) / a.seg_rows;
      x0k = a.x0[kupd < a.n_upd ? kupd : 0];
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const long long m = b * 32 + h * 16 + r16;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        float v[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const auto sw = __builtin_amdgcn_permlane16_swap(
              __float_as_uint(acc[h][2 * p][r]),
              __float_as_uint(acc[h][2 * p + 1][r]), false, false);
          v[r] = __uint_as_float(sw[0]);
          v[4 + r] = __uint_as_float(sw[1]);
        }
        const int n = n0 + p * 32 + cq;
        if (m < a.rows && n < a.Cy) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float t = v[e] + bv[p][e];
            const float sg = 1.f / (1.f + __expf(-t));
            v[e] = (n + e < a.N) ? (sig ? sg : t) : 0.f;  // padding stays zero
          }
          if (interp) {
            // fake_k of the row's update in bf16, straight into its segment of x0[k]
            // -- stores only: with one wave per SIMD nothing would hide a load here
            // (the real segments come from real_bcast_kernel).  With `real` (a
            // caller that wants x^ = a real + (1 - a) fake, wgan_gp.py:38-41) the
            // row's real values are loaded and all three segments written, as
            // cg_interp_pack forms them from the f32 values.
            const long long rem = m - kupd * a.seg_rows;
            uint16_t* q = x0k + rem * a.Cy + n;
            const long long seg = a.seg_rows * a.Cy;
            *reinterpret_cast<uint4*>(q + seg) =
                make_uint4(pack2act(v[0], v[1]), pack2act(v[2], v[3]),
                           pack2act(v[4], v[5]), pack2act(v[6], v[7]));
            if (a.real) {
              const float* rp = a.real + rem * a.Cr + n;
              float rv[8], xh[8];
#pragma unroll
              for (int e = 0; e < 8; ++e) rv[e] = (n + e < a.N) ? rp[e] : 0.f;
              *reinterpret_cast<uint4*>(q) =
                  make_uint4(pack2act(rv[0], rv[1]), pack2act(rv[2], rv[3]),
                             pack2act(rv[4], rv[5]), pack2act(rv[6], rv[7]));
              if (a.alpha) {
                const float al =
                    a.alpha[kupd * (a.seg_rows / a.L) + (int)(rem / a.L)];
#pragma unroll
                for (int e = 0; e < 8; ++e) xh[e] = al * rv[e] + (1.f - al) * v[e];
                *reinterpret_cast<uint4*>(q + 2 * seg) =
                    make_uint4(pack2act(xh[0], xh[1]), pack2act(xh[2], xh[3]),
                               pack2act(xh[4], xh[5]), pack2act(xh[6], xh[7]));
              }
            }
          } else if (a.out_act) {
            *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(a.y) +
                                      m * a.Cy + n) =
                make_uint4(pack2act(v[0], v[1]), pack2act(v[2], v[3]),
                           pack2act(v[4], v[5]), pack2act(v[6], v[7]));
          } else {
            float* dst = reinterpret_cast<float*>(a.y) + m * a.Cy + n;
            *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>(dst + 4) = f32x4{v[4], v[5], v[6], v[7]};
          }
        }
      }
    }
  };
  act8 xf0[2][KSTEPS], xf1[2][KSTEPS];
  load_x(blk, xf0);
  while (blk < nblk) {
    load_x(blk + stride, xf1);
    compute(blk, xf0);
    blk += stride;
    if (blk >= nblk) break;
    load_x(blk + stride, xf0);
    compute(blk, xf1);
    blk += stride;
  }
}

template <int KSTEPS>
int launch_dense_wide(const DenseWideArgs& a, hipStream_t s) {
  const size_t lds = (size_t)kWideCols * (KSTEPS * 64 + 16);
  static CgPerDeviceFlag attr_set;
  if (!attr_set.test()) {
    hipError_t e = hipFuncSetAttribute(
        reinterpret_cast<const void*>(&dense_rows_wide_kernel<KSTEPS>),
        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set.mark();
  }
  // one workgroup per CU (two at K = 128): 8 XCDs x (32 / panels) row groups x panels
  const int per_xcd = (KSTEPS <= 4 ? 64 : 32) / a.panels * a.panels;
  hipLaunchKernelGGL(dense_rows_wide_kernel<KSTEPS>, dim3(8 * per_xcd),
                     dim3(kWideThreads), lds, s, a);
  CG_LAUNCH_CHECK();
}

struct DenseWideInterp {  // the interpolation epilogue's arguments (or null)
  const float* real;
  const float* alpha;
  void* const* x0;
  int n, B, L, Cr;
};

int dense_rows_wide(const void* x, const void* w, const float* bias, void* y,
                    long long rows, int Cx, int N, int Cy, int epilogue,
                    int out_act, hipStream_t s,
                    const DenseWideInterp* ip = nullptr) {
  if (Cx % 32 || N < 1 || Cy < N || Cy % 8) return CG_EINVAL;
  DenseWideArgs a;
  a.x = reinterpret_cast<const uint16_t*>(x);
  a.w = reinterpret_cast<const uint16_t*>(w);
  a.bias = bias;
  a.y = y;
  a.rows = rows;
  a.Cx = Cx; a.N = N; a.Cy = Cy; a.epilogue = epilogue; a.out_act = out_act;
  a.real = nullptr; a.alpha = nullptr; a.seg_rows = 1; a.n_upd = 0; a.L = 1; a.Cr = 0;
  for (int k = 0; k < kMaxInterp; ++k) a.x0[k] = nullptr;
  if (ip) {
    const long long seg = (long long)ip->B * ip->L;
    // (real == nullptr: the fake segments only -- then there is no x^ either)
    if (!ip->x0 || ip->n < 1 || ip->n > kMaxInterp || seg % 32 ||
        rows != seg * ip->n || (ip->real && ip->Cr < N) || (!ip->real && ip->alpha))
      return CG_EINVAL;
    a.real = ip->real; a.alpha = ip->alpha; a.seg_rows = seg;
    a.n_upd = ip->n; a.L = ip->L; a.Cr = ip->Cr;
    for (int k = 0; k < ip->n; ++k) {
      a.x0[k] = reinterpret_cast<uint16_t*>(ip->x0[k]);
      if (!a.x0[k]) return CG_EINVAL;
    }
  }
  // (the packed operand is padded to 128 columns: whole panels are readable)
  a.panels = (N + kWideCols - 1) / kWideCols;
  if (a.panels > 32) return CG_EINVAL;
  switch (Cx / 32) {
    case 4: return launch_dense_wide<4>(a, s);
    case 8: return launch_dense_wide<8>(a, s);
    case 12: return launch_dense_wide<12>(a, s);
    case 16: return launch_dense_wide<16>(a, s);
    default: return CG_EINVAL;
  }
}

// ---------------------------------------------------------------------------
// Weight gradient of the per-timestep Dense: dW[cx][cg] += sum_r x[r][cx] g[r][cg]
// (rows = B * L: 262 144 at cfg2, 2 M at configs[4]).  The generic cg_wgrad path
// (taps = 1: eight waves splitting the rows of one 32 x 64 tile, operands read in
// 64-byte slivers) ran it at 0.09-0.15 PFLOP/s.  Here a workgroup owns a
// 128 x 128 tile of dW and a contiguous range of rows: 32-row stages of x and g
// (256-byte row segments) go registers -> LDS (row-major, pitch 288 B = 32 x 9:
// conflict-free transpose reads) double-buffered against the next stage's
// global loads; each of the four waves keeps a 64 x 64 block in 64 registers,
// fragments come from ds_read_b64_tr_b16 (both operands are K'-major).  The
// workgroups of one row range (all tiles) share an XCD; partial tiles meet in dW
// through f32 atomics (a few MB per launch).
constexpr int kDwThreads = 256;
constexpr int kDwPitch = 144;  // elements: 128 channels + 16 pad

struct DenseWgradArgs {
  const uint16_t* x;
  const uint16_t* g;
  float* dw;
  long long rows, rows_per_split;
  int Cx, Cg, Cx_real, Cg_real, tiles_x, tiles_g;
  float* part;  // [row range][Cx_real][Cg_real] partial sums, or null: atomics
};

__device__ __forceinline__ s16x4 dw_tr_read(const uint16_t* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4*)(p));
}
__device__ __forceinline__ act8 dw_join(s16x4 lo, s16x4 hi) {
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(act8, v);
}

__global__ __launch_bounds__(kDwThreads, 2) void dense_wgrad_kernel(
    DenseWgradArgs a) {
  __shared__ __attribute__((aligned(16))) uint16_t lx[2][32 * kDwPitch];
  __shared__ __attribute__((aligned(16))) uint16_t lg[2][32 * kDwPitch];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int r16 = lane & 15, g4 = lane >> 4, q = r16 >> 2, p = r16 & 3;
  const int wx = wave >> 1, wg = wave & 1;
  // workgroup -> (tile, split): all tiles of one row range on one XCD
  const int ntiles = a.tiles_x * a.tiles_g;
  const int xcd = blockIdx.x & 7;
  const int j = blockIdx.x >> 3;
  const int tile = j % ntiles;
  const long long split = (long long)(j / ntiles) * 8 + xcd;
  const int cx0 = (tile % a.tiles_x) * 128;
  const int cg0 = (tile / a.tiles_x) * 128;
  const long long r_begin = split * a.rows_per_split;
  long long r_end = r_begin + a.rows_per_split;
  if (r_end > a.rows) r_end = a.rows;
  // (an empty row range still stores its zeros in the partial-sum form)
  if (r_begin >= r_end && !a.part) return;

  // staging map: 512 16-byte pieces per operand and stage, two per thread
  const int srow = tid >> 4;          // + 16 for the second piece
  const int sc = (tid & 15) * 8;      // channel of the piece inside the tile
  const bool okx = cx0 + sc < a.Cx, okg = cg0 + sc < a.Cg;
  // (four named registers and unconditional loads from clamped addresses: as
  // arrays behind predicates they were demoted to scratch)
  uint4 px0, px1, pg0, pg1;
  const int scx = okx ? cx0 + sc : 0, scg = okg ? cg0 + sc : 0;
  auto fetch1 = [&](long long row, uint4& vx, uint4& vg) {
    const bool ok = row < r_end;
    const long long rc = ok ? row : r_end - 1;
    vx = *reinterpret_cast<const uint4*>(a.x + rc * a.Cx + scx);
    vg = *reinterpret_cast<const uint4*>(a.g + rc * a.Cg + scg);
    if (!(ok && okx)) vx = make_uint4(0u, 0u, 0u, 0u);
    if (!(ok && okg)) vg = make_uint4(0u, 0u, 0u, 0u);
  };
  auto fetch = [&](long long r0) {
    fetch1(r0 + srow, px0, pg0);
    fetch1(r0 + srow + 16, px1, pg1);
  };
  auto stash = [&](int buf) {
    *reinterpret_cast<uint4*>(&lx[buf][srow * kDwPitch + sc]) = px0;
    *reinterpret_cast<uint4*>(&lg[buf][srow * kDwPitch + sc]) = pg0;
    *reinterpret_cast<uint4*>(&lx[buf][(srow + 16) * kDwPitch + sc]) = px1;
    *reinterpret_cast<uint4*>(&lg[buf][(srow + 16) * kDwPitch + sc]) = pg1;
  };
  f32x4 acc[4][4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  fetch(r_begin);
  stash(0);
  __syncthreads();
  int buf = 0;
  for (long long r0 = r_begin; r0 < r_end; r0 += 32) {
    const bool more = r0 + 32 < r_end;
    if (more) fetch(r0 + 32);  // in flight during this stage's MFMAs
    const uint16_t* bx = &lx[buf][(4 * g4 + q) * kDwPitch + wx * 64 + 4 * p];
    const uint16_t* bg = &lg[buf][(4 * g4 + q) * kDwPitch + wg * 64 + 4 * p];
    act8 bf[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
      bf[nt] = dw_join(dw_tr_read(bg + nt * 16), dw_tr_read(bg + 16 * kDwPitch + nt * 16));
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const act8 af =
          dw_join(dw_tr_read(bx + mt * 16), dw_tr_read(bx + 16 * kDwPitch + mt * 16));
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
        acc[mt][nt] = cg_mfma_16x16x32(af, bf[nt], acc[mt][nt], 0, 0, 0);
    }
    if (more) stash(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  // accumulator (mt, nt, r): cx = 4 g4 + r, cg = r16 of the 16 x 16 block
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int cx = cx0 + wx * 64 + mt * 16 + 4 * g4 + r;
        const int cg = cg0 + wg * 64 + nt * 16 + r16;
        if (cx < a.Cx_real && cg < a.Cg_real) {
          const long long e = (long long)cx * a.Cg_real + cg;
          if (a.part)
            a.part[split * ((long long)a.Cx_real * a.Cg_real) + e] = acc[mt][nt][r];
          else
            atomicAdd(a.dw + e, acc[mt][nt][r]);
        }
      }
}

// dW[e] = sum over the row ranges of their partial tiles, in a fixed order (few
// output tiles: a thousand workgroups adding into the same 10 K addresses
// serialise -- 150 us of atomics at cfg2 against 30 us of streaming).  Block =
// 64 elements x 16 shares: share j adds ranges j, j + 16, ... (a lone thread
// walking all of them is 512 dependent-latency loads: 100 us), the sixteen sums
// meet through LDS in share order; the result is STORED.
__global__ __launch_bounds__(1024) void dense_wgrad_reduce_kernel(
    const float* __restrict__ part, float* __restrict__ dw, long long n,
    int nsplit) {
  __shared__ float sm[16][64];
  const int lane = threadIdx.x & 63;
  const int j = threadIdx.x >> 6;
  const long long i = (long long)blockIdx.x * 64 + lane;
  float s = 0.f;
  if (i < n) {
    int k = j;
    for (; k + 48 < nsplit; k += 64) {
      const float a0 = part[(long long)k * n + i];
      const float a1 = part[(long long)(k + 16) * n + i];
      const float a2 = part[(long long)(k + 32) * n + i];
      const float a3 = part[(long long)(k + 48) * n + i];
      s += a0;
      s += a1;
      s += a2;
      s += a3;
    }
    for (; k < nsplit; k += 16) s += part[(long long)k * n + i];
  }
  sm[j][lane] = s;
  __syncthreads();
  if (j == 0 && i < n) {
    float t = sm[0][lane];
#pragma unroll
    for (int k = 1; k < 16; ++k) t += sm[k][lane];
    dw[i] = t;
  }
}

}  // namespace

extern "C" int cg_dense_rows(const void* x, const void* w, const float* bias,
                             float* y, long long rows, int Cx, int N, int Cy,
                             int epilogue, void* stream) {
  if (!x || !w || !y || rows < 1) return CG_EINVAL;
  if (epilogue != CG_EPI_NONE && epilogue != CG_EPI_SIGMOID) return CG_EINVAL;
  if (Cx > 128 || N > 128)  // W beyond a wave's registers: the LDS-panel form
    return dense_rows_wide(x, w, bias, y, rows, Cx, N, Cy, epilogue, 0,
                           (hipStream_t)stream);
  if (Cx % 32 || Cx < 32 || Cx > 128 || N < 1 || N > 128 || Cy < N || Cy > 128 ||
      Cy % 4)
    return CG_EINVAL;
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

// real (f32 [rows][Cr]) -> the bf16 real segment of every update's input buffer:
// one read, n writes (the LDS-panel form of cg_dense_rows_interp without x^)
struct RealBcastArgs {
  const float* real;
  uint16_t* x0[kMaxInterp];
  long long total8;
  int n, N, Cr, Cp;
};
__global__ __launch_bounds__(256) void real_bcast_kernel(RealBcastArgs a) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= a.total8) return;
  const int per_row = a.Cp / 8;
  const long long row = idx / per_row;
  const int c = (int)(idx - row * per_row) * 8;
  const float* rp = a.real + row * a.Cr + c;
  float v[8];
  if ((a.Cr & 3) == 0 && c + 8 <= a.N) {
    const f32x4 lo = *reinterpret_cast<const f32x4*>(rp);
    const f32x4 hi = *reinterpret_cast<const f32x4*>(rp + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = lo[e]; v[4 + e] = hi[e]; }
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (c + e < a.N) ? rp[e] : 0.f;
  }
  const uint4 o = make_uint4(pack2act(v[0], v[1]), pack2act(v[2], v[3]),
                             pack2act(v[4], v[5]), pack2act(v[6], v[7]));
  for (int k = 0; k < a.n; ++k)
    *reinterpret_cast<uint4*>(a.x0[k] + row * a.Cp + c) = o;
}

// cg_dense_rows for the fake batches of all n critic updates of a step, fused
// with cg_interp_pack: x0[k] (bf16 [3 B][L][Cp]) receives [real | fake_k | x^_k]
// (alpha == NULL: [real | fake_k | untouched]).
extern "C" int cg_dense_rows_interp(const void* x, const void* w, const float* bias,
                                    const float* real, const float* alpha,
                                    void* const* x0, int n, int B, int L, int Cx,
                                    int N, int Cr, int Cp, int epilogue,
                                    void* stream) {
  if (!x || !w || !real || !x0 || n < 1 || n > kMaxInterp || B < 1 || L < 16 ||
      L % 16)
    return CG_EINVAL;
  if (epilogue != CG_EPI_NONE && epilogue != CG_EPI_SIGMOID) return CG_EINVAL;
  if (Cx > 128 || N > 128) {
    // the LDS-panel form (BASELINE configs[4]: 512 -> 512): the same epilogue
    // behind its accumulators; B * L must be a multiple of 32
    if (Cp < N || Cp % 8) return CG_EINVAL;
    // without x^ the Dense only STORES (its fake segments) and one more launch
    // reads `real` once for the n real segments; with x^ its epilogue loads the
    // rows of `real` itself (exposed latency at one wave per SIMD: no faster than
    // the separate launches, kept for callers that convolve x^)
    const DenseWideInterp ip = {alpha ? real : nullptr, alpha, x0, n, B, L, Cr};
    const int rc = dense_rows_wide(x, w, bias, nullptr, (long long)n * B * L, Cx, N,
                                   Cp, epilogue, 0, (hipStream_t)stream, &ip);
    if (rc != 0 || alpha) return rc;
    RealBcastArgs ra;
    ra.real = real;
    for (int k = 0; k < kMaxInterp; ++k)
      ra.x0[k] = k < n ? reinterpret_cast<uint16_t*>(x0[k]) : nullptr;
    ra.total8 = (long long)B * L * (Cp / 8);
    ra.n = n; ra.N = N; ra.Cr = Cr; ra.Cp = Cp;
    hipLaunchKernelGGL(real_bcast_kernel,
                       dim3((unsigned)((ra.total8 + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, ra);
    CG_LAUNCH_CHECK();
  }
  if (Cx % 32 || Cx < 32 || Cx > 128 || N < 1 || N > 128 || Cp != 128 || Cr < N)
    return CG_EINVAL;
  DenseInterpArgs a;
  a.x = reinterpret_cast<const uint16_t*>(x);
  a.w = reinterpret_cast<const uint16_t*>(w);
  a.bias = bias; a.real = real; a.alpha = alpha;
  for (int k = 0; k < kMaxInterp; ++k)
    a.x0[k] = k < n ? reinterpret_cast<uint16_t*>(x0[k]) : nullptr;
  for (int k = 0; k < n; ++k)
    if (!a.x0[k]) return CG_EINVAL;
  a.n = n; a.B = B; a.L = L; a.Cx = Cx; a.N = N; a.Cr = Cr; a.Cp = Cp;
  a.epilogue = epilogue;
  const long long nblk = (long long)B * L / 16;
  long long blocks = (nblk + 3) / 4;
  if (blocks > 512) blocks = 512;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)blocks), block(kDrThreads);
  switch (Cx / 32) {
    case 1: hipLaunchKernelGGL(dense_rows_interp_kernel<1>, grid, block, 0, s, a); break;
    case 2: hipLaunchKernelGGL(dense_rows_interp_kernel<2>, grid, block, 0, s, a); break;
    case 3: hipLaunchKernelGGL(dense_rows_interp_kernel<3>, grid, block, 0, s, a); break;
    default: hipLaunchKernelGGL(dense_rows_interp_kernel<4>, grid, block, 0, s, a); break;
  }
  CG_LAUNCH_CHECK();
}

// The same contraction with an activation-typed (bf16 / fp16) output and no
// epilogue: the input gradient of the per-timestep Dense, dh = dz W^T (w = the
// transposed packed operand).  LDS-panel form for every size.
extern "C" int cg_dense_rows_act(const void* x, const void* w, void* y,
                                 long long rows, int Cx, int N, int Cy,
                                 void* stream) {
  if (!x || !w || !y || rows < 1) return CG_EINVAL;
  return dense_rows_wide(x, w, nullptr, y, rows, Cx, N, Cy, CG_EPI_NONE, 1,
                         (hipStream_t)stream);
}

// dW[cx][cg] += sum_r x[r][cx] * g[r][cg]  (f32 [Cx_real][Cg_real], row-major;
// the caller zeroes dW): the weight gradient of the per-timestep Dense.
extern "C" long long cg_dense_wgrad_ws_elems(long long rows, int Cx_real,
                                             int Cg_real);

namespace {
// row ranges of a launch: ~512 workgroups for up to four output tiles (the
// partial-sum form), ~1 024 beyond; whole 32-row stages; a multiple of 8 ranges
// (the XCD residue is part of the range id)
void dense_wgrad_split(long long rows, int ntiles, long long& nsplit,
                       long long& rps) {
  nsplit = (ntiles <= 4 ? 512 : 1024) / ntiles / 8 * 8;
  if (nsplit < 8) nsplit = 8;
  rps = (rows + nsplit - 1) / nsplit;
  rps = (rps + 31) / 32 * 32;
  nsplit = ((rows + rps - 1) / rps + 7) / 8 * 8;
}
}  // namespace

extern "C" long long cg_dense_wgrad_ws_elems(long long rows, int Cx_real,
                                             int Cg_real) {
  if (rows < 1 || Cx_real < 1 || Cg_real < 1) return -1;
  const int ntiles = ((Cx_real + 127) / 128) * ((Cg_real + 127) / 128);
  long long nsplit, rps;
  dense_wgrad_split(rows, ntiles, nsplit, rps);
  return nsplit * (long long)Cx_real * Cg_real;
}

// The weight gradient of the per-timestep Dense.  ws == NULL:
// dW[cx][cg] += sum_r x[r][cx] * g[r][cg] by f32 atomics (f32 [Cx_real][Cg_real],
// row-major; the caller zeroes dW).  With `ws` (cg_dense_wgrad_ws_elems floats):
// partial tiles + a reducing launch that adds them in a fixed order and STORES
// dW (no zeroed buffer, the same bits every run).
extern "C" int cg_dense_wgrad(const void* x, const void* g, float* dw,
                              long long rows, int Cx, int Cg, int Cx_real,
                              int Cg_real, float* ws, long long ws_elems,
                              void* stream) {
  if (!x || !g || !dw || rows < 1) return CG_EINVAL;
  if (Cx % 8 || Cg % 8 || Cx_real < 1 || Cg_real < 1 || Cx_real > Cx ||
      Cg_real > Cg)
    return CG_EINVAL;
  DenseWgradArgs a;
  a.x = reinterpret_cast<const uint16_t*>(x);
  a.g = reinterpret_cast<const uint16_t*>(g);
  a.dw = dw;
  a.rows = rows;
  a.Cx = Cx; a.Cg = Cg; a.Cx_real = Cx_real; a.Cg_real = Cg_real;
  a.tiles_x = (Cx_real + 127) / 128;
  a.tiles_g = (Cg_real + 127) / 128;
  const int ntiles = a.tiles_x * a.tiles_g;
  if (ntiles > 256) return CG_EINVAL;
  long long nsplit, rps;
  dense_wgrad_split(rows, ntiles, nsplit, rps);
  a.rows_per_split = rps;
  const long long n = (long long)Cx_real * Cg_real;
  if (ws && ws_elems < nsplit * n) return CG_EINVAL;
  a.part = ws;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(dense_wgrad_kernel, dim3((unsigned)(nsplit * ntiles)),
                     dim3(kDwThreads), 0, s, a);
  if (a.part)
    hipLaunchKernelGGL(dense_wgrad_reduce_kernel, dim3((unsigned)((n + 63) / 64)),
                       dim3(1024), 0, s, a.part, dw, n, (int)nsplit);
  CG_LAUNCH_CHECK();
}
