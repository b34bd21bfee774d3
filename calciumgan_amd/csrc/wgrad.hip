// Weight gradient of the sliding-window convolution on gfx950 MFMA.
//
//   dw[tap][cx][cg] += sum_{b,u} xs[b, R*u + off + tap, cx] * g[b, u, cg]
//
// GEMM view: M' = (tap, cx), N' = cg, K' = (b, u).  Both operands are
// channels-last, i.e. K'-strided, so the MFMA fragments come from
// ds_read_b64_tr_b16 (hardware 4x16 transpose read) on row-major LDS images:
//   * a tile of TT consecutive (b,u) rows stages g[TT][64] and the matching x
//     window (R*TT + taps - R rows x 32 channels, de-interleaved by row parity
//     for R = 2 so a tap walks stride-1 rows);
//   * block = 512 threads = 8 waves; for taps > 1 wave w owns taps w, w+8,
//     w+16 and keeps their 32x64 f32 accumulators (<= 96 VGPRs) for the whole
//     K' sweep, so x and g tiles are read once from HBM/L2 per (cx,cg) block;
//     for taps == 1 (Dense) the 8 waves split the 256 staged rows instead;
//   * K' is split over blockIdx.z; partial sums land with f32 atomics
//     (row-contiguous 64-B segments).
// LDS pitches are 32*odd bytes so each 32-lane half of a tr-read touches 8
// distinct 32-B bank groups (conflict-free).
#include "cg_common.h"

#include <string.h>

#include <map>
#include <mutex>
#include <vector>

namespace {

// -DCG_WGRAD_TRACE (tools/wgrad_trace.sh; never in the product library): cycles a
// wave of the batched launch spends per part of an item (s_memtime stamps).
#ifndef CG_WGRAD_SHIFT
#define CG_WGRAD_SHIFT 96
#endif
#ifdef CG_WGRAD_TRACE
constexpr int kWTraceParts = 5;
__device__ unsigned g_wgrad_trace[256 * 8 * kWTraceParts];
#define CG_WTR_PARAMS , unsigned (&wtr)[kWTraceParts], unsigned& wtt
#define CG_WTR_ARGS , wtr, wtt
#define CG_WTR(acc, t, part) do { unsigned long long n_; \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(n_)::"memory"); \
    (acc)[part] += (unsigned)n_ - (t); (t) = (unsigned)n_; } while (0)
#else
#define CG_WTR(acc, t, part)
#define CG_WTR_PARAMS
#define CG_WTR_ARGS
#endif

struct WgradArgs {
  const uint16_t* x;
  const uint16_t* g;
  float* dw;
  const int* shifts;
  int nB, Lx, Cx, seg_size;
  int Lu, Cg, M;
  int taps, off;
  int Cx_real, Cg_real;
  int S, log2S, nseg, WR;
  int ntiles;
  float* dbias;
  long long bias_rows;
  // K'-split partial sums as plain stores (reduced by wgrad_reduce_kernel)
  // instead of f32 atomics into dw; null: atomics
  float* part;
  // ... and the splits' bias column sums [gz][gy][64] (behind the dw partials in
  // the caller's workspace; added in split order by wgrad_reduce_kernel)
  float* bias_part;
  // a launch with ONE K' split owns every dw / dbias element once: stored
  // directly when the caller asked for `store` (else added, onto zeros)
  int direct_store;
  int pgx, pgy;  // (cx, cg) block grid of the launch
  // XCD-grouped block order (speed only): the gsz workgroups that stream the
  // same x and / or g tiles of one K' split get linear ids that differ by 8
  // (same XCD under round-robin placement, adjacent in dispatch order), so the
  // tile is fetched into that XCD's L2 once.  gmode 0: group = the gx cx-blocks
  // of one (cg block, split) [they share g]; 1: the gy cg-blocks of one (cx
  // block, split) [they share x]; 2: all gx * gy blocks of a split; 3: plain
  // order (bx fastest), no grouping.
  int gmode, gsz;
  // ring-staged body (wgrad_ring_loop): tiles arrive by LDS-DMA
  int ring;
};

// linear block id -> (bx, by, bz); false: padding id of the grouped order
__device__ __forceinline__ bool wgrad_block(const WgradArgs& a, int gz, int id,
                                            int& bx, int& by, int& bz) {
  const int xcd = id & 7;
  const int j = id >> 3;
  const int member = j % a.gsz;
  const int grp = (j / a.gsz) * 8 + xcd;
  if (a.gmode == 0) {
    bx = member;
    by = grp % a.pgy;
    bz = grp / a.pgy;
  } else if (a.gmode == 1) {
    by = member;
    bx = grp % a.pgx;
    bz = grp / a.pgx;
  } else if (a.gmode == 2) {
    bx = member % a.pgx;
    by = member / a.pgx;
    bz = grp;
  } else {  // 3: plain order
    bx = id % a.pgx;
    by = (id / a.pgx) % a.pgy;
    bz = id / (a.pgx * a.pgy);
  }
  return bz < gz;
}

constexpr int kPitchX = 48;  // 32 ch + 16 pad  (96 B = 32*3)
constexpr int kPitchG = 80;  // 64 ch + 16 pad  (160 B = 32*5)

__device__ __forceinline__ s16x4 tr_read(const uint16_t* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4*)(p));
}

__device__ __forceinline__ act8 join(s16x4 lo, s16x4 hi) {
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(act8, v);
}

// MFMA work of one staged tile: every wave reads its transposed fragments from
// the row-major LDS images and accumulates its taps.
template <int R, int TPW, bool ROWSPLIT, int TT, int ALLT>
__device__ __forceinline__ void wgrad_compute(const WgradArgs& a,
                                              const uint16_t* ldsX,
                                              const uint16_t* ldsG,
                                              int regionRows, int wave, int g4,
                                              int q, int p,
                                              f32x4 (&acc)[TPW][2][4]) {
  constexpr int KSTEPS = ROWSPLIT ? 1 : TT / 32;
  if constexpr (ALLT) {
    {
      // Every wave owns TPW live taps and the tile is one sample (rows of the
      // LDS images are linear): software-pipelined order.  The fragments of
      // group n+1 (one x fragment = one (tap, 16-channel half); at a K-step
      // boundary also the four g fragments of the next K-step) are read before
      // the four MFMAs of group n issue, and scheduling barriers keep that
      // order (left alone, hipcc sinks each read to just before its use: one
      // exposed LDS round trip per eight MFMAs).  All addresses are one lane
      // base per tap plus compile-time offsets.
      constexpr int NI = 2 * TPW;
      const uint16_t* gl = ldsG + (4 * g4 + q) * kPitchG + 4 * p;
      const uint16_t* xl[TPW];
#pragma unroll
      for (int s = 0; s < TPW; ++s) {
        const int tap = wave + 8 * s;
        const int toff = R == 2 ? ((tap & 1) * regionRows + (tap >> 1)) * kPitchX
                                : tap * kPitchX;
        xl[s] = ldsX + (4 * g4 + q) * kPitchX + 4 * p + toff;
      }
      act8 bfr[2][4], afr[2];
      auto load_b = [&](int kstep, act8 (&dst)[4]) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
          dst[nt] = join(tr_read(gl + kstep * 32 * kPitchG + nt * 16),
                         tr_read(gl + (kstep * 32 + 16) * kPitchG + nt * 16));
      };
      auto load_a = [&](int kstep, int i) {
        const uint16_t* b = xl[i >> 1] + kstep * 32 * kPitchX + (i & 1) * 16;
        return join(tr_read(b), tr_read(b + 16 * kPitchX));
      };
      load_b(0, bfr[0]);
      afr[0] = load_a(0, 0);
#pragma unroll
      for (int kstep = 0; kstep < KSTEPS; ++kstep) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          if (i + 1 < NI) {
            afr[(i + 1) & 1] = load_a(kstep, i + 1);
          } else if (kstep + 1 < KSTEPS) {
            // K-step boundary: the next step's g fragments ride along
            load_b(kstep + 1, bfr[(kstep + 1) & 1]);
            afr[0] = load_a(kstep + 1, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int nt = 0; nt < 4; ++nt)
            acc[i >> 1][i & 1][nt] = cg_mfma_16x16x32(
                afr[i & 1], bfr[kstep & 1][nt], acc[i >> 1][i & 1][nt], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      return;
    }
  }
#pragma unroll
  for (int kstep = 0; kstep < KSTEPS; ++kstep) {
    const int rbase = ROWSPLIT ? wave * 32 : kstep * 32;
    // tile rows of this lane's two transposed reads (k = 8*g4 + 4*h + q')
    const int i0 = rbase + 4 * g4 + q;
    const int i1 = i0 + 16;
    act8 bfrag[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
      bfrag[nt] = join(tr_read(ldsG + i0 * kPitchG + nt * 16 + 4 * p),
                       tr_read(ldsG + i1 * kPitchG + nt * 16 + 4 * p));
    const int x0 = ((i0 >> a.log2S) * a.WR + (i0 & (a.S - 1))) * kPitchX;
    const int x1 = ((i1 >> a.log2S) * a.WR + (i1 & (a.S - 1))) * kPitchX;
#pragma unroll
    for (int s = 0; s < TPW; ++s) {
      const int tap = ROWSPLIT ? 0 : wave + 8 * s;
      if (tap < a.taps) {  // wave-uniform
        int toff;
        if (R == 2)
          toff = ((tap & 1) * regionRows + (tap >> 1)) * kPitchX;
        else
          toff = tap * kPitchX;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const act8 afrag =
              join(tr_read(ldsX + x0 + toff + mt * 16 + 4 * p),
                   tr_read(ldsX + x1 + toff + mt * 16 + 4 * p));
#pragma unroll
          for (int nt = 0; nt < 4; ++nt)
            acc[s][mt][nt] = cg_mfma_16x16x32(
                afrag, bfrag[nt], acc[s][mt][nt], 0, 0, 0);
        }
      }
    }
  }
}

// Bias gradient for free: workgroups of the first cx chunk add the column sums
// of the g tile they have just staged (rows < bias_rows only).  Thread ->
// channel pair (tid & 31), row group tid >> 5 (16 groups).
template <int TT>
__device__ __forceinline__ void colsum_tile(const uint16_t* ldsG, int m0,
                                            long long bias_rows, int tid,
                                            float& s0, float& s1) {
  const int cp = (tid & 31) * 2;
  const int rg = tid >> 5;
  constexpr int RPG = TT / 16;
#pragma unroll
  for (int k = 0; k < RPG; ++k) {
    const int row = rg * RPG + k;
    if (m0 + row < bias_rows) {
      const uint32_t w =
          *reinterpret_cast<const uint32_t*>(ldsG + row * kPitchG + cp);
      s0 += act_lo(w);
      s1 += act_hi(w);
    }
  }
}

// ---------------------------------------------------------------------------
// Ring-staged K' sweep of the stride-2, 24-tap, one-sample-per-tile form (the
// critic's and the generator's conv layers: every wave owns three live taps).
//
//   * staging is LDS-DMA (global_load_lds_dwordx4): no staging registers, no
//     LDS store instructions, and NS = 4 tiles in flight instead of one.  A
//     tile is NP 1-KiB pieces -- the x window as [parity][WRP rows][64 B], then
//     g as [TT rows][128 B], both unpadded -- and wave w issues pieces w, w + 8,
//     ...: lane L lands on bytes [16 L, 16 L + 16) of the piece and fetches the
//     16-byte group that belongs there after the swizzle.
//   * unpadded rows need a swizzle for conflict-free transpose reads (a 32-lane
//     half reads 8 rows x 32 B): x rows swap their two 32-byte chunks on row
//     bit 2, g rows XOR their chunk index with row bits 1-2.
//   * the loop is software-pipelined over K-steps of 32 rows with two register
//     fragment sets: [reads of step k + 1 | 24 MFMAs of step k | wait].  The
//     one barrier per tile sits in front of the LAST step: every wave has then
//     issued (and waited for) all its reads of tile i, so the barrier both
//     publishes tile i + 1 (each wave first waits for its own DMAs of it) and
//     frees tile i's slot for the DMA of tile i + NS.
//   * reads and MFMAs are inline assembly (see cg_common.h): the compiler
//     would wait for every LDS-DMA in flight before an LDS read it can see.
// ---------------------------------------------------------------------------
template <int TT>
struct RingGeom {
  static constexpr int WRP = (TT + 11 + 7) / 8 * 8;  // LDS rows per source-row parity
  static constexpr int XB = 2 * WRP * 64;
  static constexpr int GB = TT * 128;
  static constexpr int STAGE = XB + GB;
  static constexpr int NPX = XB / 1024;
  static constexpr int NP = STAGE / 1024;
  static constexpr int NS = 4;
  static constexpr int NMIN = NP / 8;       // pieces every wave issues per tile
  static_assert(XB % 1024 == 0 && GB % 1024 == 0, "whole DMA pieces");
  static_assert(WRP % 8 == 0, "the parity offset keeps row bit 2");
};

template <int OFF>
__device__ __forceinline__ void lds_tr_read(s16x4& d, int addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field");
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}

struct RingFrags {  // two sets: the running K-step's and the next one's
  s16x4 al[2][6], ah[2][6];  // x: group = (tap slot, 16-channel half), K rows 0-15 / 16-31
  s16x4 bl[2][4], bh[2][4];  // g: 16-column group
};

// s_waitcnt lgkmcnt(0) that NAMES the fragment set its reads filled: the reads
// are inline assembly, so the compiler takes their outputs as defined when
// issued; whatever it builds from them (the register pairs join() forms, a copy)
// must not move in front of the wait.  The in/out operands make every later use
// of the set depend on this statement.
__device__ __forceinline__ void lds_wait_set(RingFrags& f, int set) {
#define CG_T2(a, i) "+v"(f.a[set][i])
  asm volatile("s_waitcnt lgkmcnt(0)"
               : CG_T2(al, 0), CG_T2(al, 1), CG_T2(al, 2), CG_T2(al, 3), CG_T2(al, 4),
                 CG_T2(al, 5), CG_T2(ah, 0), CG_T2(ah, 1), CG_T2(ah, 2), CG_T2(ah, 3),
                 CG_T2(ah, 4), CG_T2(ah, 5), CG_T2(bl, 0), CG_T2(bl, 1), CG_T2(bl, 2),
                 CG_T2(bl, 3), CG_T2(bh, 0), CG_T2(bh, 1), CG_T2(bh, 2), CG_T2(bh, 3)
               :
               : "memory");
#undef CG_T2
}

template <int TT>
__device__ __forceinline__ void wgrad_ring_loop(const WgradArgs& a,
                                                unsigned char* smem, int bx,
                                                int by, int bz, int gz, int tn,
                                                bool do_bias, int bias_cw, int bias_m,
                                                f32x4 (&acc)[3][2][4],
                                                float& bs0, float& bs1
                                                CG_WTR_PARAMS) {
  using G = RingGeom<TT>;
  using std::integral_constant;
  constexpr int KSTEPS = TT / 32;
  static_assert(KSTEPS % 2 == 0, "a tile starts on fragment set 0");
  // tiles bz, bz + gz, ... : `tn` of them (the flex form's contiguous range, gz =
  // 1), or every one below ntiles
  const int n_i = tn >= 0 ? tn : (bz < a.ntiles ? (a.ntiles - bz + gz - 1) / gz : 0);
  if (n_i == 0) return;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15;
  const int g4 = lane >> 4;
  const int q = r16 >> 2;
  const int p = r16 & 3;
  const int cx0 = bx * 32;
  const int cg0 = by * 64;
  const int lds0 = (int)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;

  // ---- DMA pieces of this lane (fixed across tiles) ---------------------------
  // x slot k -> piece 8 k + wave (the last slot only on the first NPX % 8 waves),
  // g slot k -> piece NPX + 8 k + wave.  Per lane: dk = source-row delta from
  // the tile's first window row, pre = byte offset of its 16-byte group from
  // that row; og = byte offset from the tile's first g row.  Lanes past the
  // channel pitch carry an out-of-range offset (they fetch zeros).
  constexpr int NXF = G::NPX / 8, NXT = G::NPX % 8, NGF = (G::NP - G::NPX) / 8;
  static_assert((G::NP - G::NPX) % 8 == 0, "g pieces split evenly over the waves");
  const int rowb = a.Cx * 2;  // bytes per x row
  int dk[NXF + 1], pre[NXF + 1], og[NGF];
#pragma unroll
  for (int k = 0; k <= NXF; ++k) {
    const int rho = (k * 8 + wave) * 16 + (lane >> 2);
    const int c4 = lane & 3;
    const int par = rho >= G::WRP ? 1 : 0;
    const int wr = rho - par * G::WRP;
    const int ch = (c4 >> 1) ^ ((rho >> 2) & 1);
    const int col = cx0 + (ch * 2 + (c4 & 1)) * 8;
    dk[k] = 2 * wr + par;
    pre[k] = dk[k] * rowb + col * 2;
    if (col >= a.Cx) {  // past the channel pitch: zeros on both paths below
      dk[k] = -(1 << 28);
      pre[k] = -1;
    }
  }
#pragma unroll
  for (int k = 0; k < NGF; ++k) {
    const int r = (k * 8 + wave) * 8 + (lane >> 3);
    const int c8 = lane & 7;
    const int ch = (c8 >> 1) ^ ((r >> 1) & 3);
    const int col = cg0 + (ch * 2 + (c8 & 1)) * 8;
    og[k] = col < a.Cg ? (r * a.Cg + col) * 2 : -1;
  }
  // Running coordinates of the next tile to issue (tiles go out in order, gz
  // apart): sample b, tile-in-sample ut, and the sample's shuffle segment --
  // no division in the loop.  The (at most 64: checked on the host) per-segment
  // shifts sit in one VGPR, lane = segment, and are picked with v_readlane: a
  // load in the loop would be a vector load the compiler waits for with
  // vmcnt(0), i.e. for every DMA in flight.
  const int tps = a.Lu / TT;  // tiles per sample
  const int step_b = gz / tps, step_t = gz - step_b * tps;
  int nb = bz / tps, nut = bz - nb * tps;
  int nsb = 0, nrb = 0;
  int shv = 0;
  if (a.shifts) {
    nsb = nb / a.seg_size;
    nrb = nb - nsb * a.seg_size;
    const int nsh = (a.nB + a.seg_size - 1) / a.seg_size;
    if (lane < nsh) shv = a.shifts[lane];
  }
  // The DMA is the buffer form (buffer_load_dwordx4 ... lds): a 32-bit per-lane
  // offset + a scalar offset against a resource descriptor, and a lane whose
  // offset is past num_records fetches ZEROS -- padding rows need no zero page
  // and no 64-bit select.  (The address arithmetic of the global form, ~20 VALU
  // + exec-masked branches per piece, was 3/4 of the staging cost.)  Interior
  // tiles (no padding, no reflection in the window) use `pre` as it is; edge
  // tiles recompute the row, branch-free.
  auto dma = [&](const __amdgpu_buffer_rsrc_t& r, int soff, int piece, unsigned vo,
                 unsigned so) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(
        r, (__attribute__((address_space(3))) void*)(smem + soff + piece * 1024), 16,
        (int)vo, (int)so, 0, 0);
  };
  auto issue_tile = [&](int soff) {
    const int b = nb;
    const int u0 = nut * TT;
    const int sft = a.shifts ? __builtin_amdgcn_readlane(shv, nsb) : 0;
    {  // advance
      nut += step_t;
      int db = step_b;
      if (nut >= tps) {
        nut -= tps;
        ++db;
      }
      nb += db;
      if (a.shifts) {
        nrb += db;
        while (nrb >= a.seg_size) {
          nrb -= a.seg_size;
          ++nsb;
        }
      }
    }
    const int srow0 = 2 * u0 + a.off;
    // no window row (of the WRP staged per parity) is padding or reflected
    const bool interior = srow0 + (sft < 0 ? sft : 0) >= 0 &&
                          srow0 + 2 * G::WRP - 1 + (sft > 0 ? sft : 0) < a.Lx;
    // descriptors based at the tile's sample / first g row (4 scalar moves per
    // tile): offsets stay 32-bit whatever the batch size
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint16_t*>(a.x + (long long)b * a.Lx * a.Cx), 0, 0x7fffffff,
        0x00020000);
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint16_t*>(a.g + ((long long)b * a.Lu + u0) * a.Cg), 0,
        0x7fffffff, 0x00020000);
    const unsigned xsample = 0u;
    if (interior) {
      const unsigned xs = (unsigned)((srow0 + sft) * rowb);
#pragma unroll
      for (int k = 0; k < NXF; ++k) dma(rx, soff, k * 8 + wave, (unsigned)pre[k], xs);
      if (NXT && wave < NXT) dma(rx, soff, NXF * 8 + wave, (unsigned)pre[NXF], xs);
    } else {
      // source row of window row d: u = |srow0 + d + sft| reflected at Lx - 1
      // (PhaseShuffle, cg_common.h shuffle_src), zeros where srow0 + d is padding
      auto edge = [&](int k) {
        const int sr = srow0 + dk[k];
        int u = sr + sft;
        u = u < 0 ? -u : u;
        const int v = 2 * (a.Lx - 1) - u;
        u = v < u ? v : u;
        unsigned vo = (unsigned)((u - dk[k]) * rowb) + (unsigned)pre[k];
        asm volatile("" : "+v"(vo));  // (computed on every lane: no exec-masked region)
        return (unsigned)sr < (unsigned)a.Lx ? vo : 0xffffffffu;
      };
#pragma unroll
      for (int k = 0; k < NXF; ++k) dma(rx, soff, k * 8 + wave, edge(k), xsample);
      if (NXT && wave < NXT) dma(rx, soff, NXF * 8 + wave, edge(NXF), xsample);
    }
    const unsigned gs = 0u;
#pragma unroll
    for (int k = 0; k < NGF; ++k) dma(rg, soff, G::NPX + k * 8 + wave, (unsigned)og[k], gs);
  };

  // ---- fragment addresses (absolute LDS bytes, slot 0) ------------------------
  int abase[3][2], bbase[4];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const int tap = wave + 8 * s;
    const int row0 = (tap & 1) * G::WRP + (tap >> 1) + 4 * g4 + q;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
      abase[s][mt] = lds0 + row0 * 64 + ((mt ^ ((row0 >> 2) & 1)) * 32) + 8 * p;
  }
  {
    const int rowb = 4 * g4 + q;
    const int sw = (rowb >> 1) & 3;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
      bbase[nt] = lds0 + G::XB + rowb * 128 + ((nt ^ sw) * 32) + 8 * p;
  }
  // Fragment traffic of one K-step (32 rows): six x fragments (group g = tap
  // slot g / 2, 16-channel half g % 2; 4 MFMAs each) and four g fragments (all
  // 24 MFMAs), 20 transpose reads, double-buffered in registers.  The reads of
  // the next step go out between the first MFMAs of the running one, two per
  // MFMA: all of them in front of the MFMAs hold the wave -- and, with all eight
  // waves in that burst at once, every matrix pipe -- for the LDS queue (measured:
  // +1.3 % .. +5 % per launch); spread thinner (one per MFMA, or a rolling
  // single-buffered prefetch with counted waits) the last ones return too late.
  RingFrags f;
  // read r (0..19) of K-step KS into set SET: g fragments first, then x in MFMA order
  auto read_one = [&](auto set_tag, auto ks_tag, auto r_tag) {
    constexpr int SET = decltype(set_tag)::value;
    constexpr int KS = decltype(ks_tag)::value;
    constexpr int r = decltype(r_tag)::value;
    if constexpr (r < 8) {
      constexpr int nt = r >> 1;
      if constexpr ((r & 1) == 0)
        lds_tr_read<KS * 32 * 128>(f.bl[SET][nt], bbase[nt]);
      else
        lds_tr_read<(KS * 32 + 16) * 128>(f.bh[SET][nt], bbase[nt]);
    } else {
      constexpr int g = (r - 8) >> 1;
      if constexpr ((r & 1) == 0)
        lds_tr_read<KS * 32 * 64>(f.al[SET][g], abase[g >> 1][g & 1]);
      else
        lds_tr_read<(KS * 32 + 16) * 64>(f.ah[SET][g], abase[g >> 1][g & 1]);
    }
  };
  // the step on set SET; its reads fetch K-step KSN into the other set for the
  // step after; `mid` runs behind MFMA 11 (the DMA issue of a tile's last step)
  auto step = [&](auto set_tag, auto ksn_tag, auto&& mid) {
    constexpr int SET = decltype(set_tag)::value;
    using NSET = integral_constant<int, SET ^ 1>;
    static_for<24>([&](auto j_tag) {
      constexpr int j = decltype(j_tag)::value;
      constexpr int g = j >> 2, nt = j & 3;
      mfma_acc(acc[g >> 1][g & 1][nt], join(f.al[SET][g], f.ah[SET][g]),
               join(f.bl[SET][nt], f.bh[SET][nt]));
      if constexpr (j < 10) {
        read_one(NSET{}, ksn_tag, integral_constant<int, 2 * j>{});
        read_one(NSET{}, ksn_tag, integral_constant<int, 2 * j + 1>{});
      }
      if constexpr (j == 11) mid();
    });
    lds_wait_set(f, SET ^ 1);
  };
  // bias gradient: the column sums of the g tiles come out of the g FRAGMENTS the
  // MFMAs read anyway (round 5).  Lane (n = lane & 15, k-group lane >> 4) of a B
  // fragment holds 8 k-values of column nt * 16 + n: wave w adds those of column
  // block nt = w & 3 for the K-steps k with (k & 1) == (w >> 2) -- every (K-step,
  // column block) of a tile has exactly one wave -- into ONE float per lane; the
  // lane groups and the two waves of a column block meet once per item (wgrad_body).
  // No extra LDS read: the separate column-sum reads (8 ds_read_b32 per thread and
  // tile) made the workgroups of cx block 0 live 8 % longer than the others, and
  // the launch ended with them (profiles/r05_wgrad_flex_trace.txt).
  // (bias_rows is a multiple of 32: checked on the host)
  const int bnt = wave & 3, bhalf = wave >> 2;
  // (v_dot2c_f32_{bf16,f16} against (1, 1): two activations into an f32 sum per
  // instruction, two independent chains -- 4 VALU per fragment pair instead of 16)
  auto dot_ones = [](uint32_t v, float c) {
#if CG_ACT_F16
    typedef __attribute__((ext_vector_type(2))) _Float16 a2;
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(a2, v),
                                  __builtin_bit_cast(a2, 0x3c003c00u), c, false);
#else
    typedef __attribute__((ext_vector_type(2))) __bf16 a2;
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(a2, v),
                                           __builtin_bit_cast(a2, 0x3f803f80u), c, false);
#endif
  };
  auto frag_sum = [&](const s16x4& lo, const s16x4& hi) {
    typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
    const u32x2 l = __builtin_bit_cast(u32x2, lo), h = __builtin_bit_cast(u32x2, hi);
    bs0 = dot_ones(l[0], bs0);
    bs1 = dot_ones(l[1], bs1);
    bs0 = dot_ones(h[0], bs0);
    bs1 = dot_ones(h[1], bs1);
  };
  auto frag_colsum = [&](auto set_tag) {
    constexpr int SET = decltype(set_tag)::value;
    switch (bnt) {  // (wave-uniform: a register array cannot be indexed at run time)
      case 0: frag_sum(f.bl[SET][0], f.bh[SET][0]); break;
      case 1: frag_sum(f.bl[SET][1], f.bh[SET][1]); break;
      case 2: frag_sum(f.bl[SET][2], f.bh[SET][2]); break;
      default: frag_sum(f.bl[SET][3], f.bh[SET][3]); break;
    }
  };

  // ---- prologue: NS tiles in flight, fragments of tile 0 / step 0 -------------
#pragma unroll
  for (int j = 0; j < G::NS; ++j)
    if (j < n_i) issue_tile(j * G::STAGE);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  static_for<20>([&](auto r_tag) {
    read_one(integral_constant<int, 0>{}, integral_constant<int, 0>{}, r_tag);
  });
  lds_wait_set(f, 0);
  CG_WTR(wtr, wtt, 0);  // item set-up, ring fill, first fragments

  {
    int soff = 0;  // ring slot (byte offset) of tile i
    for (int i = 0; i < n_i; ++i) {
      const int tile = bz + i * gz;
      const long long trow = (long long)tile * TT;
      // (flex form: the bias_cw team members that stream the same g tiles take
      // the column sums of every bias_cw-th tile each)
      const bool bias_tile =
          do_bias && (tile & (bias_cw - 1)) == bias_m && trow < a.bias_rows;
      static_for<KSTEPS>([&](auto k_tag) {
        constexpr int k = decltype(k_tag)::value;
        using SET = integral_constant<int, k & 1>;
        // this wave's share of the bias column sums: K-step k's g fragments
        const bool bsum = bias_tile && (k & 1) == bhalf && trow + k * 32 < a.bias_rows;
        if constexpr (k + 1 < KSTEPS) {
          step(SET{}, integral_constant<int, k + 1>{}, [&] {
            if (bsum) frag_colsum(SET{});
          });
        } else {
          // all but the DMAs of the tiles after i + 1 have landed (this wave's
          // share; the barrier makes it everyone's)
          const int later = n_i - i - 2;
          if (later >= 2)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * G::NMIN) : "memory");
          else if (later == 1)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::NMIN) : "memory");
          else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          const int delta = soff == (G::NS - 1) * G::STAGE
                                ? -(G::NS - 1) * G::STAGE
                                : G::STAGE;
#pragma unroll
          for (int s = 0; s < 3; ++s) {
            abase[s][0] += delta;
            abase[s][1] += delta;
          }
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) bbase[nt] += delta;
          // reads: step 0 of tile i + 1 (after the last tile: resident LDS
          // nobody uses); the DMA of tile i + NS goes into tile i's slot
          step(SET{}, integral_constant<int, 0>{}, [&] {
            if (bsum) frag_colsum(SET{});
            if (i + G::NS < n_i) issue_tile(soff);
          });
          soff += delta;
        }
      });
    }
  }
  // the accumulators are read by ordinary instructions next: let the matrix
  // pipe drain (the compiler cannot see the MFMAs)
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
}


// R: source stride (1|2).  TPW: taps per wave.  ROWSPLIT: taps == 1, the waves
// split the staged rows (TT = 256) instead of the taps (TT = 64).  PIPE: one
// sample per tile (nseg == 1): tiles are double-buffered in LDS and the next
// tile's global loads are issued before the current tile's MFMAs.
// ALLT (1 | 2): every wave owns TPW live taps (taps == 8 * TPW) and a tile is one
// sample: wgrad_compute runs its software-pipelined order.
// (bx, by, bz) / gz: this workgroup's (cx chunk, cg chunk, K' split) and the
// number of K' splits -- blockIdx / gridDim.z for a single launch, decoded from
// the linear block id by the multi-layer kernel below.  tn / pslot >= 0 (the flex
// form, ring-staged only): the item is the `tn` CONSECUTIVE tiles from bz on (gz =
// 1) and leaves its partial sums (and bias column sums) in slot `pslot`.
template <int R, int TPW, bool ROWSPLIT, bool PIPE, int TT, int ALLT>
__device__ __forceinline__ void wgrad_body(const WgradArgs& a, int bx, int by,
                                           int bz, int gz, int tn,
                                           int pslot, int bias_cw CG_WTR_PARAMS) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  static_assert(!ROWSPLIT || TT == 256, "row-split tiles are 256 rows");
  constexpr int NG = TT * 8 / 512;  // g pieces per thread (1, 2 or 4)
  const int regionRows = a.nseg * a.WR;
  const int bufX = R * regionRows * kPitchX;  // elements
  const int bufG = TT * kPitchG;
  uint16_t* lds = reinterpret_cast<uint16_t*>(smem);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int r16 = lane & 15;
  const int g4 = lane >> 4;
  const int q = r16 >> 2;
  const int p = r16 & 3;
  const int cx0 = bx * 32;
  const int cg0 = by * 64;

  f32x4 acc[TPW][2][4];
#pragma unroll
  for (int s = 0; s < TPW; ++s)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[s][mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int totalX = R * regionRows * 4;
  const int totalG = TT * 8;
  // (bias_cw > 1, flex form: the members bx = 0 .. bias_cw - 1 of the team that
  // holds cx block 0 share the tiles of the bias column sums)
  const bool do_bias = a.dbias != nullptr && bx < bias_cw;
  float bs0 = 0.f, bs1 = 0.f;

  if constexpr (ALLT == 2) {
    static_assert(R == 2 && TPW == 3 && PIPE && !ROWSPLIT, "ring-staged form");
    wgrad_ring_loop<TT>(a, smem, bx, by, bz, gz, tn, do_bias, bias_cw, bx, acc, bs0,
                        bs1 CG_WTR_ARGS);
    CG_WTR(wtr, wtt, 1);  // K' loop
  } else if (PIPE) {
    // per-thread piece coordinates (fixed across tiles)
    const int xq8 = tid & 3;
    const int xc = cx0 + xq8 * 8;
    const int xrowA = tid >> 2;          // piece 0
    const int xrowB = (tid + 512) >> 2;  // piece 1 (if tid + 512 < totalX)
    const bool hasB = tid + 512 < totalX;
    const int xrowC = (tid + 1024) >> 2;  // piece 2 (128-row tiles only)
    const bool hasC = TT == 128 && tid + 1024 < totalX;
    const int rhoC = (R == 2 && xrowC >= a.WR) ? 1 : 0;
    const int dC = R * (xrowC - rhoC * a.WR) + rhoC;
    const int rhoA = (R == 2 && xrowA >= a.WR) ? 1 : 0;
    const int rhoB = (R == 2 && xrowB >= a.WR) ? 1 : 0;
    const int dA = R * (xrowA - rhoA * a.WR) + rhoA;  // source row delta
    const int dB = R * (xrowB - rhoB * a.WR) + rhoB;
    const int gq8 = tid & 7;
    const int gc = cg0 + gq8 * 8;
    const int grow = tid >> 3;  // + 64 * j
    uint4 xa, xb, xc2, gr[NG];
    const uint4 zero = make_uint4(0u, 0u, 0u, 0u);

    auto load_tile = [&](int tile) {
      const int m0 = tile * TT;
      const int b = m0 / a.Lu;
      const int u0 = m0 - b * a.Lu;
      const int sft = a.shifts ? a.shifts[b / a.seg_size] : 0;
      const uint16_t* xbase = a.x + (long long)b * a.Lx * a.Cx + xc;
      const int srow0 = R * u0 + a.off;
      xa = zero;
      xb = zero;
      xc2 = zero;
      if (xc < a.Cx) {
        int sr = srow0 + dA;
        if (sr >= 0 && sr < a.Lx) {
          if (a.shifts) sr = shuffle_src(sr, sft, a.Lx);
          xa = *reinterpret_cast<const uint4*>(xbase + (long long)sr * a.Cx);
        }
        sr = srow0 + dB;
        if (hasB && sr >= 0 && sr < a.Lx) {
          if (a.shifts) sr = shuffle_src(sr, sft, a.Lx);
          xb = *reinterpret_cast<const uint4*>(xbase + (long long)sr * a.Cx);
        }
        sr = srow0 + dC;
        if (hasC && sr >= 0 && sr < a.Lx) {
          if (a.shifts) sr = shuffle_src(sr, sft, a.Lx);
          xc2 = *reinterpret_cast<const uint4*>(xbase + (long long)sr * a.Cx);
        }
      }
#pragma unroll
      for (int j = 0; j < NG; ++j) {
        const int m = m0 + grow + 64 * j;
        gr[j] = zero;
        if (m < a.M && gc < a.Cg)
          gr[j] = *reinterpret_cast<const uint4*>(a.g + (long long)m * a.Cg + gc);
      }
    };
    auto store_tile = [&](uint16_t* base) {
      uint16_t* lx = base;
      uint16_t* lg = base + bufX;
      *reinterpret_cast<uint4*>(lx + xrowA * kPitchX + xq8 * 8) = xa;
      if (hasB) *reinterpret_cast<uint4*>(lx + xrowB * kPitchX + xq8 * 8) = xb;
      if (hasC) *reinterpret_cast<uint4*>(lx + xrowC * kPitchX + xq8 * 8) = xc2;
#pragma unroll
      for (int j = 0; j < NG; ++j)
        *reinterpret_cast<uint4*>(lg + (grow + 64 * j) * kPitchG + gq8 * 8) =
            gr[j];
    };

    const int stride = gz;
    int tile = bz;
    int cur = 0;
    if (tile < a.ntiles) {
      load_tile(tile);
      store_tile(lds);
    }
    __syncthreads();
    for (; tile < a.ntiles; tile += stride) {
      const bool more = tile + stride < a.ntiles;
      if (more) load_tile(tile + stride);
      const uint16_t* base = lds + cur * (bufX + bufG);
      if (do_bias && (long long)tile * TT < a.bias_rows)
        colsum_tile<TT>(base + bufX, tile * TT, a.bias_rows, tid, bs0, bs1);
      wgrad_compute<R, TPW, ROWSPLIT, TT, ALLT>(a, base, base + bufX, regionRows, wave,
                                          g4, q, p, acc);
      if (more) {
        store_tile(lds + (cur ^ 1) * (bufX + bufG));
        __syncthreads();
        cur ^= 1;
      }
    }
  } else {
    uint16_t* ldsX = lds;
    uint16_t* ldsG = lds + bufX;
    for (int tile = bz; tile < a.ntiles; tile += gz) {
      const int m0 = tile * TT;
      __syncthreads();
      for (int idx = tid; idx < totalX; idx += 512) {
        const int row = idx >> 2;
        const int q8 = idx & 3;
        const int rho = row / regionRows;
        const int rem = row - rho * regionRows;
        const int seg = rem / a.WR;
        const int wr = rem - seg * a.WR;
        const int mseg = m0 + seg * a.S;
        const int c = cx0 + q8 * 8;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (mseg < a.M && c < a.Cx) {
          const int b = mseg / a.Lu;
          const int u0 = mseg - b * a.Lu;
          int srow = R * u0 + a.off + R * wr + rho;
          if (srow >= 0 && srow < a.Lx) {
            if (a.shifts)
              srow = shuffle_src(srow, a.shifts[b / a.seg_size], a.Lx);
            v = *reinterpret_cast<const uint4*>(
                a.x + ((long long)b * a.Lx + srow) * a.Cx + c);
          }
        }
        *reinterpret_cast<uint4*>(ldsX + row * kPitchX + q8 * 8) = v;
      }
      for (int idx = tid; idx < totalG; idx += 512) {
        const int row = idx >> 3;
        const int q8 = idx & 7;
        const int m = m0 + row;
        const int c = cg0 + q8 * 8;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (m < a.M && c < a.Cg)
          v = *reinterpret_cast<const uint4*>(a.g + (long long)m * a.Cg + c);
        *reinterpret_cast<uint4*>(ldsG + row * kPitchG + q8 * 8) = v;
      }
      __syncthreads();
      if (do_bias && (long long)m0 < a.bias_rows)
        colsum_tile<TT>(ldsG, m0, a.bias_rows, tid, bs0, bs1);
      wgrad_compute<R, TPW, ROWSPLIT, TT, ALLT>(a, ldsX, ldsG, regionRows, wave, g4, q,
                                          p, acc);
    }
  }
  if (do_bias) {
    // 16 row groups -> one value per channel through LDS, added in row-group
    // order; then the workgroup's own slot of the bias partials (summed over the
    // K' splits by the reducing launch) or one global atomic per channel
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
    if constexpr (ALLT == 2) {
      // ring form: one partial per lane (two chains) -- column (wave & 3) * 16 +
      // (lane & 15), k-group lane >> 4 -- from the g fragments (wgrad_ring_loop)
      red[tid] = bs0 + bs1;
    } else {
      const int cp = (tid & 31) * 2;
      red[(tid >> 5) * 64 + cp] = bs0;
      red[(tid >> 5) * 64 + cp + 1] = bs1;
    }
    __syncthreads();
    if (tid < 64) {
      float t;
      if constexpr (ALLT == 2) {
        // the two waves of the column block, four k-groups each, in a fixed order
        const int base = (tid >> 4) * 64 + (tid & 15);
        t = red[base];
#pragma unroll
        for (int k = 1; k < 4; ++k) t += red[base + k * 16];
#pragma unroll
        for (int k = 0; k < 4; ++k) t += red[base + 256 + k * 16];
      } else {
        t = red[tid];
#pragma unroll
        for (int k = 1; k < 16; ++k) t += red[k * 64 + tid];
      }
      if (a.bias_part)
        a.bias_part[(pslot >= 0 ? (long long)pslot : (long long)bz * a.pgy + by) * 64 +
                    tid] = t;
      else if (cg0 + tid < a.Cg_real) {
        if (a.direct_store) a.dbias[cg0 + tid] = t;
        else atomicAdd(a.dbias + cg0 + tid, t);
      }
    }
  }

  if (!ROWSPLIT && a.part) {
    // accumulators in register order: every store instruction of a wave is one
    // contiguous 1 KiB run; block (bz, by, bx), slot (s, mt, nt), thread
    const long long ps =
        pslot >= 0 ? (long long)pslot : (long long)(bz * a.pgy + by) * a.pgx + bx;
    float* pb = a.part + ps * (TPW * 8 * 2048) + tid * 4;
#pragma unroll
    for (int s = 0; s < TPW; ++s)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
          *reinterpret_cast<f32x4*>(pb + ((s * 2 + mt) * 4 + nt) * 2048) =
              acc[s][mt][nt];
    CG_WTR(wtr, wtt, 2);  // bias column sums + accumulator flush
    return;
  }
  if constexpr (ROWSPLIT) {
    // the eight waves hold partial sums of the SAME dw tile (they split the
    // rows): added through LDS in wave order, one adder per element
    static_assert(TPW == 1, "row-split tiles have one tap");
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);  // [8 waves][32 regs][64 lanes]
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          red[(wave * 32 + (mt * 4 + nt) * 4 + r) * 64 + lane] = acc[0][mt][nt][r];
    __syncthreads();
    // thread (wave, lane) finishes registers wave * 4 .. wave * 4 + 3
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const int reg = wave * 4 + q4;
      float t = red[reg * 64 + lane];
#pragma unroll
      for (int w = 1; w < 8; ++w) t += red[(w * 32 + reg) * 64 + lane];
      const int mt = reg >> 4, nt = (reg >> 2) & 3, r = reg & 3;
      const int cx = cx0 + mt * 16 + 4 * g4 + r;
      const int cg = cg0 + nt * 16 + r16;
      if (cx < a.Cx_real && cg < a.Cg_real) {
        float* q = a.dw + (long long)cx * a.Cg_real + cg;
        if (a.direct_store) *q = t;
        else atomicAdd(q, t);
      }
    }
    return;
  }
#pragma unroll
  for (int s = 0; s < TPW; ++s) {
    const int tap = ROWSPLIT ? 0 : wave + 8 * s;
    if (tap >= a.taps) continue;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int cx = cx0 + mt * 16 + 4 * g4 + r;
          const int cg = cg0 + nt * 16 + r16;
          if (cx < a.Cx_real && cg < a.Cg_real) {
            float* q = a.dw + ((long long)tap * a.Cx_real + cx) * a.Cg_real + cg;
            if (a.direct_store) *q = acc[s][mt][nt][r];
            else atomicAdd(q, acc[s][mt][nt][r]);
          }
        }
  }
}

template <int R, int TPW, bool ROWSPLIT, bool PIPE, int TT, int ALLT>
__global__ __launch_bounds__(512) void wgrad_kernel(WgradArgs a, int gz) {
  int bx, by, bz;
  if (!wgrad_block(a, gz, blockIdx.x, bx, by, bz)) return;
#ifdef CG_WGRAD_TRACE
  unsigned wtr[kWTraceParts] = {};
  unsigned wtt = 0;
#endif
  wgrad_body<R, TPW, ROWSPLIT, PIPE, TT, ALLT>(a, bx, by, bz, gz, -1, -1, 1 CG_WTR_ARGS);
}

// Second stage of the partial-sum path: an element (block tile, slot, tid) is one
// float4 of a workgroup's accumulator image; its K' splits lie `blk_stride` apart
// (each read one contiguous 4 KiB run per wave-instruction pair).  The splits of
// an element are shared by `zc` threads of a block (the layers with few dW
// elements have 64 splits: one thread walking them 4 at a time left the launch
// latency-bound on those layers after the wide ones had finished), each with up
// to 8 loads in flight; their sums meet in LDS and are added in chunk order.
// One thread owns the element's dw entries: no atomics, a fixed order.
struct ReduceItem {
  const float* part;
  float* dw;
  int gx, gy, gz, tpw, taps, Cx_real, Cg_real;
  int store;               // dw / dbias are stored, not added to
  int zc;                  // threads per element (power of two <= 8)
  const float* bias_part;  // [gz][gy][64] or null
  float* dbias;
};
struct ReduceArgs {
  int n;
  ReduceItem it[6];
};

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(ReduceArgs ra) {
  const ReduceItem& it = ra.it[blockIdx.y];
  __shared__ f32x4 red[256];
  if (it.bias_part && blockIdx.x == 0) {
    // conv bias gradient: the K' splits' column sums, in split order
    for (int c = threadIdx.x; c < it.gy * 64; c += 256) {
      float s = 0.f;
      for (int z = 0; z < it.gz; ++z) s += it.bias_part[(long long)z * it.gy * 64 + c];
      if (c < it.Cg_real) {
        if (it.store) it.dbias[c] = s;
        else it.dbias[c] += s;
      }
    }
  }
  const int per_tile = it.tpw * 8 * 512;  // float4 per block tile
  const long long total = (long long)it.gx * it.gy * per_tile;
  const long long blk_stride = (long long)it.gx * it.gy * per_tile * 4;  // floats per split
  const int zc = it.zc;
  const int epb = 256 / zc;                 // elements per block and round
  const int el = (int)threadIdx.x % epb;    // (consecutive lanes: consecutive float4)
  const int zi = (int)threadIdx.x / epb;
  const int chunk = (it.gz + zc - 1) / zc;
  const int z0 = zi * chunk;
  const int z1 = z0 + chunk < it.gz ? z0 + chunk : it.gz;
  for (long long base = (long long)blockIdx.x * epb; base < total;
       base += (long long)gridDim.x * epb) {
    const long long e = base + el;
    const bool live = e < total;
    const float* p = it.part + (live ? e : 0) * 4;
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    for (int z = z0; z < z1; z += 8) {
      f32x4 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        v[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (live && z + k < z1)
          v[k] = *reinterpret_cast<const f32x4*>(p + (z + k) * blk_stride);
      }
      sum += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
    if (zc > 1) {
      red[threadIdx.x] = sum;
      __syncthreads();
      if (zi == 0)
        for (int k = 1; k < zc; ++k) sum += red[k * epb + el];
      __syncthreads();
    }
    if (!live || zi != 0) continue;
    const int tile = (int)(e / per_tile);
    const int rem = (int)(e - (long long)tile * per_tile);
    const int slot = rem >> 9;
    const int tid = rem & 511;
    const int bx = tile % it.gx, by = tile / it.gx;
    const int s = slot >> 3, mt = (slot >> 2) & 1, nt = slot & 3;
    const int wave = tid >> 6, lane = tid & 63;
    const int tap = wave + 8 * s;
    const int cg = by * 64 + nt * 16 + (lane & 15);
    if (tap < it.taps && cg < it.Cg_real) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int cx = bx * 32 + mt * 16 + 4 * (lane >> 4) + r;
        if (cx < it.Cx_real) {
          float* q = it.dw + ((long long)tap * it.Cx_real + cx) * it.Cg_real + cg;
          *q = it.store ? sum[r] : *q + sum[r];
        }
      }
    }
  }
}

// Several layers' weight gradients in ONE launch: workgroup b does item b of
// layer 0, then item b of layer 1, ...  Every workgroup ends an item by adding
// its 196 KB of accumulators into dW with fire-and-forget f32 atomics; inside
// one launch those drain while the workgroup already runs the next layer's
// main loop, instead of holding the whole chip at a kernel boundary (24-38 us
// per layer as separate launches).
constexpr int kMaxBatch = 6;
struct WgradMulti {
  int n;
  int gx[kMaxBatch], gy[kMaxBatch], gz[kMaxBatch], tt[kMaxBatch];
  // entry e is worked on by the workgroups [lo, lo + cnt) only, which share its
  // K' splits [zofs, zofs + zcnt) of gz (the plain form: every workgroup, every
  // split).  "Halves": the layers are dealt to two halves of the grid and one
  // layer's splits are shared between them (two entries), so a workgroup flushes
  // its accumulators three times per pass instead of once per layer.
  int lo[kMaxBatch], cnt[kMaxBatch], zofs[kMaxBatch], zcnt[kMaxBatch];
  WgradArgs a[kMaxBatch];
};

// ALLT: 1 register-staged tiles, 2 the LDS-DMA ring (all layers of the launch)
template <int R, int TPW, int ALLT>
__global__ __launch_bounds__(512) void wgrad_multi_kernel(WgradMulti m) {
  // Item order rotated per 64 consecutive workgroups (the XCD groups that share
  // operand tiles stay together): with every workgroup on the same layer at the
  // same time, all 256 flush their 196 KB of accumulators at once, once per
  // layer; rotated, a quarter of them does while the others are in a K loop
  // (-0.5 % step, cg_wgrad +1.3 %; classes of 32 or 16 ids the same, of 8 worse).
#ifdef CG_WGRAD_NO_ROTATE
  const int rot = 0;
#else
  const int rot = ((int)blockIdx.x >> 6) % m.n;
#endif
#ifdef CG_WGRAD_TRACE
  unsigned wtr[kWTraceParts] = {};
  unsigned wtt;
  {
    unsigned long long n_;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(n_)::"memory");
    wtt = (unsigned)n_;
  }
#endif
  for (int k = 0; k < m.n; ++k) {
    int li = k + rot;
    if (li >= m.n) li -= m.n;
    const int gz = m.gz[li];
    int bx, by, bz;
    // A layer whose grid is smaller than the launch leaves ids without a share;
    // each layer hands its blocks out from another starting id (a multiple of
    // 8: XCD residues stay), so that the idle turns are not always the same
    // workgroups' (the last 32 ran 20 % less: tools/wgrad_trace.py).
    const int cnt = m.cnt[li];
    int id = (int)blockIdx.x - m.lo[li];
    const bool mine = id >= 0 && id < cnt;
#ifndef CG_WGRAD_NO_SHIFT
    id += (li * CG_WGRAD_SHIFT) % cnt;
    if (id >= cnt) id -= cnt;
#endif
    if (mine && wgrad_block(m.a[li], m.zcnt[li], id, bx, by, bz)) {
      bz += m.zofs[li];
      if (m.tt[li] == 128)
        wgrad_body<R, TPW, false, true, 128, ALLT>(m.a[li], bx, by, bz, gz, -1, -1, 1 CG_WTR_ARGS);
      else
        wgrad_body<R, TPW, false, true, 64, ALLT>(m.a[li], bx, by, bz, gz, -1, -1, 1 CG_WTR_ARGS);
    }
    CG_WTR(wtr, wtt, 3);  // (items this workgroup has no share of; flush of atomics forms)
    __syncthreads();  // LDS is reused by the next item
    CG_WTR(wtr, wtt, 4);  // barrier between items
  }
#ifdef CG_WGRAD_TRACE
  if ((threadIdx.x & 63) == 0 && blockIdx.x < 256)
    for (int k = 0; k < kWTraceParts; ++k)
      g_wgrad_trace[((int)blockIdx.x * 8 + (int)(threadIdx.x >> 6)) * kWTraceParts + k] = wtr[k];
#endif
}

// ---------------------------------------------------------------------------
// "Flex" form of the batched launch (round 5): the layers run SIDE BY SIDE.  All
// (layer, output tile, K' tile) work of the pass is one sequence, ordered (layer,
// column, K'), and cut into equal-cost shares; a share is a contiguous K' range
// that may end inside one column and continue in the next.  A column is up to S
// output tiles that stream the same x and / or g tiles (4 x 1, 1 x 4 or 2 x 2 of
// the (cx, cg) block grid); a TEAM of S workgroups, consecutive slots of one XCD,
// walks a share in step -- member m on the column's tile m -- so an operand tile
// is fetched into that XCD's L2 once, as in the XCD-grouped order above.  A
// workgroup flushes its accumulators once per (share, column) it touches: ~1.4
// times per pass instead of once per layer (plain form) or three times (halves),
// and the pass is balanced to one K' tile instead of to whole split counts.
// The host plans the shares (plan_flex) and hands every workgroup its items
// through a table in device memory; partial sums of one output tile occupy
// consecutive slots, in ascending K' order, and wgrad_flex_reduce_kernel adds
// them in that order (one owner per dW element: deterministic).
// ---------------------------------------------------------------------------
constexpr int kFlexMaxItems = 8;  // per workgroup
constexpr int kFlexItemInts = 6;  // layer (< 0: end), bx, by, first K' tile, tiles, slot

struct WgradFlex {
  int n;
  int tt[kMaxBatch];
  int cw[kMaxBatch];  // live cx blocks (power of two) of the team that holds cx block 0
  const int* table;  // [workgroups][kFlexMaxItems][kFlexItemInts]
  WgradArgs a[kMaxBatch];
};

template <int R, int TPW, int ALLT>
__global__ __launch_bounds__(512) void wgrad_flex_kernel(WgradFlex m) {
#ifdef CG_WGRAD_TRACE
  unsigned wtr[kWTraceParts] = {};
  unsigned wtt;
  {
    unsigned long long n_;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(n_)::"memory");
    wtt = (unsigned)n_;
  }
#endif
  const int* it = m.table + (size_t)blockIdx.x * (kFlexMaxItems * kFlexItemInts);
  for (int k = 0; k < kFlexMaxItems; ++k, it += kFlexItemInts) {
    const int li = __builtin_amdgcn_readfirstlane(it[0]);
    if (li < 0) break;
    const int bx = __builtin_amdgcn_readfirstlane(it[1]);
    const int by = __builtin_amdgcn_readfirstlane(it[2]);
    const int k0 = __builtin_amdgcn_readfirstlane(it[3]);
    const int kn = __builtin_amdgcn_readfirstlane(it[4]);
    const int ps = __builtin_amdgcn_readfirstlane(it[5]);
    if (m.tt[li] == 128)
      wgrad_body<R, TPW, false, true, 128, ALLT>(m.a[li], bx, by, k0, 1, kn, ps, m.cw[li] CG_WTR_ARGS);
    else
      wgrad_body<R, TPW, false, true, 64, ALLT>(m.a[li], bx, by, k0, 1, kn, ps, m.cw[li] CG_WTR_ARGS);
    CG_WTR(wtr, wtt, 3);
    __syncthreads();  // LDS is reused by the next item
    CG_WTR(wtr, wtt, 4);
  }
#ifdef CG_WGRAD_TRACE
  if ((threadIdx.x & 63) == 0 && blockIdx.x < 256)
    for (int k = 0; k < kWTraceParts; ++k)
      g_wgrad_trace[((int)blockIdx.x * 8 + (int)(threadIdx.x >> 6)) * kWTraceParts + k] = wtr[k];
#endif
}

// Reducing launch of the flex form: output tile t of a layer owns the slots
// [tiles[2 t], tiles[2 t] + tiles[2 t + 1]); otherwise as wgrad_reduce_kernel.
struct FlexReduceItem {
  const float* part;
  float* dw;
  const int* tiles;        // [gx * gy][2] in device memory
  int gx, gy, tpw, taps, Cx_real, Cg_real;
  int store, zc;
  int cw;                  // tiles (0 .. cw - 1, cg block) hold shares of the bias sums
  const float* bias_part;  // [slot][64] or null
  float* dbias;
};
struct FlexReduceArgs {
  int n;
  FlexReduceItem it[kMaxBatch];
};

__global__ __launch_bounds__(256) void wgrad_flex_reduce_kernel(FlexReduceArgs ra) {
  const FlexReduceItem& it = ra.it[blockIdx.y];
  __shared__ f32x4 red[256];
  if (it.bias_part && blockIdx.x == 0) {
    // conv bias gradient: column sums of the (cx block 0, cg block) tiles' items,
    // in slot (= ascending K') order
    for (int c = threadIdx.x; c < it.gy * 64; c += 256) {
      float s = 0.f;
      for (int bx = 0; bx < it.cw; ++bx) {  // (the team members' tile shares, in order)
        const int t = (c >> 6) * it.gx + bx;
        const int p0 = it.tiles[2 * t], cnt = it.tiles[2 * t + 1];
        for (int z = 0; z < cnt; ++z) s += it.bias_part[(long long)(p0 + z) * 64 + (c & 63)];
      }
      if (c < it.Cg_real) {
        if (it.store) it.dbias[c] = s;
        else it.dbias[c] += s;
      }
    }
  }
  const int per_tile = it.tpw * 8 * 512;  // float4 per slot
  const long long total = (long long)it.gx * it.gy * per_tile;
  const int zc = it.zc;
  const int epb = 256 / zc;  // elements per block and round (divides per_tile: one tile)
  const int el = (int)threadIdx.x % epb;
  const int zi = (int)threadIdx.x / epb;
  for (long long base = (long long)blockIdx.x * epb; base < total;
       base += (long long)gridDim.x * epb) {
    const long long e = base + el;
    const int tile = (int)(base / per_tile);
    const int rem = (int)(e - (long long)tile * per_tile);
    const int p0 = it.tiles[2 * tile], cnt = it.tiles[2 * tile + 1];
    const int chunk = (cnt + zc - 1) / zc;
    const int z0 = zi * chunk;
    const int z1 = z0 + chunk < cnt ? z0 + chunk : cnt;
    const float* p = it.part + ((long long)p0 * per_tile + rem) * 4;
    const long long stride = (long long)per_tile * 4;
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    for (int z = z0; z < z1; z += 8) {
      f32x4 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        v[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (z + k < z1) v[k] = *reinterpret_cast<const f32x4*>(p + (z + k) * stride);
      }
      sum += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
    if (zc > 1) {
      red[threadIdx.x] = sum;
      __syncthreads();
      if (zi == 0)
        for (int k = 1; k < zc; ++k) sum += red[k * epb + el];
      __syncthreads();
    }
    if (zi != 0) continue;
    const int slot = rem >> 9;
    const int tid = rem & 511;
    const int bx = tile % it.gx, by = tile / it.gx;
    const int s = slot >> 3, mt = (slot >> 2) & 1, nt = slot & 3;
    const int wave = tid >> 6, lane = tid & 63;
    const int tap = wave + 8 * s;
    const int cg = by * 64 + nt * 16 + (lane & 15);
    if (tap < it.taps && cg < it.Cg_real) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int cx = bx * 32 + mt * 16 + 4 * (lane >> 4) + r;
        if (cx < it.Cx_real) {
          float* q = it.dw + ((long long)tap * it.Cx_real + cx) * it.Cg_real + cg;
          *q = it.store ? sum[r] : *q + sum[r];
        }
      }
    }
  }
}

inline int ilog2(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}

// linear grid of the XCD-grouped block order: whole groups, a multiple of 8 of
// them (padding ids return at once)
inline unsigned wgrad_grid(const WgradArgs& a, int gz) {
  if (a.gmode == 3) return (unsigned)(a.pgx * a.pgy * gz);
  const int per_split = a.pgx * a.pgy / a.gsz;  // groups per K' split
  const int groups = per_split * gz;
  return (unsigned)((groups + 7) / 8 * 8 * a.gsz);
}

template <int R, int TPW, bool ROWSPLIT, bool PIPE, int TT, int ALLT = 0>
int launch_wgrad1(const WgradArgs& a, dim3 grid, size_t lds, hipStream_t s) {
  static CgPerDeviceFlag attr_set;
  if (!attr_set.test()) {
    hipError_t e = hipFuncSetAttribute(
        reinterpret_cast<const void*>(
            &wgrad_kernel<R, TPW, ROWSPLIT, PIPE, TT, ALLT>),
        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set.mark();
  }
  CG_LAUNCH_PROF(CG_FAMILY_WGRAD, (wgrad_kernel<R, TPW, ROWSPLIT, PIPE, TT, ALLT>),
                 dim3(wgrad_grid(a, grid.z)), dim3(512), lds, s, a, (int)grid.z);
  CG_LAUNCH_CHECK();
}

template <int R, int TPW, bool ROWSPLIT>
int launch_wgrad(const WgradArgs& a, dim3 grid, size_t lds, bool pipe, int tt,
                 hipStream_t s) {
  constexpr int T0 = ROWSPLIT ? 256 : 64;
  // pipe implies one sample per tile (nseg == 1)
  const bool allt = !ROWSPLIT && pipe && a.taps == 8 * TPW && a.nseg == 1;
  if constexpr (R == 2 && TPW == 3 && !ROWSPLIT) {
    if (a.ring && allt) {
      if (tt == 128)
        return launch_wgrad1<R, TPW, ROWSPLIT, true, 128, 2>(a, grid, 2 * lds, s);
      return launch_wgrad1<R, TPW, ROWSPLIT, true, 64, 2>(a, grid, 2 * lds, s);
    }
  }
  if (!ROWSPLIT && tt == 128) {  // only chosen with pipe
    if (allt)
      return launch_wgrad1<R, TPW, ROWSPLIT, true, ROWSPLIT ? 256 : 128,
                           !ROWSPLIT>(a, grid, 2 * lds, s);
    return launch_wgrad1<R, TPW, ROWSPLIT, true, ROWSPLIT ? 256 : 128>(
        a, grid, 2 * lds, s);
  }
  if (pipe && allt)
    return launch_wgrad1<R, TPW, ROWSPLIT, true, T0, !ROWSPLIT>(a, grid, 2 * lds,
                                                                s);
  if (pipe) return launch_wgrad1<R, TPW, ROWSPLIT, true, T0>(a, grid, 2 * lds, s);
  return launch_wgrad1<R, TPW, ROWSPLIT, false, T0>(a, grid, lds, s);
}

}  // namespace

namespace {

struct WgradPlan {
  WgradArgs a;
  int gx, gy, nsplit, TT, R, tpw;
  long long part_elems, bias_elems;
  int store;
  size_t lds, ring_lds;
  bool pipe, rowsplit;
};

int plan_wgrad(const cg_wgrad_desc* d, WgradPlan& p) {
  if (!d || !d->x || !d->g || !d->dw) return CG_EINVAL;
  const bool rowsplit = d->taps == 1;
  if (rowsplit) {
    if (d->stride != 1) return CG_EINVAL;
  } else {
    if (d->stride != 2 || (d->taps & 1) || d->taps > 24) return CG_EINVAL;
  }
  if (d->Cx % 8 || d->Cg % 8 || d->Cx_real > d->Cx || d->Cg_real > d->Cg)
    return CG_EINVAL;
  if (d->shifts && d->seg_size < 1) return CG_EINVAL;
  // 128-row tiles (half the per-tile barrier / staging overhead) when one
  // sample spans whole tiles, else 64
  int TT = rowsplit ? 256 : 64;
  if (!rowsplit && d->Lu % 128 == 0 && d->tile_rows != 64) TT = 128;
  if (d->tile_rows == 128 && (rowsplit || d->Lu % 128)) return CG_EINVAL;
  int S;
  if (d->Lu >= TT) {
    if (d->Lu % TT) return CG_EINVAL;
    S = TT;
  } else {
    if (TT % d->Lu) return CG_EINVAL;
    S = d->Lu;
  }
  WgradArgs& a = p.a;
  a.x = reinterpret_cast<const uint16_t*>(d->x);
  a.g = reinterpret_cast<const uint16_t*>(d->g);
  a.dw = d->dw;
  a.shifts = d->shifts;
  a.nB = d->nB; a.Lx = d->Lx; a.Cx = d->Cx; a.seg_size = d->seg_size;
  a.Lu = d->Lu; a.Cg = d->Cg; a.M = d->nB * d->Lu;
  a.taps = d->taps; a.off = d->off;
  a.Cx_real = d->Cx_real; a.Cg_real = d->Cg_real;
  a.S = S; a.log2S = ilog2(S); a.nseg = TT / S;
  const int R = d->stride;
  a.WR = S + d->taps / R - 1;
  a.ntiles = (a.M + TT - 1) / TT;
  a.dbias = d->dbias;
  a.bias_rows = d->bias_rows;
  const size_t lds =
      ((size_t)R * a.nseg * a.WR * kPitchX + (size_t)TT * kPitchG) * 2;
  if (lds > 160 * 1024) return CG_EINVAL;
  p.pipe = a.nseg == 1 && 2 * lds <= 160 * 1024;
  // ring-staged sweep (LDS-DMA, 4 tiles in flight): the stride-2 24-tap form
  // with one sample per tile
  a.ring = 0;
  p.ring_lds = 0;
  if (!rowsplit && p.pipe && d->taps == 24 && !d->classic_staging &&
      (long long)d->Lx * d->Cx * 2 < (1ll << 31) &&
      (!d->shifts || (d->nB + d->seg_size - 1) / d->seg_size <= 64) &&
      // (the ring form sums the bias columns per 32-row K-step)
      (!d->dbias || d->bias_rows % 32 == 0)) {
    a.ring = 1;
    p.ring_lds = TT == 128 ? 4 * RingGeom<128>::STAGE : 4 * RingGeom<64>::STAGE;
  }
  p.gx = (d->Cx_real + 31) / 32;
  p.gy = (d->Cg_real + 63) / 64;
  int nsplit = d->nsplit;
  if (nsplit <= 0) {
    // one 8-wave workgroup per CU is resident; every extra K' split costs a
    // full dW tile of f32 atomics (chip-wide ~1.8 TB/s), so split just enough
    // to fill the 256 CUs once
    nsplit = 256 / (p.gx * p.gy);
    if (nsplit < 1) nsplit = 1;
  }
  if (nsplit > a.ntiles) nsplit = a.ntiles;
  if (nsplit < 1) nsplit = 1;
  p.nsplit = nsplit;
  p.TT = TT; p.R = R; p.rowsplit = rowsplit;
  // (pipelined launches ask for 2 * p.lds bytes: two register-staged tiles, or
  // the ring)
  p.lds = a.ring ? p.ring_lds / 2 : lds;
  p.tpw = rowsplit ? 1 : (d->taps <= 8 ? 1 : (d->taps <= 16 ? 2 : 3));
  a.part = nullptr;
  a.pgx = p.gx; a.pgy = p.gy;
  // XCD grouping: a group must pack the 32 CUs of an XCD (size divides 32);
  // among the feasible groupings the one with the least re-read bytes
  {
    auto fits = [](int n) { return n >= 1 && n <= 32 && 32 % n == 0; };
    const double xb = (double)d->nB * d->Lx * d->Cx, gb = (double)a.M * d->Cg;
    a.gmode = 3; a.gsz = 1;
    double best = xb * p.gy + gb * p.gx;  // no sharing
    if (fits(p.gx) && xb * p.gy + gb < best) {
      best = xb * p.gy + gb; a.gmode = 0; a.gsz = p.gx;
    }
    if (fits(p.gy) && xb + gb * p.gx < best) {
      best = xb + gb * p.gx; a.gmode = 1; a.gsz = p.gy;
    }
    if (fits(p.gx * p.gy) && !d->no_xcd_group) {
      a.gmode = 2; a.gsz = p.gx * p.gy;
    }
    if (d->no_xcd_group) { a.gmode = 3; a.gsz = 1; }
  }
  const long long dw_part = (rowsplit || nsplit < 2)
                                ? 0
                                : (long long)p.gx * p.gy * nsplit * p.tpw * 8 * 2048;
  p.bias_elems = (dw_part && d->dbias) ? (long long)nsplit * p.gy * 64 : 0;
  p.part_elems = dw_part + p.bias_elems;
  a.bias_part = nullptr;
  a.direct_store = 0;
  p.store = 0;
  if (d->partials && p.part_elems > 0) {
    if (d->partials_elems < p.part_elems) return CG_EINVAL;
    a.part = d->partials;
    if (p.bias_elems) a.bias_part = d->partials + dw_part;
    p.store = d->store ? 1 : 0;
  } else if (d->store) {
    // stores need a single owner per element: one K' split
    if (nsplit != 1) return CG_EINVAL;
    a.direct_store = 1;
  }
  return 0;
}

void reduce_item(const WgradPlan& p, ReduceItem& it) {
  it.part = p.a.part; it.dw = p.a.dw;
  it.gx = p.gx; it.gy = p.gy; it.gz = p.nsplit; it.tpw = p.tpw;
  it.taps = p.a.taps; it.Cx_real = p.a.Cx_real; it.Cg_real = p.a.Cg_real;
  it.store = p.store; it.bias_part = p.a.bias_part; it.dbias = p.a.dbias;
  // threads per element: chunks of at most 8 splits where 8 threads allow it
  it.zc = 1;
  while (it.zc < 8 && (it.gz + it.zc - 1) / it.zc > 8) it.zc *= 2;
}

int launch_reduce(const ReduceArgs& ra, hipStream_t s) {
  long long most = 0;
  for (int i = 0; i < ra.n; ++i) {
    const long long t = (long long)ra.it[i].gx * ra.it[i].gy * ra.it[i].tpw * 8 * 512 *
                        ra.it[i].zc;
    if (t > most) most = t;
  }
  long long bx = (most + 255) / 256;
  if (bx > 2048) bx = 2048;
  // (timed with the wgrad family: it is the second half of those launches)
  CG_LAUNCH_PROF(CG_FAMILY_WGRAD, wgrad_reduce_kernel, dim3((unsigned)bx, ra.n),
                 dim3(256), 0, s, ra);
  CG_LAUNCH_CHECK();
}

int launch_plan(const WgradPlan& p, hipStream_t s) {
  dim3 grid(p.gx, p.gy, p.nsplit);
  if (p.rowsplit)
    return launch_wgrad<1, 1, true>(p.a, grid, p.lds, p.pipe, p.TT, s);
  if (p.tpw == 1)
    return launch_wgrad<2, 1, false>(p.a, grid, p.lds, p.pipe, p.TT, s);
  if (p.tpw == 2)
    return launch_wgrad<2, 2, false>(p.a, grid, p.lds, p.pipe, p.TT, s);
  return launch_wgrad<2, 3, false>(p.a, grid, p.lds, p.pipe, p.TT, s);
}

template <int TPW, int ALLT = 1>
int launch_multi(const WgradMulti& m, int blocks, size_t lds, hipStream_t s) {
  static CgPerDeviceFlag attr_set;
  if (!attr_set.test()) {
    hipError_t e = hipFuncSetAttribute(
        reinterpret_cast<const void*>(&wgrad_multi_kernel<2, TPW, ALLT>),
        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set.mark();
  }
  CG_LAUNCH_PROF(CG_FAMILY_WGRAD, (wgrad_multi_kernel<2, TPW, ALLT>), dim3(blocks),
                 dim3(512), lds, s, m);
  CG_LAUNCH_CHECK();
}

}  // namespace

extern "C" int cg_wgrad(const cg_wgrad_desc* d, void* stream) {
  WgradPlan p;
  int rc = plan_wgrad(d, p);
  if (rc) return rc;
  rc = launch_plan(p, (hipStream_t)stream);
  if (rc || !p.a.part) return rc;
  ReduceArgs ra;
  ra.n = 1;
  reduce_item(p, ra.it[0]);
  return launch_reduce(ra, (hipStream_t)stream);
}

extern "C" long long cg_wgrad_partials_elems(const cg_wgrad_desc* d) {
  if (!d) return -1;
  cg_wgrad_desc c = *d;
  c.partials = nullptr;
  c.store = 0;  // (a size query: `store` without a workspace is its own error)
  WgradPlan p;
  const int rc = plan_wgrad(&c, p);
  return rc ? -1 : p.part_elems;
}

// entries of a deal: the two halves' layers alternate (the rotation of the
// kernel's item order starts neighbouring 64-id classes on different entries),
// the shared layer last
static bool apply_halves(const WgradPlan* np, int n, int best_s, int best_mask, int zs,
                         WgradPlan* plans, WgradMulti& m, int& blocks, size_t& lds) {
  constexpr int G = 128;
  int order[kMaxBatch], half[kMaxBatch], ne = 0;
  {
    int a0[kMaxBatch], n0 = 0, a1[kMaxBatch], n1 = 0;
    for (int i = 0; i < n; ++i)
      if (i != best_s) (((best_mask >> i) & 1) ? a1[n1++] : a0[n0++]) = i;
    for (int k = 0; k < n0 || k < n1; ++k) {
      if (k < n0) { order[ne] = a0[k]; half[ne++] = 0; }
      if (k < n1) { order[ne] = a1[k]; half[ne++] = 1; }
    }
    order[ne] = best_s; half[ne++] = 0;
    order[ne] = best_s; half[ne++] = 1;
  }
  m.n = ne;
  blocks = 2 * G;
  lds = 0;
  for (int e = 0; e < ne; ++e) {
    const int i = order[e];
    const WgradPlan& p = np[i];
    m.a[e] = p.a;
    m.gx[e] = p.gx; m.gy[e] = p.gy; m.gz[e] = p.nsplit; m.tt[e] = p.TT;
    m.lo[e] = half[e] * G; m.cnt[e] = G;
    m.zofs[e] = (i == best_s && half[e]) ? zs : 0;
    m.zcnt[e] = i == best_s ? zs : p.nsplit;
    if (2 * p.lds > lds) lds = 2 * p.lds;
  }
  for (int i = 0; i < n; ++i) plans[i] = np[i];
  return true;
}

// "Halves" form of the batched launch (WgradMulti): deal the layers to two halves
// of a 256-workgroup grid so that both halves carry the same multiply-adds, the
// K' splits of one layer shared between them.  Every layer is re-planned with
// the split count that fills ONE half, so a pass writes (and the reducing launch
// reads) ~0.6 x the partial sums.  Returns false -- plans untouched -- when the
// launch does not have this shape or no deal balances within 4 %
// (CALCIUMGAN_WGRAD_HALVES=0: never).
static bool plan_halves(const cg_wgrad_desc* descs, int n, WgradPlan* plans,
                        WgradMulti& m, int& blocks, size_t& lds) {
  static int enabled = -1;
  if (enabled < 0) {
    const char* e = getenv("CALCIUMGAN_WGRAD_HALVES");
    enabled = (e && e[0] == '0') ? 0 : (e && e[0] == '2') ? 2 : 1;  // 2: print deals
  }
  constexpr int G = 128;  // workgroups per half (a 256-CU chip: checked below)
  if (!enabled || n < 3 || n + 1 > kMaxBatch) return false;
  {
    static int cus[kCgMaxDevices];  // (the halves assume 256 CUs)
    int& c = cus[cg_device_index()];
    if (!c && hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount,
                                    cg_device_index()) != hipSuccess)
      c = 256;
    if (c != 2 * G) return false;
  }
  // The deal is a search over split counts and layer subsets: memoised on the
  // launch's geometry (it used to be re-derived on every eager call).
  struct Deal { bool ok; int best_s, best_mask, zs; int nsplit[kMaxBatch]; };
  static std::mutex deal_mutex;
  static std::map<std::vector<long long>, Deal> deals;
  std::vector<long long> key;
  for (int i = 0; i < n; ++i)
    for (long long v : {(long long)plans[i].gx, (long long)plans[i].gy, (long long)plans[i].a.M,
                        (long long)plans[i].a.taps, (long long)plans[i].a.ntiles,
                        (long long)plans[i].TT, (long long)descs[i].store,
                        (long long)(plans[i].a.part != nullptr), (long long)descs[i].no_xcd_group})
      key.push_back(v);
  {
    std::lock_guard<std::mutex> lock(deal_mutex);
    auto f = deals.find(key);
    if (f != deals.end()) {
      const Deal& dl = f->second;
      if (!dl.ok) return false;
      WgradPlan np[kMaxBatch];
      for (int i = 0; i < n; ++i) {
        cg_wgrad_desc d = descs[i];
        d.nsplit = dl.nsplit[i];
        if (plan_wgrad(&d, np[i]) || np[i].nsplit != d.nsplit) return false;
      }
      return apply_halves(np, n, dl.best_s, dl.best_mask, dl.zs, plans, m, blocks, lds);
    }
  }
  auto remember = [&](bool ok, int bs, int bm, int zs_, const WgradPlan* np) {
    Deal dl{ok, bs, bm, zs_, {}};
    for (int i = 0; i < n && np; ++i) dl.nsplit[i] = np[i].nsplit;
    std::lock_guard<std::mutex> lock(deal_mutex);
    deals[key] = dl;
    return ok;
  };
  double cost[kMaxBatch];
  for (int i = 0; i < n; ++i) {
    const WgradPlan& p = plans[i];
    if (!p.a.part && !descs[i].store) {
      // (atomics forms work too, but nothing is gained: no partial sums)
      return false;
    }
    if ((int)wgrad_grid(p.a, p.nsplit) > 2 * G || p.gx * p.gy > G) return false;
    cost[i] = (double)p.a.M * p.a.taps * (p.gx * 32.0) * (p.gy * 64.0);
  }
  // split counts that fill ONE half: z_ex as a half's own layer, 2 * z_sh as the
  // layer both halves share (whole XCD groups of the block order must fit)
  WgradPlan pex[kMaxBatch], psh[kMaxBatch];
  int z_ex[kMaxBatch], z_sh[kMaxBatch];
  double t_plain = 0;
  for (int i = 0; i < n; ++i) {
    const int tiles = plans[i].gx * plans[i].gy;
    t_plain += cost[i] / ((double)tiles * plans[i].nsplit);
    z_ex[i] = z_sh[i] = 0;
    for (int shared = 0; shared < 2; ++shared) {
      cg_wgrad_desc d = descs[i];
      WgradPlan& q = shared ? psh[i] : pex[i];
      for (int z = G / tiles; z >= 1; --z) {
        d.nsplit = shared ? 2 * z : z;
        if (plan_wgrad(&d, q)) return false;
        if (q.nsplit != d.nsplit) continue;  // (clamped: too few K' tiles)
        if ((int)wgrad_grid(q.a, z) <= G) { (shared ? z_sh : z_ex)[i] = z; break; }
      }
    }
  }
  // the deal with the shortest longest half, in per-workgroup time (one item of
  // every layer the half holds): cost / items, items = tiles x splits
  int best_s = -1, best_mask = 0;
  double best = 1e30;
  for (int sh = 0; sh < n; ++sh) {
    if (!z_sh[sh]) continue;
    const double tsh = cost[sh] / ((double)plans[sh].gx * plans[sh].gy * 2 * z_sh[sh]);
    for (int mask = 0; mask < (1 << n); ++mask) {
      if (mask & (1 << sh)) continue;
      double t0 = tsh, t1 = tsh;
      bool ok = true;
      int n0 = 0, n1 = 0;
      for (int i = 0; i < n && ok; ++i) {
        if (i == sh) continue;
        if (!z_ex[i]) { ok = false; break; }
        const double ti = cost[i] / ((double)plans[i].gx * plans[i].gy * z_ex[i]);
        if ((mask >> i) & 1) { t1 += ti; ++n1; } else { t0 += ti; ++n0; }
      }
      if (!ok || !n0 || !n1) continue;
      const double tm = t0 > t1 ? t0 : t1;
      if (tm < best) { best = tm; best_s = sh; best_mask = mask; }
    }
  }
  // (the plain form's own figure has its idle workgroups in it: allow 4 % on top)
  if (best_s < 0 || best > 1.04 * t_plain) return remember(false, -1, 0, 0, nullptr);
  WgradPlan np[kMaxBatch];
  for (int i = 0; i < n; ++i) np[i] = i == best_s ? psh[i] : pex[i];
  const int zs = z_sh[best_s];
  remember(true, best_s, best_mask, zs, np);
  if (enabled == 2) {
    fprintf(stderr, "cg_wgrad_batched halves: shared layer %d (2 x %d splits), half 1 = mask 0x%x, "
            "longest half %.3f of the plain form's time; splits", best_s, zs, best_mask,
            best / t_plain);
    for (int i = 0; i < n; ++i) fprintf(stderr, " %d", np[i].nsplit);
    fprintf(stderr, "\n");
  }
  return apply_halves(np, n, best_s, best_mask, zs, plans, m, blocks, lds);
}

// ---------------------------------------------------------------------------
// Host side of the flex form: the share plan of one set of layer geometries,
// built once and cached with its device table.
// ---------------------------------------------------------------------------
namespace {

struct FlexPlan {
  bool ok = false;
  int S = 4;       // team size
  int nteams = 0;
  int nwg = 0;     // grid
  std::vector<int> host;  // items [nwg][kFlexMaxItems][kFlexItemInts], then per-layer tile tables
  int* dev = nullptr;
  int nslots[kMaxBatch] = {};
  int tile_ofs[kMaxBatch] = {};  // ints from the start of the table
  int zc[kMaxBatch] = {};
  int cw[kMaxBatch] = {};        // members that share the bias column sums of a layer
  int nitems = 0;                // (team, column) pairs x live members
};

// 0: never, 1: when every team gets a worthwhile share (default), 2: whenever the
// launch has the ring form (tests: small shapes)
std::atomic<int> g_flex_mode{-1};
int flex_mode() {
  int m = g_flex_mode.load();
  if (m < 0) {
    const char* e = getenv("CALCIUMGAN_WGRAD_FLEX");
    m = e ? atoi(e) : 1;
    if (m < 0 || m > 2) m = 1;
    g_flex_mode.store(m);
  }
  return m;
}
int flex_env(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

struct FlexCol {
  int layer, cw, ch, bx0, by0;
  long long n;     // K' tiles
  long long cost;  // per K' tile, in team time
};

// plans[i]: the standard plan of layer i (geometry only is read)
void plan_flex(const WgradPlan* plans, int n, int mode, FlexPlan& fp) {
  fp.ok = false;
  // cost of a K' tile in team time, from the in-kernel trace of the cfg2 critic
  // pass (tools/wgrad_trace.py, profiles/r05_wgrad_flex_trace.txt: K' loop cycles
  // per tile 4141 / 4361 / 4324 / 4274 for the four layers with 128-row tiles,
  // 2517 for the 64-row tiles of the last): 64-row tiles pay the per-tile barrier
  // and the DMA issue twice per 128 rows (0.59, not 0.5); the layer without a
  // PhaseShuffle in front of it, whose tiles are mostly interior, runs 3-5 % under
  // the others
  const int c128 = flex_env("CALCIUMGAN_WGRAD_FLEX_C128", 64);
  const int c64 = flex_env("CALCIUMGAN_WGRAD_FLEX_C64", 38);
  const int cplain = flex_env("CALCIUMGAN_WGRAD_FLEX_CPLAIN", 62);
  const int snap = flex_env("CALCIUMGAN_WGRAD_FLEX_SNAP", 4);
  const long long min_share = 8 * c128;  // a team's share must be worth its set-up
  // team size: the largest whose padded columns cost at most 5 % over the best
  int bestS = 0;
  double ts[5] = {0, 0, 0, 0, 0};
  std::vector<FlexCol> cols_of[5];
  const int force_s = flex_env("CALCIUMGAN_WGRAD_FLEX_TEAM", 0);
  for (int S = 4; S >= 1; S >>= 1) {
    if (force_s && S != force_s) continue;
    std::vector<FlexCol>& cols = cols_of[S];
    long long total = 0;
    for (int i = 0; i < n; ++i) {
      const WgradPlan& p = plans[i];
      // column shape: fewest columns (least padding), then least re-read bytes
      const double xb = (double)p.a.nB * p.a.Lx * p.a.Cx, gb = (double)p.a.M * p.a.Cg;
      int bcw = 1, bch = 1;
      long long bcols = -1;
      double bbytes = 0;
      for (int cw = 1; cw <= S; cw <<= 1) {
        const int ch = S / cw;
        const int ncx = (p.gx + cw - 1) / cw, ncy = (p.gy + ch - 1) / ch;
        const long long nc = (long long)ncx * ncy;
        const double bytes = ncy * xb + ncx * gb;
        if (bcols < 0 || nc < bcols || (nc == bcols && bytes < bbytes)) {
          bcols = nc; bbytes = bytes; bcw = cw; bch = ch;
        }
      }
      const long long cost = p.TT == 128 ? (p.a.shifts ? c128 : cplain) : c64;
      for (int by0 = 0; by0 < p.gy; by0 += bch)
        for (int bx0 = 0; bx0 < p.gx; bx0 += bcw) {
          cols.push_back(FlexCol{i, bcw, bch, bx0, by0, (long long)p.a.ntiles, cost});
          total += (long long)p.a.ntiles * cost;
        }
    }
    ts[S] = (double)total * S;  // / 256 workgroups
  }
  {
    double tmin = 1e300;
    for (int S = 1; S <= 4; S <<= 1)
      if (ts[S] > 0 && ts[S] < tmin) tmin = ts[S];
    for (int S = 4; S >= 1; S >>= 1)
      if (ts[S] > 0 && ts[S] <= 1.05 * tmin) { bestS = S; break; }
  }
  if (!bestS) return;
  const int S = bestS;
  const std::vector<FlexCol>& cols = cols_of[S];
  long long total = 0;
  for (const FlexCol& c : cols) total += c.n * c.cost;
  const int max_teams = 256 / S;
  int nteams = max_teams;
  if (total < (long long)nteams * min_share) {
    if (mode < 2) return;  // too little work: the split forms are the better fit
    nteams = (int)(total / min_share);
    if (nteams < 1) nteams = 1;
  }
  // cut t = (column, K' tile); tiny heads / tails of a column are snapped away
  struct Cut { int col; long long k; };
  std::vector<Cut> cuts(nteams + 1);
  {
    size_t ci = 0;
    long long start = 0;  // cost at the start of column ci
    for (int t = 0; t <= nteams; ++t) {
      const long long b = t == nteams ? total : (long long)((__int128)total * t / nteams);
      while (ci < cols.size() && start + cols[ci].n * cols[ci].cost <= b) {
        start += cols[ci].n * cols[ci].cost;
        ++ci;
      }
      if (ci >= cols.size()) { cuts[t] = Cut{(int)cols.size(), 0}; continue; }
      long long k = (b - start) / cols[ci].cost;
      // (snapping moves a cut by at most ~2 % of a share)
      long long sn = total / nteams / (50 * cols[ci].cost);
      sn = sn < 1 ? 1 : (sn > snap ? snap : sn);
      if (k < sn) k = 0;
      if (cols[ci].n - k < sn) { cuts[t] = Cut{(int)ci + 1, 0}; continue; }
      cuts[t] = Cut{(int)ci, k};
    }
    cuts[0] = Cut{0, 0};
    cuts[nteams] = Cut{(int)cols.size(), 0};
  }
  const int nwg = (nteams + 7) / 8 * 8 * S;
  fp.S = S; fp.nteams = nteams; fp.nwg = nwg;
  // members that share the bias column sums: the largest power of two of LIVE cx
  // blocks in the column of cx block 0 (they stream the same g tiles)
  for (int i = 0; i < n; ++i) fp.cw[i] = 1;
  for (const FlexCol& c : cols) {
    if (c.bx0 != 0) continue;
    const int live = c.cw < plans[c.layer].gx ? c.cw : plans[c.layer].gx;
    int b = 1;
    while (b * 2 <= live) b *= 2;
    fp.cw[c.layer] = b;
  }
  const size_t items_ints = (size_t)nwg * kFlexMaxItems * kFlexItemInts;
  size_t tiles_ints = 0;
  for (int i = 0; i < n; ++i) {
    fp.tile_ofs[i] = (int)(items_ints + tiles_ints);
    tiles_ints += (size_t)plans[i].gx * plans[i].gy * 2;
  }
  fp.host.assign(items_ints + tiles_ints, 0);
  for (int w = 0; w < nwg; ++w)
    for (int k = 0; k < kFlexMaxItems; ++k)
      fp.host[((size_t)w * kFlexMaxItems + k) * kFlexItemInts] = -1;
  // pass 1: count the items of every output tile; pass 2: slots + table rows
  std::vector<int> wg_items(nwg, 0);
  for (int pass = 0; pass < 2; ++pass) {
    if (pass == 1) {
      for (int i = 0; i < n; ++i) {
        int* tl = fp.host.data() + fp.tile_ofs[i];
        int run = 0, most = 0;
        for (int t = 0; t < plans[i].gx * plans[i].gy; ++t) {
          const int cnt = tl[2 * t + 1];
          tl[2 * t] = run;
          tl[2 * t + 1] = 0;  // refilled below, in K' order
          run += cnt;
          if (cnt > most) most = cnt;
        }
        fp.nslots[i] = run;
        fp.zc[i] = 1;
        while (fp.zc[i] < 8 && (most + fp.zc[i] - 1) / fp.zc[i] > 8) fp.zc[i] *= 2;
      }
    }
    for (int t = 0; t < nteams; ++t) {
      const Cut a = cuts[t], b = cuts[t + 1];
      for (int ci = a.col; ci <= b.col && ci < (int)cols.size(); ++ci) {
        const FlexCol& c = cols[ci];
        const long long k0 = ci == a.col ? a.k : 0;
        const long long k1 = ci == b.col ? b.k : c.n;
        if (k1 <= k0) continue;
        const WgradPlan& p = plans[c.layer];
        int* tl = fp.host.data() + fp.tile_ofs[c.layer];
        for (int m = 0; m < c.cw * c.ch; ++m) {
          const int bx = c.bx0 + m % c.cw, by = c.by0 + m / c.cw;
          if (bx >= p.gx || by >= p.gy) continue;
          const int tile = by * p.gx + bx;
          if (pass == 0) { ++tl[2 * tile + 1]; continue; }
          const int wg = ((t >> 3) * S + m) * 8 + (t & 7);
          const int k = wg_items[wg]++;
          if (k >= kFlexMaxItems) return;  // (fp.ok stays false)
          int* row = fp.host.data() + ((size_t)wg * kFlexMaxItems + k) * kFlexItemInts;
          row[0] = c.layer; row[1] = bx; row[2] = by;
          row[3] = (int)k0; row[4] = (int)(k1 - k0);
          row[5] = tl[2 * tile] + tl[2 * tile + 1]++;
          ++fp.nitems;
        }
      }
    }
  }
  fp.ok = true;
}

std::mutex g_flex_mutex;
std::map<std::vector<long long>, FlexPlan*> g_flex_cache;

// the cached plan of this geometry set, its table uploaded on `s` when new
FlexPlan* flex_plan_for(const WgradPlan* plans, int n, int mode, hipStream_t s) {
  std::vector<long long> key;
  key.push_back(cg_device_index());
  key.push_back(mode);
  for (int i = 0; i < n; ++i) {
    const WgradPlan& p = plans[i];
    for (long long v : {(long long)p.gx, (long long)p.gy, (long long)p.TT, (long long)p.a.ntiles,
                        (long long)p.a.nB, (long long)p.a.Lx, (long long)p.a.Cx, (long long)p.a.M,
                        (long long)p.a.Cg})
      key.push_back(v);
  }
  std::lock_guard<std::mutex> lock(g_flex_mutex);
  auto f = g_flex_cache.find(key);
  if (f != g_flex_cache.end()) return f->second;
  FlexPlan* fp = new FlexPlan;  // (lives as long as the process: graphs keep its table)
  plan_flex(plans, n, mode, *fp);
  if (fp->ok) {
    const size_t bytes = fp->host.size() * sizeof(int);
    // (a first call inside a stream capture: the allocation is not a captured
    // operation, the copy becomes a node that re-sends the same bytes)
    hipStreamCaptureMode cm = hipStreamCaptureModeRelaxed;
    (void)hipThreadExchangeStreamCaptureMode(&cm);
    const hipError_t e = hipMalloc(reinterpret_cast<void**>(&fp->dev), bytes);
    (void)hipThreadExchangeStreamCaptureMode(&cm);
    if (e != hipSuccess ||
        hipMemcpyAsync(fp->dev, fp->host.data(), bytes, hipMemcpyHostToDevice, s) != hipSuccess) {
      (void)hipGetLastError();
      fp->ok = false;
    }
    // (the table is shared by every later launch of this geometry, on whatever
    // stream: outside a capture, wait for the copy once)
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (fp->ok && hipStreamIsCapturing(s, &cs) == hipSuccess &&
        cs == hipStreamCaptureStatusNone)
      (void)hipStreamSynchronize(s);
    if (fp->ok && flex_env("CALCIUMGAN_WGRAD_FLEX_PRINT", 0)) {
      fprintf(stderr, "cg_wgrad_batched flex: %d layers, teams of %d x %d, %d workgroups, %d items, slots",
              n, fp->S, fp->nteams, fp->nwg, fp->nitems);
      for (int i = 0; i < n; ++i) fprintf(stderr, " %d", fp->nslots[i]);
      fprintf(stderr, "\n");
    }
  }
  g_flex_cache[key] = fp;
  return fp;
}

template <int TPW, int ALLT>
int launch_flex(const WgradFlex& m, int blocks, size_t lds, hipStream_t s) {
  static CgPerDeviceFlag attr_set;
  if (!attr_set.test()) {
    hipError_t e = hipFuncSetAttribute(
        reinterpret_cast<const void*>(&wgrad_flex_kernel<2, TPW, ALLT>),
        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set.mark();
  }
  CG_LAUNCH_PROF(CG_FAMILY_WGRAD, (wgrad_flex_kernel<2, TPW, ALLT>), dim3(blocks),
                 dim3(512), lds, s, m);
  CG_LAUNCH_CHECK();
}

// 0: done; > 0: HIP error; < 0: not this form (the caller takes the split forms)
int run_flex(const cg_wgrad_desc* descs, int n, WgradPlan* plans, hipStream_t s) {
  const int mode = flex_mode();
  if (!mode || n > kMaxBatch) return -1;
  {
    // (the share plan fills 256 workgroups, one per CU, 32 per XCD: MI355X)
    static int cus[kCgMaxDevices];
    int& c = cus[cg_device_index()];
    if (!c && hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount,
                                    cg_device_index()) != hipSuccess)
      c = 256;
    if (c != 256) return -1;
  }
  for (int i = 0; i < n; ++i) {
    const WgradPlan& p = plans[i];
    if (!p.a.ring || p.tpw != 3 || !descs[i].partials) return -1;
  }
  FlexPlan* fp = flex_plan_for(plans, n, mode, s);
  if (!fp->ok) return -1;
  WgradFlex m;
  FlexReduceArgs ra;
  m.n = n;
  m.table = fp->dev;
  ra.n = n;
  size_t lds = 0;
  long long most = 0;
  for (int i = 0; i < n; ++i) {
    const WgradPlan& p = plans[i];
    const long long per_slot = (long long)p.tpw * 8 * 2048;
    const long long dw_part = (long long)fp->nslots[i] * per_slot;
    const long long bias = descs[i].dbias ? (long long)fp->nslots[i] * 64 : 0;
    if (descs[i].partials_elems < dw_part + bias) return -1;
    m.a[i] = p.a;
    m.a[i].part = descs[i].partials;
    m.a[i].bias_part = bias ? descs[i].partials + dw_part : nullptr;
    m.a[i].direct_store = 0;
    m.tt[i] = p.TT;
    m.cw[i] = fp->cw[i];
    if (2 * p.lds > lds) lds = 2 * p.lds;
    FlexReduceItem& it = ra.it[i];
    it.cw = fp->cw[i];
    it.part = m.a[i].part; it.dw = p.a.dw;
    it.tiles = fp->dev + fp->tile_ofs[i];
    it.gx = p.gx; it.gy = p.gy; it.tpw = p.tpw; it.taps = p.a.taps;
    it.Cx_real = p.a.Cx_real; it.Cg_real = p.a.Cg_real;
    it.store = descs[i].store ? 1 : 0;
    it.zc = fp->zc[i];
    it.bias_part = m.a[i].bias_part; it.dbias = p.a.dbias;
    const long long t = (long long)p.gx * p.gy * p.tpw * 8 * 512 * it.zc;
    if (t > most) most = t;
  }
  int rc = launch_flex<3, 2>(m, fp->nwg, lds, s);
  if (rc) return rc > 0 ? rc : 1;
  long long bx = (most + 255) / 256;
  if (bx > 2048) bx = 2048;
  CG_LAUNCH_PROF(CG_FAMILY_WGRAD, wgrad_flex_reduce_kernel, dim3((unsigned)bx, n), dim3(256),
                 0, s, ra);
  return (int)hipGetLastError();
}

}  // namespace

// The share plan of a batched launch, for inspection (host only, no device call):
// the table wgrad_flex_kernel would read -- items [nwg][8][6] = (layer | -1, bx,
// by, first K' tile, tiles, slot), then per layer [gx * gy][2] = (first slot,
// slots) -- is copied to `out` (when it holds the returned count of ints) and
// info[0..3] = team size, teams, workgroups, items, info[4 + i] = slots of layer
// i, info[10 + i] = int offset of layer i's tile table.  < 0: this launch does not
// take the flex form under `mode` (1 | 2).
extern "C" long long cg_wgrad_flex_plan(const cg_wgrad_desc* descs, int n, int mode,
                                        int* out, long long out_ints, int* info) {
  if (!descs || n < 2 || n > kMaxBatch || mode < 1 || mode > 2) return -1;
  WgradPlan plans[kMaxBatch];
  for (int i = 0; i < n; ++i) {
    if (plan_wgrad(descs + i, plans[i])) return -1;
    const WgradPlan& p = plans[i];
    if (p.rowsplit || !p.pipe || p.a.nseg != 1 || !p.a.ring || p.tpw != 3) return -1;
  }
  FlexPlan fp;
  plan_flex(plans, n, mode, fp);
  if (!fp.ok) return -1;
  if (info) {
    info[0] = fp.S; info[1] = fp.nteams; info[2] = fp.nwg; info[3] = fp.nitems;
    for (int i = 0; i < kMaxBatch; ++i) {
      info[4 + i] = i < n ? fp.nslots[i] : 0;
      info[10 + i] = i < n ? fp.tile_ofs[i] : 0;
    }
  }
  if (out && out_ints >= (long long)fp.host.size())
    memcpy(out, fp.host.data(), fp.host.size() * sizeof(int));
  return (long long)fp.host.size();
}

extern "C" int cg_debug_wgrad_flex(int mode) {
  const int was = flex_mode();
  if (mode >= 0 && mode <= 2) g_flex_mode.store(mode);
  return was;
}

extern "C" int cg_wgrad_batched(const cg_wgrad_desc* descs, int n, void* stream) {
  if (!descs || n < 1) return CG_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  WgradPlan plans[kMaxBatch];
  bool fuse = n >= 2 && n <= kMaxBatch;
  for (int i = 0; i < n && i < kMaxBatch; ++i) {
    const int rc = plan_wgrad(descs + i, plans[i]);
    if (rc) return rc;
    const WgradPlan& p = plans[i];
    // the fused kernel is the pipelined stride-2 instantiation with every tap
    // live; anything else is launched on its own
    if (p.rowsplit || !p.pipe || p.a.taps != 8 * p.tpw || p.a.nseg != 1 ||
        p.tpw != plans[0].tpw || p.a.ring != plans[0].a.ring)
      fuse = false;
  }
  if (!fuse) {
    for (int i = 0; i < n; ++i) {
      const int rc = cg_wgrad(descs + i, stream);
      if (rc) return rc;
    }
    return 0;
  }
  {
    const int rc = run_flex(descs, n, plans, s);
    if (rc >= 0) return rc;
  }
  WgradMulti m;
  int blocks = 0;
  size_t lds = 0;
  if (!plan_halves(descs, n, plans, m, blocks, lds)) {
    m.n = n;
    for (int i = 0; i < n; ++i) {
      const WgradPlan& p = plans[i];
      m.a[i] = p.a;
      m.gx[i] = p.gx; m.gy[i] = p.gy; m.gz[i] = p.nsplit; m.tt[i] = p.TT;
      const int nb = (int)wgrad_grid(p.a, p.nsplit);
      if (nb > blocks) blocks = nb;
      if (2 * p.lds > lds) lds = 2 * p.lds;
    }
    for (int i = 0; i < n; ++i) {
      m.lo[i] = 0; m.cnt[i] = blocks; m.zofs[i] = 0; m.zcnt[i] = m.gz[i];
    }
  }
  int rc;
  switch (plans[0].tpw) {
    case 1: rc = launch_multi<1>(m, blocks, lds, s); break;
    case 2: rc = launch_multi<2>(m, blocks, lds, s); break;
    default:
      rc = plans[0].a.ring ? launch_multi<3, 2>(m, blocks, lds, s)
                           : launch_multi<3>(m, blocks, lds, s);
      break;
  }
  if (rc) return rc;
  ReduceArgs ra;
  ra.n = 0;
  for (int i = 0; i < n; ++i)
    if (plans[i].a.part) reduce_item(plans[i], ra.it[ra.n++]);
  return ra.n ? launch_reduce(ra, s) : 0;
}

#ifdef CG_WGRAD_TRACE
extern "C" int cg_debug_wgrad_trace(unsigned* dst, int n) {
  if (n > 256 * 8 * kWTraceParts) return CG_EINVAL;
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_wgrad_trace),
                                  (size_t)n * sizeof(unsigned));
}
#endif
