// Sliding-window convolution as an implicit GEMM on gfx950 MFMA.
//
// One kernel family serves every dense contraction of the CalciumGAN hot path
// (reference call sites in include/calciumgan_hip.h):
//   * Conv1D forward (stride 2, k taps)            -> R = 2
//   * Conv1D input-gradient / Conv1DTranspose fwd  -> R = 1, k/2 taps, 2 phases
//   * Conv1DTranspose input-gradient (stride 2)    -> R = 2
//   * per-timestep Dense (taps = 1)                -> R = 1
//
// Data layout / tiling (DESIGN.md "swconv"):
//   * activations are channels-last, so output row u needs ONE contiguous
//     window of `taps` source rows; a tile of TM consecutive rows shares a
//     window of (R*TM + taps - R) rows -> each source byte is staged into LDS
//     once per tile and reused by up to taps/R output rows (no im2col).
//   * stride-2 windows are de-interleaved by row parity while staging, so every
//     tap becomes a stride-1 walk over one parity region: MFMA A fragments are
//     single 16-byte ds_read_b128 with a 2*odd 16-B-slot row pitch (bank-conflict
//     free), and the channel chunk CK bounds LDS use.
//   * the weight operand is pre-packed (cg_pack_weights) in exactly the K order
//     the kernel walks, so a B stage is a contiguous 256-B-per-row copy.
//   * block = 256 threads = 4 waves, 4x1 or 2x2 over (M, N); wave tile
//     (MF*MT) x 64 on v_mfma_f32_16x16x32_bf16 (MF = 16) or
//     v_mfma_f32_32x32x16_bf16 (MF = 32), fp32 accumulate; the 2x2 layout
//     shares one source window between two 64-column halves.  The epilogue
//     goes through LDS so global stores are 16-byte and row-contiguous, with
//     bias / LeakyReLU / LeakyReLU-derivative mask / sigmoid fused.
#include <type_traits>
#include <vector>

#include "cg_common.h"
#include "swconv_args.h"

namespace {


// LDS row pitches, 16x16x32: 2*odd 16-byte slots: with the MFMA operand map
// (lane -> row l&15, k-group l>>4) and ds_read_b128's lane groups
// {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32), the 8 lanes of k-group g land on
// the even slots and the 8 lanes of k-group g+1 (next 16-B chunk) on the odd
// slots: no bank conflicts.  (An odd pitch gives 2-way conflicts there.)
// 32x32x16 (lane -> row l&31, k-group l>>5): the 16 lanes of a group share one
// k-group and cover 16 rows that are distinct mod 16, so the pitch is an ODD
// number of slots (row r -> slot r*odd mod 16, a permutation).
// Weight stages live in a 3-deep LDS ring filled by LDS-DMA
// (global_load_lds_dwordx4: lane-linear 1 KiB per wave-instruction, so rows are
// unpadded) and XOR-swizzled per 16-byte chunk instead: chunk c of row n sits at
// slot c ^ swz(n), swz = (n >> 1) & 7 for 128-B rows, n & 15 for 256-B rows ->
// the 16 rows of a fragment read hit 16 distinct bank slots, and the two
// k-groups of a ds_read_b128 lane group stay disjoint.
constexpr int kNBufB = 3;
constexpr int kScrPitch = 68;         // fp32 epilogue scratch pitch
constexpr int ldsB_bytes(int ks, int tn) {  // ring of tn x (32*ks) bf16 stages
  return kNBufB * tn * (32 * ks) * 2;
}
constexpr int kScratchBytes = 4 * 16 * kScrPitch * 4;

// UNI: 4 | c8, the K walk is wave-uniform (see the chunk loop).
// R: source stride.  MF: MFMA rows (16: 16x16x32, 32: 32x32x16).  WGN: waves
// along N (1: 4x1 waves, 64-column tiles; 2: 2x2 waves, 128-column tiles that
// share one source window).  MT: MF-row subtiles per wave (tile =
// (4/WGN)*MT*MF rows).  KS: MFMA K-steps per weight stage (4: 128-deep stages;
// 2: 64-deep stages, half the LDS so stride-2 windows still fit two
// workgroups per CU).  (The 128 x 64 wave tile holds 128 accumulator
// registers: the second launch-bound argument keeps the rest within 128 so
// two workgroups share a CU.)
// (The PIPE instantiation -- 256-row 16x16x32 tile, two-K-step stages -- sits
// at the edge of the three-waves-per-SIMD register budget: pin it there.)
// SP (stride 2, parity-major weights): the window of a channel chunk is staged
// one source-row parity at a time -- half the LDS, so a 256-row stride-2 tile
// leaves room for a third workgroup on the CU -- at twice the staging phases.
template <int R, int MF, int WGN, int MT, int KS, bool UNI, bool SP = false,
          bool LN = false>
__global__ __launch_bounds__(256, ((MF == 32 && MT == 4) || MT == 8)    ? 2
                                  : (MF == 16 && MT == 4 && KS == 2 && UNI && WGN == 1) ? 3
                                                                     : 1) void
swconv_kernel(ConvArgs a) {
  static_assert(!SP || (R == 2 && UNI), "split-parity staging is a stride-2 mode");
  static_assert(!LN || WGN == 2, "the fused LayerNorm needs a 128-column tile");
  constexpr int NREG = SP ? 1 : R;  // parity regions resident in LDS at a time
  constexpr int NPART = SP ? 2 : 1; // staging parts per channel chunk
  // PIPE: all fragment reads of a weight stage are issued ahead of its MFMAs
  // (see the stage loop); measured 2-5 % faster for the 256-row 16x16x32
  // tile with two-K-step stages, where the 16 fragments fit the register
  // budget of three waves per SIMD; mixed for 128-row tiles and slower for
  // the 32x32x16 ones, which keep the compiler's own order
  constexpr bool PIPE = MF == 16 && MT == 4 && KS == 2;
  static_assert(MF == 16 || (MF == 32 && UNI), "32x32x16 needs the uniform K walk");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int WGM = 4 / WGN;
  constexpr int TM = WGM * MT * MF;
  constexpr int TN = 64 * WGN;
  constexpr int NT = 64 / MF;       // MF-column subtiles per wave (4 or 2)
  constexpr int KH = MF / 16;       // MFMAs per 32-deep K-step (1 or 2)
  constexpr int FS = 4 * KS;        // 16-byte K groups per stage (16 or 8)
  constexpr int kRowB = FS * 8;     // bf16 elements per B row in LDS (no pad)
  constexpr int kBufB = TN * kRowB; // elements per ring slot
  constexpr int NDMA = (kBufB * 2 / 1024) / 4;  // DMA instructions per wave/stage
  using acc_t = typename std::conditional<MF == 16, f32x4, f32x16>::type;
  uint16_t* ldsA = reinterpret_cast<uint16_t*>(smem);
  uint16_t* ldsB = ldsA + a.ldsA_elems;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WGN;        // wave position along M / N
  const int wn = wave % WGN;
  const int rM = lane & (MF - 1);   // operand row / column of this lane
  const int g = lane / MF;          // its 16-byte k-group inside one MFMA
  // XCD-aware mapping: the gn*gp workgroups that share one row tile's source
  // window get linear ids congruent mod 8 (same XCD / L2) and adjacent in
  // dispatch order; speed only, results do not depend on placement.
  const int lin = blockIdx.x;
  const int xcd = lin & 7;
  const int jq = lin >> 3;
  const int gnp = a.gn * a.gp;
  const int np_i = jq % gnp;
  const int bm = (jq / gnp) * 8 + xcd;
  if (bm >= a.gm) return;
  const int bn = np_i % a.gn;
  const int phase = np_i / a.gn;
  const uint16_t* __restrict__ wp = a.w + (long long)phase * a.w_phase_stride;
  if (a.ksplit > 1) {
    // split-K: this workgroup owns channel chunks [z * nchunks, (z+1) * nchunks)
    // (a.nchunks is the per-split count) and writes f32 partial sums to its
    // own slice of the workspace; everything below sees a shorter K walk
    const int z = blockIdx.y;
    a.x += z * a.nchunks * a.CK;
    wp += (long long)z * a.nchunks * a.Fp * 8;
    a.y = reinterpret_cast<float*>(a.y) + z * a.split_stride;
    if (z != a.ksplit - 1) a.narrow = 0;
  }
  const int off = a.off + phase * a.off_phase_step;
  const int y_off = a.y_off + phase * a.yoff_phase_step;
  const int m0 = bm * TM;
  const int n0 = bn * TN;
  const int regionRows = a.nseg * a.WR;

  int rowbase[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int i = (wm * MT + mt) * MF + rM;
    const int seg = i >> a.log2S;
    const int ui = i & (a.S - 1);
    rowbase[mt] = (seg * a.WR + ui) * a.pitchA;
  }

  acc_t acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < MF * MF / 64; ++r) acc[mt][nt][r] = 0.f;

  const int totalA = NREG * regionRows * a.c8;
  // B staging by LDS-DMA: wave w issues instructions j = w*NDMA + i, each
  // writing LDS bytes [j KiB, (j+1) KiB) of the ring slot; lane L lands at byte
  // j*1024 + L*16 = (row, slot c'), and fetches global chunk c = c' ^ swz(row).
  const uint16_t* dsrc[NDMA];
#pragma unroll
  for (int i = 0; i < NDMA; ++i) {
    const int p = ((wave * NDMA + i) * 1024 + lane * 16) / 2;  // element offset
    const int row = p / kRowB;
    const int cs = (p % kRowB) / 8;
    const int c = cs ^ (KS == 4 ? (row & 15) : ((row >> 1) & 7));
    dsrc[i] = wp + (long long)(n0 + row) * a.Kpack + c * 8;
  }
  const int nstages = a.Fp / FS;
  // narrow last chunk: 32 K groups (16 per tap parity) instead of taps * 4
  const int total_stages =
      a.narrow ? (a.nchunks - 1) * nstages + 32 / FS : a.nchunks * nstages;
  auto issue_dma = [&](int gs) {
    // packed weights are contiguous over (chunk, stage): offset gs * FS * 8
    uint16_t* slot = ldsB + (gs % kNBufB) * kBufB;
#pragma unroll
    for (int i = 0; i < NDMA; ++i)
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(dsrc[i] +
                                                          (long long)gs * FS * 8),
          (__attribute__((address_space(3))) void*)(slot +
                                                    (wave * NDMA + i) * 512),
          16, 0, 0);
  };
  // fragment-read offsets of this lane inside a ring slot (swizzled chunk);
  // MFMA kh of K-step ks reads k-group 4*ks + 2*kh + g
  const int swz = KS == 4 ? (rM & 15) : ((rM >> 1) & 7);
  int boff[KS][KH];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int kh = 0; kh < KH; ++kh)
      boff[ks][kh] =
          (wn * 64 + rM) * kRowB + (((4 * ks + 2 * kh + g) ^ swz) * 8);
  issue_dma(0);
#pragma unroll
  for (int i = 1; i < kNBufB - 1; ++i)
    if (i < total_stages) issue_dma(i);

  const int nst_part = nstages / NPART;  // weight stages per staged part
  const int half_taps = a.taps >> 1;
  int tap = 0;              // position in the packed tap order
  int q8l = UNI ? 0 : g;    // 16-byte group inside the tap (c8 >= 4)
  for (int ccp = 0; ccp < a.nchunks * NPART; ++ccp) {
    const int cc = ccp / NPART;
    const int part = ccp % NPART;  // SP: the source-row parity staged now
    if (part == 0) {
      tap = 0;
      q8l = UNI ? 0 : g;
    }
    // previous part's fragment reads are done (LDS only: a plain
    // __syncthreads() would also wait vmcnt(0) and drain the weight DMAs that
    // are in flight across this staging phase)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // ---- stage the source window of this channel chunk ------------------
    if (a.nseg == 1) {
      // fast path (per-sample length >= tile): one sample per tile, no
      // per-piece integer division (row = idx / c8 through an exact float
      // reciprocal; everything else is wave-uniform)
      const int b = m0 / a.Lu;
      const int u0 = m0 - b * a.Lu;
      const bool valid = m0 < a.M;
      const int sft = (a.shifts && valid) ? a.shifts[b / a.seg_size] : 0;
      const uint16_t* xb = a.x + (long long)b * a.Lx * a.Cx + cc * a.CK;
      const int srow0 = R * u0 + off;
      if (a.log2c8 >= 0) {
        // c8 is a power of two: a thread keeps its 16-byte column and walks
        // rows with a constant step -> no per-piece index arithmetic
        const int q8 = tid & (a.c8 - 1);
        const int rstep = 256 >> a.log2c8;
        const uint16_t* xq = xb + q8 * 8;
        uint16_t* dst = ldsA + (tid >> a.log2c8) * a.pitchA + q8 * 8;
        const int dstep = rstep * a.pitchA;
        for (int row = tid >> a.log2c8; row < NREG * a.WR; row += rstep) {
          const int rho = SP ? part : ((R == 2 && row >= a.WR) ? 1 : 0);
          int srow = srow0 + R * (SP ? row : row - rho * a.WR) + rho;
          uint4 v = make_uint4(0u, 0u, 0u, 0u);
          if (valid && srow >= 0 && srow < a.Lx) {
            if (a.shifts) srow = shuffle_src(srow, sft, a.Lx);
            v = *reinterpret_cast<const uint4*>(xq + (long long)srow * a.Cx);
          }
          *reinterpret_cast<uint4*>(dst) = v;
          dst += dstep;
        }
      } else {
        for (int idx = tid; idx < totalA; idx += 256) {
          const int row = __float2int_rz(((float)idx + 0.5f) * a.inv_c8);
          const int q8 = idx - row * a.c8;
          const int rho = SP ? part : ((R == 2 && row >= a.WR) ? 1 : 0);
          const int wr = SP ? row : row - rho * a.WR;
          int srow = srow0 + R * wr + rho;
          uint4 v = make_uint4(0u, 0u, 0u, 0u);
          if (valid && srow >= 0 && srow < a.Lx) {
            if (a.shifts) srow = shuffle_src(srow, sft, a.Lx);
            v = *reinterpret_cast<const uint4*>(xb + (long long)srow * a.Cx +
                                                q8 * 8);
          }
          *reinterpret_cast<uint4*>(ldsA + row * a.pitchA + q8 * 8) = v;
        }
      }
    } else if (a.log2c8 >= 0) {
      // several whole samples per tile (per-sample length < tile, so S == Lu
      // and every segment starts at row 0 of its sample).  A thread keeps its
      // 16-byte column; its window row advances by a constant step, carried
      // into (sample, parity region) incrementally -- no per-piece divisions
      const int q8 = tid & (a.c8 - 1);
      const int rstep = 256 >> a.log2c8;
      int row = tid >> a.log2c8;
      int rho = SP ? 0 : row / regionRows;
      int seg = (row - rho * regionRows) / a.WR;
      int wr = row - rho * regionRows - seg * a.WR;
      if (SP) rho = part;
      const int b0 = m0 / a.Lu;
      const uint16_t* xq = a.x + cc * a.CK + q8 * 8;
      uint16_t* dst = ldsA + row * a.pitchA + q8 * 8;
      const int dstep = rstep * a.pitchA;
      for (; row < NREG * regionRows; row += rstep) {
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        const int b = b0 + seg;
        int srow = off + R * wr + rho;
        if (b < a.nB && srow >= 0 && srow < a.Lx) {
          if (a.shifts) srow = shuffle_src(srow, a.shifts[b / a.seg_size], a.Lx);
          v = *reinterpret_cast<const uint4*>(
              xq + ((long long)b * a.Lx + srow) * a.Cx);
        }
        *reinterpret_cast<uint4*>(dst) = v;
        dst += dstep;
        wr += rstep;
        while (wr >= a.WR) {
          wr -= a.WR;
          ++seg;
        }
        if (!SP && seg >= a.nseg) {
          seg -= a.nseg;
          ++rho;
        }
      }
    } else {
      for (int idx = tid; idx < totalA; idx += 256) {
        const int row = idx / a.c8;
        const int q8 = idx - row * a.c8;
        const int rho = row / regionRows;
        const int rem = row - rho * regionRows;
        const int seg = rem / a.WR;
        const int wr = rem - seg * a.WR;
        const int mseg = m0 + seg * a.S;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (mseg < a.M) {
          const int b = mseg / a.Lu;
          const int u0 = mseg - b * a.Lu;
          int srow = R * u0 + off + R * wr + rho;
          if (srow >= 0 && srow < a.Lx) {
            if (a.shifts)
              srow = shuffle_src(srow, a.shifts[b / a.seg_size], a.Lx);
            const uint16_t* src =
                a.x + ((long long)b * a.Lx + srow) * a.Cx + cc * a.CK + q8 * 8;
            v = *reinterpret_cast<const uint4*>(src);
          }
        }
        *reinterpret_cast<uint4*>(ldsA + row * a.pitchA + q8 * 8) = v;
      }
    }
    // (no barrier here: the first weight stage below waits for this wave's
    // window stores -- lgkmcnt(0) -- right before its own s_barrier)

    // Flattened K position inside the chunk: 16-byte group f = 4*kstep + g.
    // When 4 | c8 the four k-groups of a K-step share one tap and the walk is
    // wave-uniform (scalar registers, lane part g*8 folded into the row base);
    // otherwise (e.g. c8 = 13) every lane tracks its own (tap, group).
    {
      const bool narrow_now = R == 2 && UNI && a.narrow && cc == a.nchunks - 1;
      const int goff = (UNI && !narrow_now) ? g * 8 : 0;
      const int nsp = narrow_now ? (32 / FS) / NPART : nst_part;
      int npos = part * 4;  // narrow walk: K-step counter (4 per tap parity)
      for (int s = part * nsp; s < (part + 1) * nsp; ++s) {
        const int gs = cc * nstages + s;
        // stage gs has landed once all but this wave's newest (ring depth - 2)
        // stages of DMAs are done (only DMA(gs+1 ..) may stay in flight); the
        // barrier then (a) makes every wave's part of stage gs visible and (b)
        // guarantees the slot of stage gs-1 is no longer being read before it
        // is refilled with stage gs + depth - 1
        const int ahead = total_stages - 1 - gs;  // later stages, issued or not
        if (ahead >= kNBufB - 2)
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"((kNBufB - 2) * NDMA) : "memory");
        else if (kNBufB > 3 && ahead == 1)
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
        else
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // window stores
        __builtin_amdgcn_s_barrier();
        if (gs + kNBufB - 1 < total_stages) issue_dma(gs + kNBufB - 1);
        const uint16_t* curB = ldsB + (gs % kNBufB) * kBufB;
        // PIPE instantiations: the fragments of K-step ks+1 are read into a
        // second register set before the MFMAs of K-step ks issue, and a
        // scheduling barrier keeps them there (left alone, hipcc sinks every
        // read to just before its use and waits on it with lgkmcnt(0): ~one
        // exposed LDS round trip per four MFMAs).  The stage then runs as
        // "all reads, one wait, 16*KS MFMAs back to back".
        act8 afrag[2][KH][MT], bfrag[2][KH][NT];
        auto read_frags = [&](int buf, int ks) {
          const int ti = tap < a.taps ? tap : a.taps - 1;  // padded K: B is zero
          int aoff;
          if (R == 2) {
            // packed position -> tap (parity-major operands: evens, then odds)
            const int t = !a.pmajor       ? ti
                          : ti < half_taps ? 2 * ti
                                           : 2 * (ti - half_taps) + 1;
            // SP: only the parity region of this part is resident, at row 0
            aoff = ((SP ? 0 : (t & 1) * regionRows) + (t >> 1)) * a.pitchA +
                   q8l * 8;
          } else {
            aoff = ti * a.pitchA + q8l * 8;
          }
          if constexpr (R == 2 && UNI) {
            if (narrow_now) {
              // the four k-groups of a K-step are four consecutive taps of one
              // parity (same 8 channels): k-group p of K-step npos reads window
              // row (4 * (npos & 3) + p) of parity region npos >> 2; slots past
              // taps/2 carry zero weights (row clamped: LDS may hold anything)
              const int kq = 4 * (npos & 3);
              const int reg = SP ? 0 : (npos >> 2) * regionRows;
              ++npos;
#pragma unroll
              for (int kh = 0; kh < KH; ++kh) {
                int idx = kq + (MF == 32 ? 2 * kh : 0) + g;
                idx = idx < half_taps ? idx : half_taps - 1;
                aoff = (reg + idx) * a.pitchA;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                  bfrag[buf][kh][nt] = *reinterpret_cast<const act8*>(
                      curB + nt * MF * kRowB + boff[ks][kh]);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                  afrag[buf][kh][mt] = *reinterpret_cast<const act8*>(
                      ldsA + rowbase[mt] + aoff);
              }
              return;
            }
          }
#pragma unroll
          for (int kh = 0; kh < KH; ++kh) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              bfrag[buf][kh][nt] = *reinterpret_cast<const act8*>(
                  curB + nt * MF * kRowB + boff[ks][kh]);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
              afrag[buf][kh][mt] = *reinterpret_cast<const act8*>(
                  ldsA + (rowbase[mt] + goff) + aoff + kh * 16);
          }
          q8l += 4;
          if (q8l >= a.c8) {
            q8l -= a.c8;
            ++tap;
          }
        };
        read_frags(0, 0);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          if (PIPE && ks + 1 < KS) {
            read_frags((ks + 1) & 1, ks + 1);
            // keep these reads ahead of the MFMAs below (the scheduler would
            // otherwise sink each read to just before its use and wait on it)
            __builtin_amdgcn_sched_barrier(0);
          }
#pragma unroll
          for (int kh = 0; kh < KH; ++kh)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
              for (int nt = 0; nt < NT; ++nt) {
                if constexpr (MF == 16)
                  acc[mt][nt] = cg_mfma_16x16x32(
                      afrag[ks & 1][kh][mt], bfrag[ks & 1][kh][nt],
                      acc[mt][nt], 0, 0, 0);
                else
                  acc[mt][nt] = cg_mfma_32x32x16(
                      afrag[ks & 1][kh][mt], bfrag[ks & 1][kh][nt],
                      acc[mt][nt], 0, 0, 0);
              }
          if (PIPE) __builtin_amdgcn_sched_barrier(0);
          if (!PIPE && ks + 1 < KS) read_frags((ks + 1) & 1, ks + 1);
        }
      }
    }
  }

  // ---- epilogue: accumulators -> LDS -> whole-line row-contiguous stores ----
  __syncthreads();
  float* scr = reinterpret_cast<float*>(smem) + wave * (16 * kScrPitch);
  // Read-back map: 8 rows per pass, 8 lanes per row, 8 output channels per lane
  // as two groups of four, placed so that EVERY store instruction writes whole
  // 128-byte runs per row (full sectors; 16-byte pieces at a 32/64-byte stride
  // would leave the L2 to merge half-written sectors):
  //   bf16 out: columns cg*8 + {0..3 | 4..7}      -> one 16-byte store
  //   f32  out: columns cg*4 + {0..3} | 32 + same -> two 16-byte stores
  const int erow = lane >> 3;
  const int cg8 = lane & 7;
  const bool of32 = !LN && a.out_f32;  // (the LayerNorm form stores bf16)
  const int colA = of32 ? cg8 * 4 : cg8 * 8;
  const int colB = of32 ? colA + 32 : colA + 4;
  const int nA = n0 + wn * 64 + colA;
  const int nB = n0 + wn * 64 + colB;
  float ssq = 0.f;  // sum of squares of this lane's outputs (rowsumsq)
  float bv[8];      // this lane's 8 output channels' bias (0 when absent/pad)
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    bv[e] = (a.bias && nA + e < a.N) ? a.bias[nA + e] : 0.f;
    bv[4 + e] = (a.bias && nB + e < a.N) ? a.bias[nB + e] : 0.f;
  }
  // fused LayerNorm + LeakyReLU (128-column tiles): the row statistics span
  // the two waves that share a row block (wave ^ 1); their partial sums meet
  // in a small LDS table behind the transpose scratch
  float* part = reinterpret_cast<float*>(smem) + kScratchBytes / 4;
  float* lnp = part + 4 * 16 * 2;  // gamma[128] | beta[128] (zero past N)
  if constexpr (LN) {
    if (tid < 128) {
      // (kept in LDS rather than 16 registers per lane; first read after the
      // first pass's workgroup barrier)
      lnp[tid] = tid < a.N ? a.ln_gamma[tid] : 0.f;
      lnp[128 + tid] = tid < a.N ? a.ln_beta[tid] : 0.f;
    }
  }
  // 16 rows x 64 columns of the wave tile at a time
#pragma unroll
  for (int mh = 0; mh < MT * KH; ++mh) {
    const int mt = mh / KH;
    const int h = mh % KH;  // 16-row half of a 32-row subtile
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      if constexpr (MF == 16) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          scr[(4 * g + r) * kScrPitch + nt * 16 + rM] = acc[mt][nt][r];
      } else {
        // 32x32 accumulator: register j <-> row 8*(j/4) + 4*g + j%4, column rM
#pragma unroll
        for (int j = 0; j < 8; ++j)
          scr[((j >> 2) * 8 + 4 * g + (j & 3)) * kScrPitch + nt * 32 + rM] =
              acc[mt][nt][8 * h + j];
      }
    }
    // the scratch is private to this wave and a wave's LDS ops complete in
    // order: a wave barrier (no s_barrier) is enough
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int row = pass * 8 + erow;
      const int m = m0 + (wm * MT + mt) * MF + h * 16 + row;
      if constexpr (LN) {
        {
          // (every lane takes part: the workgroup barrier below is uniform)
          const f32x4 v0 =
              *reinterpret_cast<const f32x4*>(scr + row * kScrPitch + colA);
          const f32x4 v1 =
              *reinterpret_cast<const f32x4*>(scr + row * kScrPitch + colB);
          float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
          float s1 = 0.f, s2 = 0.f;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int n = (e < 4 ? nA : nB - 4) + e;
            // statistics of the STORED (bf16) pre-activation, as the separate
            // cg_ln_lrelu_fwd pass sees it
            v[e] = n < a.N ? act2f(f2act(v[e] + bv[e])) : 0.f;
            s1 += v[e];
            s2 += v[e] * v[e];
          }
#pragma unroll
          for (int o = 1; o < 8; o <<= 1) {
            s1 += __shfl_xor(s1, o, 64);
            s2 += __shfl_xor(s2, o, 64);
          }
          if (cg8 == 0)
            *reinterpret_cast<float2*>(part + (wave * 16 + row) * 2) =
                make_float2(s1, s2);
          __syncthreads();
          const float2 o2 = *reinterpret_cast<const float2*>(
              part + ((wave ^ 1) * 16 + row) * 2);
          const float invn = 1.f / (float)a.N;
          const float mean = (s1 + o2.x) * invn;
          const float var = fmaxf((s2 + o2.y) * invn - mean * mean, 0.f);
          const float rstd = rsqrtf(var + a.ln_eps);
          if (m < a.M && nA < a.Cy) {
            const int b = m / a.Lu;
            const int u = m - b * a.Lu;
            const long long ridx =
                (long long)b * a.Ly + (long long)a.y_stride * u + y_off;
            const long long rowoff = ridx * a.Cy;
            const int lc = wn * 64 + colA;
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(lnp + lc);
            const f32x4 g1 = *reinterpret_cast<const f32x4*>(lnp + lc + 4);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(lnp + 128 + lc);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(lnp + 128 + lc + 4);
            float hv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float t = (v[e] - mean) * rstd * (e < 4 ? g0[e] : g1[e - 4]) +
                              (e < 4 ? b0[e] : b1[e - 4]);
              hv[e] = fmaxf(t, a.alpha * t);
            }
            // (forward-only callers -- G(z) of a critic update -- pass no
            // statistics buffers: the pre-activation is then not stored either)
            if (a.ln_mean)
              *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(a.y) + rowoff +
                                        nA) =
                  make_uint4(pack2act(v[0], v[1]), pack2act(v[2], v[3]),
                             pack2act(v[4], v[5]), pack2act(v[6], v[7]));
            *reinterpret_cast<uint4*>(a.ln_h + rowoff + nA) =
                make_uint4(pack2act(hv[0], hv[1]), pack2act(hv[2], hv[3]),
                           pack2act(hv[4], hv[5]), pack2act(hv[6], hv[7]));
            if (a.ln_mean && wn == 0 && cg8 == 0) {
              a.ln_mean[ridx] = mean;
              a.ln_rstd[ridx] = rstd;
            }
          }
        }
      }
      if (!LN && m < a.M && nA < a.Cy) {
        const int b = m / a.Lu;
        const int u = m - b * a.Lu;
        int t = a.y_stride * u + y_off;
        bool to_side = false;  // reflected-branch row of the unshuffle
        if (a.out_shifts) {
          const int s = a.out_shifts[b / a.out_seg];
          if (s > 0) {
            to_side = t >= a.Ly - s;
            t = to_side ? t - (a.Ly - s) : t + s;
          } else {
            to_side = t < -s;
            t = to_side ? t : t + s;
          }
        }
        const long long rowoff =
            ((long long)b * (to_side ? a.side_rows : a.Ly) + t) * a.Cy;
        const f32x4 v0 =
            *reinterpret_cast<const f32x4*>(scr + row * kScrPitch + colA);
        const f32x4 v1 =
            *reinterpret_cast<const f32x4*>(scr + row * kScrPitch + colB);
        float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        const bool okB = nB < a.Cy;  // f32 out: second half past the pitch
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += bv[e];
        if (a.epilogue == CG_EPI_LRELU) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], a.alpha * v[e]);
        } else if (a.epilogue == CG_EPI_MASK && !to_side) {
          if (a.out_shifts) {
            // the unfused form stores this gradient in bf16 before masking
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = act2f(f2act(v[e]));
          }
          const uint2 ha = *reinterpret_cast<const uint2*>(a.mask + rowoff + nA);
          const uint2 hb = okB ? *reinterpret_cast<const uint2*>(a.mask + rowoff + nB)
                               : make_uint2(0u, 0u);
          const uint32_t hw[4] = {ha.x, ha.y, hb.x, hb.y};
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const uint16_t hv = (uint16_t)(hw[e >> 1] >> ((e & 1) * 16));
            v[e] *= (act2f(hv) > 0.f) ? 1.f : a.alpha;
          }
        } else if (a.epilogue == CG_EPI_SIGMOID) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = 1.f / (1.f + __expf(-v[e]));
        }
        // channel padding [N, Cy) stays exactly zero
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (nA + e >= a.N) v[e] = 0.f;
          if (nB + e >= a.N) v[4 + e] = 0.f;
        }
        if (a.rowsumsq) {
#pragma unroll
          for (int e = 0; e < 8; ++e) ssq += v[e] * v[e];
        }
        if (a.out_f32) {
          float* dst = reinterpret_cast<float*>(a.y) + rowoff;
          *reinterpret_cast<f32x4*>(dst + nA) = f32x4{v[0], v[1], v[2], v[3]};
          if (okB)
            *reinterpret_cast<f32x4*>(dst + nB) = f32x4{v[4], v[5], v[6], v[7]};
        } else {
          uint16_t* dst =
              (to_side ? a.side : reinterpret_cast<uint16_t*>(a.y)) + rowoff + nA;
          *reinterpret_cast<uint4*>(dst) =
              make_uint4(pack2act(v[0], v[1]), pack2act(v[2], v[3]),
                         pack2act(v[4], v[5]), pack2act(v[6], v[7]));
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  if (a.rowsumsq) {
    // the whole tile belongs to one sample (nseg == 1, checked on the host):
    // one f32 atomic per WORKGROUP -- the atomics of a sample all hit one
    // address and serialise in the L2 (measured 24 us of an 88 us launch with
    // one per wave)
    ssq = wave_sum(ssq);
    float* wsum = reinterpret_cast<float*>(smem) + kScratchBytes / 4;
    if (lane == 0) wsum[wave] = ssq;
    __syncthreads();
    if (tid == 0 && m0 < a.M) {
      const float t = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
      const int b = m0 / a.Lu;
      if (a.ssq_ws) {
        const int rt = (m0 - b * a.Lu) / TM;
        a.ssq_ws[(long long)b * a.ssq_P + (rt * a.gp + phase) * a.gn + bn] = t;
      } else {
        atomicAdd(a.rowsumsq + b, t);
      }
    }
  }
}

// second half of the ordered penalty norm: rowsumsq[b] = the sample's slots
// added in slot order
__global__ void rowsumsq_finish_kernel(const float* __restrict__ ws, int P,
                                       float* __restrict__ out, int nB) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nB) return;
  float s = 0.f;
  for (int j = 0; j < P; ++j) s += ws[(long long)b * P + j];
  out[b] = s;
}

// Split-K finishing pass (cg_conv_desc.ksplit): one thread per 8 output
// channels adds the splits' f32 partial sums, then bias / LeakyReLU / mask,
// and stores bf16 (channel padding [N, Cy) stays zero).
struct SplitFinishArgs {
  const float* ws;
  int nsplit;
  long long stride;  // f32 elements per split
  const float* bias;
  const uint16_t* mask;
  uint16_t* y;
  int N, Cy, epilogue;
  float alpha;
  long long total8;
};

__global__ __launch_bounds__(256) void split_finish_kernel(SplitFinishArgs f) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= f.total8) return;
  const long long e0 = idx * 8;
  const int c = (int)(e0 % f.Cy);
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
  for (int z = 0; z < f.nsplit; ++z) {
    const float* p = f.ws + z * f.stride + e0;
    s0 += *reinterpret_cast<const f32x4*>(p);
    s1 += *reinterpret_cast<const f32x4*>(p + 4);
  }
  float v[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
#pragma unroll
  for (int e = 0; e < 8; ++e)
    if (f.bias && c + e < f.N) v[e] += f.bias[c + e];
  if (f.epilogue == CG_EPI_LRELU) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], f.alpha * v[e]);
  } else if (f.epilogue == CG_EPI_MASK) {
    const uint4 h = *reinterpret_cast<const uint4*>(f.mask + e0);
    const uint32_t hw[4] = {h.x, h.y, h.z, h.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const uint16_t hv = (uint16_t)(hw[e >> 1] >> ((e & 1) * 16));
      v[e] *= (act2f(hv) > 0.f) ? 1.f : f.alpha;
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e)
    if (c + e >= f.N) v[e] = 0.f;
  *reinterpret_cast<uint4*>(f.y + e0) =
      make_uint4(pack2act(v[0], v[1]), pack2act(v[2], v[3]), pack2act(v[4], v[5]),
                 pack2act(v[6], v[7]));
}

// ---------------------------------------------------------------------------
// weight packing
// ---------------------------------------------------------------------------
struct PackArgs {
  const float* src;
  uint16_t* dst;
  int taps, tap0, tap_step;
  long long s_tap, s_c, s_n;
  int C_real, N_real, CK, c8, nchunks, Fp;
  long long Kpack, total;
  int npad;          // padded column count: total / 8 / (nchunks * Fp)
  int parity_major;  // packed tap i holds tap 2i (i < taps/2) else 2(i-taps/2)+1
  int narrow_last;   // last chunk: 8 channels per tap, 16 slots per tap parity
};

// One thread packs one 16-byte group: 8 consecutive channels of one
// (column n, chunk, tap) -- a single 16-byte store instead of eight 2-byte ones.
// the 16-byte group (column n, chunk cc, position f) of a packed operand
__device__ __forceinline__ uint4 pack_values(const PackArgs& a, int n, int cc, int f) {
  int tap = f / a.c8;
  int q8 = f - tap * a.c8;
  const int half = a.taps >> 1;
  if (a.narrow_last && cc == a.nchunks - 1) {
    // narrow last chunk (<= 8 real channels): position f < 32 is the first
    // 8-channel group of tap parity f >> 4, index f & 15 (slots past taps/2
    // and positions >= 32 stay zero; the kernel stops after position 31)
    const int idx = f & 15;
    tap = (f < 32 && idx < half) ? (f >> 4) * half + idx : a.taps;
    q8 = 0;
  }
  const int c = cc * a.CK + q8 * 8;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = 0.f;
  if (tap < a.taps && n < a.N_real) {
    const int tsrc = !a.parity_major ? tap
                     : tap < half   ? 2 * tap
                                    : 2 * (tap - half) + 1;
    const float* src = a.src + (long long)(a.tap0 + tsrc * a.tap_step) * a.s_tap +
                       n * a.s_n;
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (c + e < a.C_real) v[e] = src[(long long)(c + e) * a.s_c];
  }
  return make_uint4(pack2act(v[0], v[1]), pack2act(v[2], v[3]), pack2act(v[4], v[5]),
                    pack2act(v[6], v[7]));
}

// Block `blk` of an operand (256 groups).  Source contiguous in the channels
// (s_c == 1: the operand's own order is the source's): thread = group, in
// destination order.  Source contiguous in n (s_n == 1: the TF kernel layout read
// as a forward operand -- a transpose): the block takes a tile of 32 columns x 8
// positions; the gathers run along n (128-byte runs), the groups turn around
// through LDS, and the stores run along the operand's rows (8 x 16 = 128
// contiguous bytes per column) -- as 64 scattered 16-byte stores per wave they
// held this kernel at 2.9 TB/s (profiles/r05_hbm_rates.txt).
// (32-bit index arithmetic: an operand has < 2^31 groups -- checked on the host)
__device__ __forceinline__ void pack_block(const PackArgs& a, unsigned blk, uint4* lds) {
  const unsigned tid = threadIdx.x;
  if (a.s_n == 1) {
    // tiles: position tile fastest, then column tile, then chunk
    const unsigned ft = (unsigned)a.Fp >> 3, nt = (unsigned)a.npad >> 5;
    const unsigned f0 = (blk % ft) * 8;
    const unsigned q = blk / ft;
    const unsigned n0 = (q % nt) * 32;
    const int cc = (int)(q / nt);
    if (cc >= a.nchunks) return;
    lds[(tid & 31) * 9 + (tid >> 5)] =
        pack_values(a, (int)(n0 + (tid & 31)), cc, (int)(f0 + (tid >> 5)));
    __syncthreads();
    const unsigned n = n0 + (tid >> 3), f = f0 + (tid & 7);
    const unsigned r = (n * (unsigned)a.nchunks + (unsigned)cc) * (unsigned)a.Fp + f;
    *reinterpret_cast<uint4*>(a.dst + (long long)r * 8) = lds[(tid >> 3) * 9 + (tid & 7)];
    return;
  }
  const unsigned r = blk * 256 + tid;
  if ((long long)r * 8 >= a.total) return;
  const int f = (int)(r % (unsigned)a.Fp);
  const unsigned q = r / (unsigned)a.Fp;
  const int cc = (int)(q % (unsigned)a.nchunks);
  const int n = (int)(q / (unsigned)a.nchunks);
  *reinterpret_cast<uint4*>(a.dst + (long long)r * 8) = pack_values(a, n, cc, f);
}

constexpr int kPackBlock = 256 * 8;  // elements packed by one 256-thread block

__global__ __launch_bounds__(256) void pack_kernel(PackArgs a) {
  __shared__ uint4 lds[32 * 9];
  pack_block(a, blockIdx.x, lds);
}

// All operands of one model in ONE launch: `table` holds n PackArgs, block b
// works on descriptor desc_of_block[b] starting at 16-byte group
// (b - first_block[desc]) * 256.
__global__ __launch_bounds__(256) void pack_batched_kernel(const PackArgs* __restrict__ table,
                                    const int* __restrict__ desc_of_block,
                                    const int* __restrict__ first_block) {
  __shared__ uint4 lds[32 * 9];
  const int di = desc_of_block[blockIdx.x];
  const PackArgs a = table[di];
  pack_block(a, blockIdx.x - (unsigned)first_block[di], lds);
}

inline int ilog2(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}

}  // namespace

extern "C" int cg_abi_version(void) { return CG_ABI_VERSION; }

extern "C" int cg_act_dtype(void) { return CG_ACT_F16 ? CG_DTYPE_F16 : CG_DTYPE_BF16; }

extern "C" int cg_struct_size(int which) {
  switch (which) {
    case 0: return (int)sizeof(cg_conv_desc);
    case 1: return (int)sizeof(cg_pack_desc);
    case 2: return (int)sizeof(cg_wgrad_desc);
    default: return -1;
  }
}

// ---------------------------------------------------------------------------
// launch profiler (process-wide, eager launches only; see cg_common.h)
// ---------------------------------------------------------------------------
namespace {
struct ProfState {
  std::vector<hipEvent_t> start, stop;
  std::vector<int> family;
  int used = 0;
  bool on = false;
} g_prof;
}  // namespace

bool cg_prof_next(int family, hipEvent_t* start, hipEvent_t* stop) {
  if (!g_prof.on || g_prof.used >= (int)g_prof.start.size()) return false;
  *start = g_prof.start[g_prof.used];
  *stop = g_prof.stop[g_prof.used];
  g_prof.family[g_prof.used] = family;
  ++g_prof.used;
  return true;
}

extern "C" int cg_profile_enable(int max_launches) {
  g_prof.used = 0;
  g_prof.on = max_launches > 0;
  while ((int)g_prof.start.size() < max_launches) {
    hipEvent_t s, e;
    hipError_t err = hipEventCreate(&s);
    if (err == hipSuccess) err = hipEventCreate(&e);
    if (err != hipSuccess) return (int)err;
    g_prof.start.push_back(s);
    g_prof.stop.push_back(e);
    g_prof.family.push_back(0);
  }
  return 0;
}

extern "C" int cg_profile_collect(float* ms, int* family, int capacity) {
  const int n = g_prof.used < capacity ? g_prof.used : capacity;
  for (int i = 0; i < n; ++i) {
    hipError_t err = hipEventSynchronize(g_prof.stop[i]);
    if (err == hipSuccess)
      err = hipEventElapsedTime(&ms[i], g_prof.start[i], g_prof.stop[i]);
    if (err != hipSuccess) return -(int)err;
    family[i] = g_prof.family[i];
  }
  g_prof.used = 0;
  g_prof.on = false;
  return n;
}

extern "C" long long cg_packed_elems(int N, int taps, int Cx, int CK) {
  if (CK < 32 || CK % 8 || Cx % CK || taps < 1 || N < 1) return -1;
  const int c8 = CK / 8;
  const int Fp = (taps * c8 + 15) / 16 * 16;
  const long long Npad = (N + 127) / 128 * 128;  // whole 128-column tiles
  return Npad * (long long)(Cx / CK) * Fp * 8;
}

static bool narrow_ok(const cg_pack_desc* d) {
  return !d->narrow_last ||
         (d->parity_major && d->CK == 32 && d->taps <= 32 && !(d->taps & 1) &&
          d->Cx >= 64 && d->C_real > d->Cx - 32 && d->C_real <= d->Cx - 24);
}

static int fill_pack_args(const cg_pack_desc* d, PackArgs& a) {
  const long long total = cg_packed_elems(d->N_real, d->taps, d->Cx, d->CK);
  if (total < 0 || d->C_real > d->Cx || !narrow_ok(d)) return CG_EINVAL;
  a.src = d->src;
  a.dst = reinterpret_cast<uint16_t*>(d->dst);
  a.taps = d->taps; a.tap0 = d->tap0; a.tap_step = d->tap_step;
  a.parity_major = d->parity_major;
  a.narrow_last = d->narrow_last;
  a.s_tap = d->s_tap; a.s_c = d->s_c; a.s_n = d->s_n;
  a.C_real = d->C_real; a.N_real = d->N_real; a.CK = d->CK;
  a.c8 = d->CK / 8; a.nchunks = d->Cx / d->CK;
  a.Fp = (d->taps * a.c8 + 15) / 16 * 16;
  a.Kpack = (long long)a.nchunks * a.Fp * 8;
  a.total = total;
  if (total / 8 >= (1ll << 31)) return CG_EINVAL;
  a.npad = (int)(total / 8 / ((long long)a.nchunks * a.Fp));
  return 0;
}

extern "C" long long cg_pack_plan_bytes(int n, long long total_blocks) {
  return (long long)n * sizeof(PackArgs) + (total_blocks + n) * sizeof(int);
}

extern "C" long long cg_pack_plan_build(const cg_pack_desc* descs, int n,
                                        void* host_buf, long long host_bytes) {
  // pass 1: block counts
  long long blocks = 0;
  for (int i = 0; i < n; ++i) {
    PackArgs a;
    if (fill_pack_args(descs + i, a)) return -1;
    blocks += (a.total + kPackBlock - 1) / kPackBlock;
  }
  if (!host_buf) return blocks;
  if (host_bytes < cg_pack_plan_bytes(n, blocks)) return -1;
  PackArgs* table = reinterpret_cast<PackArgs*>(host_buf);
  int* desc_of_block = reinterpret_cast<int*>(table + n);
  int* first_block = desc_of_block + blocks;
  long long b = 0;
  for (int i = 0; i < n; ++i) {
    fill_pack_args(descs + i, table[i]);
    first_block[i] = (int)b;
    const long long nb = (table[i].total + kPackBlock - 1) / kPackBlock;
    for (long long k = 0; k < nb; ++k) desc_of_block[b + k] = i;
    b += nb;
  }
  return blocks;
}

extern "C" int cg_pack_batched(const void* dev_plan, int n, long long blocks,
                               void* stream) {
  if (!dev_plan || n < 1 || blocks < 1) return CG_EINVAL;
  const PackArgs* table = reinterpret_cast<const PackArgs*>(dev_plan);
  const int* desc_of_block = reinterpret_cast<const int*>(table + n);
  const int* first_block = desc_of_block + blocks;
  hipLaunchKernelGGL(pack_batched_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     (hipStream_t)stream, table, desc_of_block, first_block);
  CG_LAUNCH_CHECK();
}

extern "C" int cg_pack_weights(const cg_pack_desc* d, void* stream) {
  PackArgs a;
  if (fill_pack_args(d, a)) return CG_EINVAL;
  const long long total = a.total;
  const int threads = 256;
  const long long blocks = (total + kPackBlock - 1) / kPackBlock;
  hipLaunchKernelGGL(pack_kernel, dim3((unsigned)blocks), dim3(threads), 0,
                     (hipStream_t)stream, a);
  CG_LAUNCH_CHECK();
}

struct SplitProf { bool on; hipEvent_t start; };
static SplitProf g_split_prof = {false, nullptr};
// cg_swconv_check: walk the whole validation + dispatch path without launching
static thread_local bool g_dry_run = false;  // (per calling thread: the C ABI may be driven from several)

template <int R, int MF, int WGN, int MT, int KS, bool UNI, bool SP = false,
          bool LN = false>
static int launch_swconv1(const ConvArgs& a, dim3 grid, size_t lds,
                          hipStream_t stream) {
  if (g_dry_run) return 0;  // this instantiation exists: the launch is valid
  static CgPerDeviceFlag attr_set;
  if (!attr_set.test()) {
    hipError_t e = hipFuncSetAttribute(
        reinterpret_cast<const void*>(
            &swconv_kernel<R, MF, WGN, MT, KS, UNI, SP, LN>),
        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set.mark();
  }
  if (g_split_prof.on) {
    // split-K launch being timed: cg_swconv holds the event pair; this kernel
    // carries its start, the finishing launch its stop (one timed launch)
    hipExtLaunchKernelGGL((swconv_kernel<R, MF, WGN, MT, KS, UNI, SP, LN>), grid,
                          dim3(256), lds, stream, g_split_prof.start, nullptr, 0,
                          a);
    CG_LAUNCH_CHECK();
  }
  CG_LAUNCH_PROF(CG_FAMILY_SWCONV,
                 (swconv_kernel<R, MF, WGN, MT, KS, UNI, SP, LN>), grid,
                 dim3(256), lds, stream, a);
  CG_LAUNCH_CHECK();
}

template <int R, int MF, int WGN, int MT, int KS>
static int launch_swconv(const ConvArgs& a, dim3 grid, size_t lds, bool sp,
                         hipStream_t stream) {
  if constexpr (R == 2) {
    if (sp) return launch_swconv1<R, MF, WGN, MT, KS, true, true>(a, grid, lds, stream);
  }
  if (a.epilogue == CG_EPI_LN_LRELU) {
    // Conv1DTranspose + LayerNorm: stride-1 phases on the 128-column tiles
    // (uniform 32-channel K walk)
    if constexpr (R == 1 && WGN == 2) {
      if ((a.c8 & 3) == 0)
        return launch_swconv1<R, MF, WGN, MT, KS, true, false, true>(a, grid, lds,
                                                                    stream);
    }
    return CG_EINVAL;
  }
  if ((a.c8 & 3) == 0)
    return launch_swconv1<R, MF, WGN, MT, KS, true>(a, grid, lds, stream);
  if constexpr (MF == 16)
    return launch_swconv1<R, MF, WGN, MT, KS, false>(a, grid, lds, stream);
  return CG_EINVAL;
}

// cg_conv_desc.tile -> (MFMA rows, waves along N, subtiles per wave)
// (swp_wm > 0: a software-pipelined tile of swconv_swp.hip: swp_wm x wgn waves,
// mt 16-row subtiles per wave)
struct TileCfg { int mf, wgn, mt, swp_wm; };
static const TileCfg kTileCfgs[CG_NUM_TILES] = {
    {16, 1, 4, 0},  // CG_TILE_256x64
    {16, 1, 1, 0},  // CG_TILE_64x64
    {16, 1, 2, 0},  // CG_TILE_128x64
    {32, 1, 2, 0},  // CG_TILE_256x64_M32
    {32, 1, 1, 0},  // CG_TILE_128x64_M32
    {32, 2, 4, 0},  // CG_TILE_256x128_M32
    {32, 2, 2, 0},  // CG_TILE_128x128_M32
    {16, 2, 8, 0},  // CG_TILE_256x128
    {16, 2, 4, 0},  // CG_TILE_128x128
    {16, 1, 4, 8},  // CG_TILE_SWP_512x64
    {16, 1, 4, 4},  // CG_TILE_SWP_256x64
    {16, 2, 4, 4},  // CG_TILE_SWP_256x128
    {16, 2, 4, 2},  // CG_TILE_SWP_128x128
    {16, 1, 2, 4},  // CG_TILE_SWP_128x64
    {16, 1, 2, 8},  // CG_TILE_SWP_256x64_W8
    {16, 2, 2, 4},  // CG_TILE_SWP_128x128_W8
};

extern "C" int cg_tile_shape(int tile, int* rows, int* cols) {
  if (tile < 0 || tile >= CG_NUM_TILES) return CG_EINVAL;
  const TileCfg& t = kTileCfgs[tile];
  if (rows) *rows = t.swp_wm ? t.swp_wm * t.mt * 16 : (4 / t.wgn) * t.mt * t.mf;
  if (cols) *cols = 64 * t.wgn;
  return 0;
}

// split-K finishing launch: y = epi(sum_z ws[z] + bias), bf16.  own_pair: the
// launch is profiled as a family launch of its own (software-pipelined tiles);
// else `stop` (if any) closes the event pair the main launch opened.
static int split_finish_launch(const cg_conv_desc* d, const ConvArgs& a, hipStream_t s,
                        bool own_pair, hipEvent_t stop) {
  SplitFinishArgs f;
  f.ws = d->split_ws; f.nsplit = a.ksplit; f.stride = a.split_stride;
  f.bias = d->bias; f.mask = reinterpret_cast<const uint16_t*>(d->mask_src);
  f.y = reinterpret_cast<uint16_t*>(d->y);
  f.N = d->N; f.Cy = d->Cy; f.epilogue = d->epilogue; f.alpha = d->alpha;
  f.total8 = a.split_stride / 8;
  const long long blocks = (f.total8 + 255) / 256;
  if (own_pair) {
    CG_LAUNCH_PROF(CG_FAMILY_SWCONV, split_finish_kernel, dim3((unsigned)blocks),
                   dim3(256), 0, s, f);
  } else if (stop) {
    hipExtLaunchKernelGGL(split_finish_kernel, dim3((unsigned)blocks), dim3(256),
                          0, s, nullptr, stop, 0, f);
  } else {
    hipLaunchKernelGGL(split_finish_kernel, dim3((unsigned)blocks), dim3(256), 0,
                       s, f);
  }
  CG_LAUNCH_CHECK();
}

static int swconv_run(const cg_conv_desc* d, void* stream);

extern "C" int cg_swconv(const cg_conv_desc* d, void* stream) {
  return swconv_run(d, stream);
}

extern "C" int cg_swconv_check(const cg_conv_desc* d) {
  g_dry_run = true;
  const int rc = swconv_run(d, nullptr);
  g_dry_run = false;
  return rc;
}

// CALCIUMGAN_LAUNCH_LOG=<file> (diagnostics: tools/traffic_by_geometry.sh): one
// line per cg_swconv launch, in launch order, with the geometry that decides its
// algorithmic bytes -- joined with the per-dispatch PMC rows of a profiled eager
// run to attribute HBM traffic to launch geometries.
static FILE* launch_log() {
  static FILE* f = [] {
    const char* p = getenv("CALCIUMGAN_LAUNCH_LOG");
    return p && *p ? fopen(p, "a") : (FILE*)nullptr;
  }();
  return f;
}
static void log_launch(const cg_conv_desc* d) {
  FILE* f = launch_log();
  if (!f) return;
  fprintf(f, "swconv stride=%d taps=%d nB=%d Lx=%d Cx=%d Lu=%d N=%d Ly=%d Cy=%d CK=%d "
          "nphase=%d epi=%d f32=%d tile=%d ksplit=%d narrow=%d mask=%d shifts=%d "
          "oshifts=%d ln=%d ssq=%d rscale=%d ystride=%d sp=%d\n",
          d->stride, d->taps, d->nB, d->Lx, d->Cx, d->Lu, d->N, d->Ly, d->Cy, d->CK,
          d->nphase, d->epilogue, d->out_f32, d->tile, d->ksplit > 1 ? d->ksplit : 1,
          d->w_narrow_last, d->mask_src ? 1 : 0, d->shifts ? 1 : 0,
          d->out_shifts ? 1 : 0, d->ln_gamma ? (d->ln_mean ? 2 : 1) : 0,
          d->rowsumsq ? 1 : 0, d->row_scale ? 1 : 0, d->y_stride, d->split_parity);
  fflush(f);
}

static int swconv_run(const cg_conv_desc* d, void* stream) {
  if (!d || !d->x || !d->w || !d->y) return CG_EINVAL;
  if (d->stride != 1 && d->stride != 2) return CG_EINVAL;
  if (d->stride == 2 && (d->taps & 1)) return CG_EINVAL;
  if (d->CK < 32 || d->CK % 8 || d->Cx % d->CK || d->Cy % 8) return CG_EINVAL;
  if (d->taps < 1 || d->Lu < 1 || d->nB < 1 || d->N < 1 || d->N > d->Cy)
    return CG_EINVAL;
  if (d->nphase != 1 && d->nphase != 2) return CG_EINVAL;
  if (d->epilogue == CG_EPI_MASK && !d->mask_src) return CG_EINVAL;
  if (d->shifts && d->seg_size < 1) return CG_EINVAL;
  const int R = d->stride;
  if (d->tile < 0 || d->tile >= CG_NUM_TILES) return CG_EINVAL;
  const TileCfg tc = kTileCfgs[d->tile];
  if (d->out_shifts &&
      (d->out_f32 || d->rowsumsq || d->out_seg_size < 1 || !d->side ||
       d->side_rows < 1 || d->epilogue == CG_EPI_LN_LRELU ||
       d->epilogue == CG_EPI_SIGMOID || d->epilogue == CG_EPI_LRELU))
    return CG_EINVAL;
  if (d->epilogue == CG_EPI_LN_LRELU &&
      (tc.wgn != 2 || d->N > 128 || d->out_f32 || d->rowsumsq || !d->ln_gamma ||
       !d->ln_beta || !d->ln_h || (!d->ln_mean != !d->ln_rstd)))
    return CG_EINVAL;
  const int TM = tc.swp_wm ? tc.swp_wm * tc.mt * 16 : (4 / tc.wgn) * tc.mt * tc.mf;
  const int TN = 64 * tc.wgn;
  if (tc.mf == 32 && (d->CK / 8) % 4) return CG_EINVAL;  // uniform K walk only
  int S;
  if (d->Lu >= TM) {
    if (d->Lu % TM) return CG_EINVAL;
    S = TM;
  } else {
    if (TM % d->Lu) return CG_EINVAL;
    S = d->Lu;
  }
  ConvArgs a;
  a.x = reinterpret_cast<const uint16_t*>(d->x);
  a.w = reinterpret_cast<const uint16_t*>(d->w);
  a.y = d->y;
  a.bias = d->bias;
  a.mask = reinterpret_cast<const uint16_t*>(d->mask_src);
  a.shifts = d->shifts;
  a.rowsumsq = d->rowsumsq;
  a.nB = d->nB; a.Lx = d->Lx; a.Cx = d->Cx; a.seg_size = d->seg_size;
  a.taps = d->taps; a.off = d->off; a.Lu = d->Lu;
  a.M = d->nB * d->Lu;
  a.N = d->N; a.Ly = d->Ly; a.Cy = d->Cy; a.y_stride = d->y_stride;
  a.y_off = d->y_off;
  a.CK = d->CK; a.c8 = d->CK / 8; a.nchunks = d->Cx / d->CK;
  a.inv_c8 = 1.0f / (float)a.c8;
  a.log2c8 = (a.c8 & (a.c8 - 1)) ? -1 : ilog2(a.c8);
  a.Fp = (d->taps * a.c8 + 15) / 16 * 16;
  a.nstages = a.Fp / 16;  // 128-wide stages; the kernel derives its own
  a.Kpack = (long long)a.nchunks * a.Fp * 8;
  if (tc.mf == 16)
    a.pitchA = d->CK + 8 * ((6 - (a.c8 & 3)) & 3);  // slots == 2 (mod 4)
  else
    a.pitchA = d->CK + 8;                            // 4 | c8: odd slots
  a.S = S; a.log2S = ilog2(S); a.nseg = TM / S;
  a.WR = S + d->taps / R - 1;
  if (d->rowsumsq && a.nseg != 1) return CG_EINVAL;  // one sample per tile
  // stride-2 operand order / split-parity staging
  a.pmajor = (R == 2 && d->w_parity_major) ? 1 : 0;
  const bool sp = R == 2 && d->split_parity != 0;
  if (sp && (!a.pmajor || (a.c8 & 3) || (d->taps & 1))) return CG_EINVAL;
  a.ldsA_elems = (sp ? 1 : R) * a.nseg * a.WR * a.pitchA;
  a.epilogue = d->epilogue; a.out_f32 = d->out_f32; a.alpha = d->alpha;
  a.ln_gamma = d->ln_gamma; a.ln_beta = d->ln_beta;
  a.ln_h = reinterpret_cast<uint16_t*>(d->ln_h);
  a.ln_mean = d->ln_mean; a.ln_rstd = d->ln_rstd; a.ln_eps = d->ln_eps;
  a.ksplit = d->ksplit > 1 ? d->ksplit : 1;
  a.split_stride = (long long)d->nB * d->Ly * d->Cy;
  if (a.ksplit > 1) {
    if (a.nchunks % a.ksplit || d->out_f32 || d->rowsumsq || d->out_shifts ||
        (d->epilogue != CG_EPI_NONE && d->epilogue != CG_EPI_LRELU &&
         d->epilogue != CG_EPI_MASK) ||
        !d->split_ws || d->split_ws_elems < a.ksplit * a.split_stride)
      return CG_EINVAL;
    // the main launch stores raw f32 partial sums; cg_swconv's finishing
    // launch applies bias and epilogue
    a.nchunks /= a.ksplit;
    a.y = d->split_ws;
    a.bias = nullptr;
    a.out_f32 = 1;
    a.epilogue = CG_EPI_NONE;
  }
  a.narrow = 0;
  if (d->w_narrow_last) {
    if (R != 2 || !a.pmajor || d->CK != 32 || d->Cx / d->CK < 2 || d->taps > 32)
      return CG_EINVAL;
    a.narrow = 1;
  }
  a.out_shifts = d->out_shifts; a.out_seg = d->out_seg_size;
  a.side = reinterpret_cast<uint16_t*>(d->side); a.side_rows = d->side_rows;
  a.row_scale = d->row_scale;
  // (software-pipelined tiles only; the split-K finishing launch and the fused
  // LayerNorm do not carry it)
  if (d->row_scale && (!tc.swp_wm || a.ksplit > 1 ||
                       d->epilogue == CG_EPI_LN_LRELU))
    return CG_EINVAL;
  a.w_phase_stride = d->w_phase_stride;
  a.off_phase_step = d->off_phase_step;
  a.yoff_phase_step = d->yoff_phase_step;
  // ordered penalty norm: one slot per (row tile of the sample, phase, column
  // tile); the finishing launch below adds a sample's slots in order
  a.ssq_ws = nullptr;
  a.ssq_P = 0;
  if (d->rowsumsq_defer && !(d->rowsumsq && d->rowsumsq_ws)) return CG_EINVAL;
  if (d->rowsumsq && d->rowsumsq_ws) {
    a.ssq_P = (d->Lu / TM) * d->nphase * ((d->N + TN - 1) / TN);
    if (d->rowsumsq_ws_elems < (long long)d->nB * a.ssq_P) return CG_EINVAL;
    a.ssq_ws = d->rowsumsq_ws;
  }
  auto ssq_finish = [&](int rc) {
    if (rc || !a.ssq_ws || g_dry_run || d->rowsumsq_defer) return rc;
    hipLaunchKernelGGL(rowsumsq_finish_kernel, dim3((d->nB + 255) / 256), dim3(256),
                       0, (hipStream_t)stream, a.ssq_ws, a.ssq_P, d->rowsumsq,
                       d->nB);
    return (int)hipGetLastError();
  };
  if (tc.swp_wm) {
    // software-pipelined tile: its own LDS plan and launch (swconv_swp.hip;
    // inherently one parity at a time: split_parity is not consulted)
    if (a.Fp != d->taps * a.c8) return CG_EINVAL;
    a.gm = (a.M + TM - 1) / TM;
    a.gn = (d->N + TN - 1) / TN;
    a.gp = d->nphase;
    if (!g_dry_run) log_launch(d);
    const int rc = swconv_swp_launch(a, R, tc.swp_wm, tc.wgn, tc.mt, a.ksplit,
                                     g_dry_run, (hipStream_t)stream);
    if (rc || a.ksplit == 1 || g_dry_run) return ssq_finish(rc);
    // (both launches carry their own event pair under cg_profile_enable)
    return split_finish_launch(d, a, (hipStream_t)stream, true, nullptr);
  }
  size_t ldsA_bytes = (size_t)a.ldsA_elems * 2;
  if (ldsA_bytes < (size_t)kScratchBytes) ldsA_bytes = kScratchBytes;
  ldsA_bytes = (ldsA_bytes + 15) / 16 * 16;
  a.ldsA_elems = (int)(ldsA_bytes / 2);
  // weight-stage width: the narrow (64-wide) stage when it lets one more
  // workgroup fit on a CU (occupancy beats stage length: measured), else the
  // wide (128-wide) one
  const size_t cu_lds = 160 * 1024;
  int ks = 4;
  if (cu_lds / (ldsA_bytes + ldsB_bytes(2, TN)) >
      cu_lds / (ldsA_bytes + ldsB_bytes(4, TN)))
    ks = 2;
  if (d->stage_ksteps == 2 || d->stage_ksteps == 4) ks = d->stage_ksteps;
  const size_t lds = ldsA_bytes + ldsB_bytes(ks, TN);
  if (lds > 160 * 1024) return CG_EINVAL;
  // each parity's taps must fill whole weight stages (no K padding in between)
  if (sp && (a.Fp != d->taps * a.c8 || ((d->taps / 2) * a.c8) % (4 * ks)))
    return CG_EINVAL;
  a.gm = (a.M + TM - 1) / TM;
  a.gn = (d->N + TN - 1) / TN;
  a.gp = d->nphase;
  dim3 grid((unsigned)(((a.gm + 7) / 8) * 8 * a.gn * a.gp), (unsigned)a.ksplit);
  hipStream_t s = (hipStream_t)stream;
  int rc = CG_EINVAL;
  if (!g_dry_run) log_launch(d);
  hipEvent_t split_start = nullptr, split_stop = nullptr;
  const bool split_timed =
      !g_dry_run && a.ksplit > 1 &&
      cg_prof_next(CG_FAMILY_SWCONV, &split_start, &split_stop);
  g_split_prof.on = split_timed;
  g_split_prof.start = split_start;
#define CG_DISPATCH(RR, FF, WW, MM, KK)                                        \
  else if (R == RR && tc.mf == FF && tc.wgn == WW && tc.mt == MM && ks == KK) \
    rc = launch_swconv<RR, FF, WW, MM, KK>(a, grid, lds, sp, s);
#define CG_DISPATCH_RK(FF, WW, MM)                              \
  CG_DISPATCH(1, FF, WW, MM, 4) CG_DISPATCH(2, FF, WW, MM, 4) \
  CG_DISPATCH(1, FF, WW, MM, 2) CG_DISPATCH(2, FF, WW, MM, 2)
  if (false) {}
  CG_DISPATCH_RK(16, 1, 4) CG_DISPATCH_RK(16, 1, 1) CG_DISPATCH_RK(16, 1, 2)
  CG_DISPATCH_RK(32, 1, 2) CG_DISPATCH_RK(32, 1, 1)
  CG_DISPATCH_RK(32, 2, 4) CG_DISPATCH_RK(32, 2, 2) CG_DISPATCH_RK(16, 2, 8)
  CG_DISPATCH_RK(16, 2, 4)
#undef CG_DISPATCH_RK
#undef CG_DISPATCH
  g_split_prof.on = false;
  if (rc || a.ksplit == 1 || g_dry_run) return ssq_finish(rc);
  return split_finish_launch(d, a, s, false, split_timed ? split_stop : nullptr);
}

extern "C" long long cg_rowsumsq_ws_elems(const cg_conv_desc* d) {
  if (!d || d->tile < 0 || d->tile >= CG_NUM_TILES || d->Lu < 1 || d->nB < 1)
    return -1;
  if (!d->rowsumsq) return 0;
  const TileCfg tc = kTileCfgs[d->tile];
  const int TM = tc.swp_wm ? tc.swp_wm * tc.mt * 16 : (4 / tc.wgn) * tc.mt * tc.mf;
  const int TN = 64 * tc.wgn;
  if (d->Lu % TM) return -1;  // (rowsumsq needs one sample per tile)
  return (long long)d->nB * (d->Lu / TM) * d->nphase * ((d->N + TN - 1) / TN);
}
