// Software-pipelined tiles of the sliding-window convolution (cg_swconv,
// CG_TILE_SWP_*).
//
// Same contraction, operand packing and epilogues as swconv.hip; what differs is
// how a wave spends its time.  The tile kernels of swconv.hip run a serial
// chain per weight stage (wait DMA, barrier, read fragments, wait LDS, MFMA) at
// three waves per SIMD and rely on the co-resident workgroups to overlap each
// other; their register budget (<= 168) has no room for a second fragment set.
// Here a wave owns half a SIMD's registers (two waves per SIMD: one 8-wave
// workgroup, or two 4-wave workgroups, per CU) and pipelines itself:
//
//   * fragments are double-buffered at K-step granularity: the ds_reads of
//     K-step k+1 are issued BEFORE the MFMAs of K-step k, so LDS latency, the
//     per-stage barrier and the DMA waits sit in the shadow of 16-32 queued
//     MFMAs instead of in front of them;
//   * every global -> LDS byte moves by LDS-DMA (buffer_load_dwordx4 ... lds:
//     a 32-bit per-lane offset + a scalar offset, no address arithmetic in the
//     loop), the source window included: a window row of one 32-channel chunk is 64 B, a
//     DMA piece 16 rows; rows cannot be padded (the DMA writes lane-linear), so
//     the four 16-byte chunks of a row are XOR-swizzled by bit 2 of the row
//     (slot = chunk ^ 2*((row >> 2) & 1)): conflict-free ds_read_b128 for the
//     16x16x32 A operand at every tap offset.  Zero padding of 'same' and rows
//     past the batch are offsets past the descriptor's num_records (the load
//     returns zeros); the PhaseShuffle gather is the per-lane source offset;
//   * a stride-2 window is walked one source-row parity at a time (the packed
//     operand is parity-major), so stride 1 and stride 2 share one loop: a PASS
//     = (channel chunk, parity) = taps/stride taps over a (rows + taps/stride
//     - 1)-row window.  Windows are double-buffered across passes: the pieces of
//     pass p+1 are issued during the stages of pass p and retired by counted
//     vmcnt two stages later (never in the iteration that issues them);
//   * weight ring: 3 slots of 64-deep stages.  In the second half of stage s
//     (after the barrier that publishes stage s+1 and frees slot s) the DMA of
//     stage s+3 is issued into slot s; it has two whole stages to land;
//   * workgroups are persistent (one resident set walks all tiles, XCD residue
//     kept) and start in four phases a few thousand cycles apart, so that the
//     two workgroups of a CU are not at a tile boundary together; the epilogue
//     works in registers (lane swaps, no LDS), its bias comes from an LDS table
//     filled once per launch, and the next tile's id, scalars and window row
//     words are worked out between the MFMAs of the running tile's last pass.
//     Buffer resources stay in SCALAR registers (to_sgpr): built from a
//     vector-ALU quotient they would wrap every DMA in a waterfall loop.
#include <cstdlib>

#include "swconv_args.h"

namespace {

struct SwpArgs {
  ConvArgs c;
  int tpp;           // taps per pass (taps / stride)
  int WRs;           // window rows per segment: S + tpp - 1
  int wrows;         // nseg * WRs
  int npa;           // 1 KiB DMA pieces per window
  int apw;           // window pieces per wave per issuing stage
  int abytes;        // bytes per window buffer
  int npass;         // (nchunks - narrow) * stride + narrow
  int nst;           // weight stages per full pass (tpp / 2)
  int total_stages;
  int npad_rows;     // rows of the packed operand (N rounded up to 128)
  int ntl;           // linear tile ids: gm rounded up to 8, x gn x gp
  int bias_off;      // byte offset of the bias table in LDS (gn * TN floats)
  float inv_WRs;
  // stride 2: the order of a tile's full passes.  Two 32-channel chunks share
  // every 128-byte line of a source row.  0: (chunk, parity), parity innermost --
  // the second half of a line is fetched TWO passes after the first, ~4.4 MB of
  // window lines per XCD later: gone from the 4 MB L2, and the critic's first
  // layer read its source 1.85 x (profiles/r05_swconv_traffic_by_geometry.txt).
  // 1 (default): by LINE -- the pair of chunks that share a line, then parity, then
  // the chunk of the pair: (c0,p0) (c1,p0) (c0,p1) (c1,p1) (c2,p0) ... -- every
  // second half one pass (~2.2 MB) after the first, and the narrow last chunk
  // right behind the passes of the chunk it shares its line with.  The packed
  // operand keeps its layout ([chunk][parity][taps]); the weight stages' scalar
  // offset jumps at pass boundaries instead of running linearly.
  int chunk_inner;
};

constexpr int kSwpRing = 3;       // weight ring depth
constexpr int kSwpTapsPerPass = 12;  // 24-tap stride-2 / 12-tap stride-1 windows

// -DCG_SWP_TRACE (tools/swp_trace.sh; never in the product library): every wave
// adds up the shader-clock cycles it spends in each part of the tile loop and
// leaves them in g_swp_trace[workgroup][wave][part] (cg_debug_swp_trace reads
// them back).  A stamp is s_memtime + s_waitcnt lgkmcnt(0), placed only where
// the wave has no LDS read in flight.
#ifdef CG_SWP_TRACE
constexpr int kTraceParts = 10;
__device__ unsigned g_swp_trace[1024 * 8 * kTraceParts];
// (32-bit cycle counts: a wave lives well under 2^32 cycles, and ten 64-bit
// accumulators would push the kernel's scalar registers into spills)
#define CG_TR_DECL unsigned tr_[kTraceParts] = {}; unsigned tr_t = 0
#define CG_TR_START tr_t = tr_now()
#define CG_TR(part) do { const unsigned n_ = tr_now(); tr_[part] += n_ - tr_t; tr_t = n_; } while (0)
__device__ __forceinline__ unsigned tr_now() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return (unsigned)t;
}
#else
#define CG_TR_DECL
#define CG_TR_START
#define CG_TR(part)
#endif

// A wave-uniform value back in a scalar register.  The compiler computes integer
// divisions and 64-bit products on the vector ALU; a buffer resource built from
// such a value (a tile's sample index -> the window's base pointer) then sits in
// VECTOR registers and every DMA that uses it is wrapped in a waterfall loop
// (4 v_readfirstlane + 2 v_cmp + s_and_saveexec + branch -- round 2's kernels
// had one around each of their LDS-DMA issues).  readfirstlane of the quotient /
// of both pointer halves keeps the resource scalar.
__device__ __forceinline__ int to_sgpr(int v) {
  return __builtin_amdgcn_readfirstlane(v);
}

__device__ __forceinline__ int sw64(int byte) {
  // XOR-swizzle of a byte offset into a window of 64-byte rows: row bit 2
  // (address bit 8) flips chunk bit 1 (address bit 5)
  return byte ^ ((byte >> 3) & 32);
}

// Epilogue forms (template parameter EPI).  The generic one decides everything
// at run time (epilogue kind, f32 / split-K output, per-sample scale, penalty
// norm, output-side PhaseShuffle adjoint): ~700 instructions per wave and tile,
// issued at a quarter of the SIMD's rate while all four waves of the SIMD sit at
// the same tile boundary (profiles/r03_swp_wave_cycles.txt: 4.8 k of the 8.5 k
// boundary cycles).  The lean forms cover what the cfg2 step launches most,
// with everything else compiled out, wave-uniform row addressing on the scalar
// ALU (one 64-bit base per wave, 32-bit lane offsets, buffer stores) and no
// integer division:
//   kEpiLrelu      bias + max(v, alpha v) (alpha = 1: plain), bf16 rows
//   kEpiMask       bias, optional per-sample scale, LeakyReLU' mask (in place
//                  or not), bf16 rows
//   kEpiMaskShift  the same with the output-side PhaseShuffle adjoint
//                  (cg_conv_desc.out_shifts): rows land at their source positions
//   kEpiLreluSsq   kEpiLrelu + the penalty norm (cg_conv_desc.rowsumsq): the x^
//                  input gradient's launch (its own form: the scalars it adds
//                  cost the plain one registers inside the K loop)
constexpr int kEpiGeneric = 0, kEpiLrelu = 1, kEpiMask = 2, kEpiMaskShift = 3,
              kEpiLreluSsq = 4;

// R: source stride.  WM x WN waves (4 or 8); wave tile (16 * MT) x 64.
// LN: CG_EPI_LN_LRELU (LayerNorm + LeakyReLU in the epilogue; 128-column tiles).
template <int R, int WM, int WN, int MT, bool LN, bool NRW, int EPI>
__device__ __forceinline__ void swconv_swp_body(const SwpArgs& pa) {
  static_assert(WM * WN == 4 || WM * WN == 8, "one or two waves per SIMD");
  static_assert(!NRW || R == 2, "narrow last chunks exist for stride 2 only");
  static_assert(!LN || WN == 2, "the fused LayerNorm needs a 128-column tile");
  const ConvArgs& a = pa.c;
  constexpr int NW = WM * WN;
  constexpr int NT = 4;
  constexpr int KS = 2;                   // MFMA K-steps per weight stage
  constexpr int TM = WM * MT * 16;
  constexpr int TN = WN * 64;
  constexpr int kRowB = KS * 32;          // bf16 per weight row of a ring slot
  constexpr int kBufB = TN * kRowB;       // elements per ring slot
  constexpr int NBW = TN / 8 / NW;        // weight DMA pieces per wave per stage
  static_assert(NBW >= 1, "at least one weight piece per wave and stage");
  // window pieces per wave (windows of at most TM + 8 * 11 rows: nseg <= 8)
  constexpr int KPW = (TM / 16 + 6 + NW - 1) / NW;
  // one pass = TPP taps = NST 64-deep weight stages, fully unrolled below: tap
  // and ring-slot numbers are compile-time, so every fragment read is a
  // per-lane base register + an immediate offset (no address arithmetic in the
  // loop).  The window buffers have a compile-time stride for the same reason.
  constexpr int TPP = kSwpTapsPerPass;
  constexpr int NST = TPP / KS;
  static_assert(NST % kSwpRing == 0, "the ring slot of a stage is compile-time");
  // (windows of at most TM + 8 * 11 rows = TM / 16 + 6 pieces: nseg <= 8)
  constexpr int ABYTES = (TM / 16 + 6) * 1024;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* ldsA = smem;
  uint16_t* ldsB = reinterpret_cast<uint16_t*>(smem + 2 * ABYTES);
  // behind the ring: the rowsumsq slots of the waves (16 floats), for the fused
  // LayerNorm the row-statistics exchange table and gamma / beta, and last the
  // bias of every column tile (gn * TN floats, zero past N; read once per launch:
  // as a global load in the epilogue its round trip, 2 900 cycles, sat in
  // front of every tile's stores -- profiles/r03_swp_wave_cycles.txt).  Nothing of the epilogue
  // lives in the window / ring area: the next tile's DMAs land there meanwhile.
  float* wsum = reinterpret_cast<float*>(smem + 2 * ABYTES + kSwpRing * kBufB * 2);
  float* part = wsum + 16;                // [2][NW][16][2]
  float* lnp = part + 2 * NW * 32;        // gamma[128] | beta[128] (zero past N)
  // the 128-register stride-2 256 x 64 tile keeps the row words of its window
  // pieces in LDS, [piece slot][thread] (its K loop uses every register: left to
  // the compiler, they became scratch reloads behind s_waitcnt vmcnt(0), i.e. a
  // drain of the DMA pipeline, in the middle of a pass)
  constexpr bool ALDS = R == 2 && NW == 8 && WN == 1 && MT == 2;
  uint32_t* arow_lds = reinterpret_cast<uint32_t*>(wsum + 16);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  // (readfirstlane: the compiler then keeps everything derived from the wave
  // id in scalar registers and branches on it without exec masks)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN;
  const int wn = wave % WN;
  const int rM = lane & 15;
  const int g = lane >> 4;
  // split-K: this workgroup walks channel chunks [chunk0, chunk0 + nchunks) and
  // leaves f32 partial sums in its slice of the workspace (a.y; the host set
  // out_f32 and no epilogue: cg_swconv's finishing launch applies them)
  const int zsplit = a.ksplit > 1 ? (int)blockIdx.y : 0;
  const int chunk0 = zsplit * a.nchunks;
  const int gnp = a.gn * a.gp;
  const int full_passes = (a.nchunks - (NRW ? 1 : 0)) * R;
  const int nfull = a.nchunks - (NRW ? 1 : 0);  // full channel chunks of this walk
  // logical 16-byte chunk this lane fetches: pieces start at multiples of 16
  // rows, so row bit 2 is lane bit 4 for every piece
  const int aq = (lane & 3) ^ (((lane >> 4) & 1) << 1);

  if constexpr (LN) {
    if (tid < 128) {
      lnp[tid] = tid < a.N ? a.ln_gamma[tid] : 0.f;
      lnp[128 + tid] = tid < a.N ? a.ln_beta[tid] : 0.f;
    }
  }

  float* bias_lds = reinterpret_cast<float*>(smem + pa.bias_off);
  for (int i = tid; i < a.gn * TN; i += NW * 64)
    bias_lds[i] = (a.bias != nullptr && i < a.N) ? a.bias[i] : 0.f;
  // (published by the barriers of the first tile's K loop, long before the
  // first epilogue)

  // ---- tile walk ---------------------------------------------------------------
  // The workgroup is persistent: linear tile ids lin = blockIdx.x, + gridDim.x,
  // ... (gridDim.x is a multiple of 8, so a workgroup keeps its XCD residue).
  // XCD-aware mapping (as swconv_kernel): the workgroups that share one row
  // tile's source window get linear ids congruent mod 8.
  //
  // STREAM (-DCG_SWP_STREAM, off in the product library; instantiations without
  // a narrow pass): the tiles of a workgroup form ONE pass stream.  The last pass
  // of tile i issues the window of tile i + 1's first pass and its first three
  // weight stages exactly where a middle pass issues the next pass's -- into the
  // other window buffer and the rolling ring -- so the only thing left at a tile
  // boundary is the epilogue: no prologue DMAs queueing behind the epilogue's
  // stores, no vmcnt(0), no extra barrier.  (A narrow pass has 4 stages: the
  // ring slot of a stage would no longer be compile-time across tiles.)
  // Bit-exact on every kernel test, and 0.8 % SLOWER on the cfg2 step
  // (tools/ab_stream.sh, profiles/r03_swp_wave_cycles.txt): the boundary is the
  // epilogue's store burst -- 17 MB from all workgroups at once, 3.7 us -- and
  // vmcnt retires in order, so the first stage boundary of the next tile waits
  // for those stores whether the loads were prefetched or not.
#ifdef CG_SWP_STREAM
  constexpr bool STREAM = !NRW;
#else
  constexpr bool STREAM = false;
#endif
  auto tile_bm = [&](int lin) { return ((lin >> 3) / gnp) * 8 + (lin & 7); };
  auto next_tile = [&](int lin) {
    while (lin < pa.ntl && tile_bm(lin) >= a.gm) lin += (int)gridDim.x;
    return lin;
  };
  // what the K loop and the epilogue need of a tile (all wave-uniform)
  struct TileS {
    int m0, n0, y_off, b0, u00, off;
    int wtile;  // byte offset of the tile's (phase, first column) in the operand
    int ss;     // the tile's slot of the ordered penalty norm (a.ssq_ws)
  };
  auto tile_of = [&](int lin) {
    TileS t;
    const int jq = lin >> 3;
    const int jm = to_sgpr(jq / gnp);
    const int np_i = jq - jm * gnp;
    const int bm = jm * 8 + (lin & 7);
    const int phase = to_sgpr(np_i / a.gn);
    const int bn = np_i - phase * a.gn;
    t.off = a.off + phase * a.off_phase_step;
    t.y_off = a.y_off + phase * a.yoff_phase_step;
    t.m0 = bm * TM;
    t.n0 = bn * TN;
    t.b0 = to_sgpr(t.m0 / a.Lu);      // first sample of the tile
    t.u00 = t.m0 - t.b0 * a.Lu;       // its first output row (nseg == 1)
    t.wtile = (int)(((long long)phase * a.w_phase_stride +
                     (long long)t.n0 * a.Kpack) * 2);
    t.ss = ((t.u00 / TM) * a.gp + phase) * a.gn + bn;
    return t;
  };
  // per-tile state of the K loop
  int m0 = 0, n0 = 0, y_off = 0, b0 = 0, u00 = 0, sslot = 0;
  CG_TR_DECL;
  // source rows of this lane's window pieces, packed: bits [0, 14) the row of
  // source-row parity 0, [14, 28) of parity 1 (kRowPad = zero padding), [28, 31)
  // the sample of the tile (one register per piece instead of a byte offset per
  // parity: the 128-register tiles have none to spare)
  constexpr uint32_t kRowPad = 0x3fffu;
  uint32_t arow[KPW];
  int bstage = 0;
  int wbuf = 0;      // window buffer of the running pass
  __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t*>(a.x), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t*>(a.w), 0, 0x7fffffff, 0x00020000);
  // weight stages: wave w issues pieces w*NBW + i of a stage: lane L lands at
  // slot byte piece*1024 + L*16 = (row, chunk slot c') and fetches chunk
  // c' ^ swz(row) of that row of the tile (tile, phase and stage are the scalar
  // offset: these per-lane offsets hold for the whole launch; columns past N
  // stay inside the operand, which is padded to 128 rows)
  int boff[NBW];
#pragma unroll
  for (int i = 0; i < NBW; ++i) {
    const int pe = ((wave * NBW + i) * 1024 + lane * 16) / 2;  // element offset
    const int row = pe / kRowB;
    const int cs = (pe % kRowB) / 8;
    const int c = cs ^ ((row >> 1) & 7);
    boff[i] = (int)(((long long)row * a.Kpack + c * 8) * 2);
  }

  // Window pieces: piece j of a pass's window -> bytes [j KiB, (j+1) KiB) of its
  // buffer: lane L lands on (row 16 j + L/4, slot L & 3) and fetches the chunk
  // that slot holds after the swizzle.  Wave w owns pieces j = k NW + w.  The
  // packed source rows of piece slot k for tile t (rows counted from the tile's
  // first sample; padding rows become an offset past num_records at issue time:
  // the buffer form of the DMA then fetches zeros, so there is no zero page and
  // no 64-bit select in the loop):
  auto row_word = [&](int k, int tb0, int tu00, int toff) {
    // (lane ids re-derived from an opaque copy of the thread id: nothing of this
    // block then stays in registers across the K loop)
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    tb0 = to_sgpr(tb0);
    tu00 = to_sgpr(tu00);
    toff = to_sgpr(toff);
    const int row = (k * NW + wave) * 16 + (lane >> 2);
    int seg = 0;
    if (a.nseg > 1) seg = __float2int_rz(((float)row + 0.5f) * pa.inv_WRs);
    const int wr = row - seg * pa.WRs;
    const int b = tb0 + seg;
    const int u0 = a.nseg > 1 ? 0 : tu00;
    // per-sample phase shift of the lane's segment: scalar loads (a vector load
    // here would put an s_waitcnt vmcnt(0) -- every DMA in flight -- in the loop)
    int sft = 0;
    if (a.shifts != nullptr) {
      for (int q = 0; q < a.nseg; ++q) {
        const int bq = tb0 + q;
        const int sq = bq < a.nB ? a.shifts[bq / a.seg_size] : 0;
        sft = seg == q ? sq : sft;
      }
    }
    uint32_t w = (uint32_t)seg << 28;
#pragma unroll
    for (int par = 0; par < R; ++par) {
      uint32_t rr = kRowPad;
      if (row < pa.wrows && b < a.nB) {
        int srow = R * (u0 + wr) + toff + par;
        if (srow >= 0 && srow < a.Lx) {
          if (a.shifts) srow = shuffle_src(srow, sft, a.Lx);
          rr = (uint32_t)srow;
        }
      }
      w |= rr << (14 * par);
    }
    return w;
  };
  // (window rows are addressed from the tile's first sample)
  auto x_rsrc = [&](int tb0) {
    // (the 64-bit product is computed on the vector ALU: both halves of the
    // pointer go back to scalar registers)
    const unsigned long long p = reinterpret_cast<unsigned long long>(
        a.x + (long long)tb0 * a.Lx * a.Cx);
    const unsigned lo = (unsigned)to_sgpr((int)(unsigned)p);
    const unsigned hi = (unsigned)to_sgpr((int)(unsigned)(p >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<uint16_t*>(((unsigned long long)hi << 32) | lo), 0,
        0x7fffffff, 0x00020000);
  };
  // Everything the K loop needs of tile t (the row words of its window pieces
  // are computed ONCE per tile; the hot loop only passes the chunk offset as the
  // scalar operand).
  auto store_word = [&](auto k_tag, int tb0, int tu00, int toff) {
    constexpr int K = decltype(k_tag)::value;
    if constexpr (K < KPW) {
      const uint32_t w = row_word(K, tb0, tu00, toff);
      if constexpr (ALDS) arow_lds[K * (NW * 64) + (int)threadIdx.x] = w;
      else arow[K] = w;
    }
  };
  // (words_done: the last pass of the previous tile left this tile's row words)
  auto setup_tile = [&](const TileS& t, bool words_done) {
    m0 = t.m0;
    n0 = t.n0;
    y_off = t.y_off;
    b0 = t.b0;
    u00 = t.u00;
    sslot = t.ss;
    if (!words_done)
      static_for<KPW>([&](auto k_tag) { store_word(k_tag, t.b0, t.u00, t.off); });
    rx = x_rsrc(t.b0);
  };
  // piece slot k of a window: source rows w, channel chunk cc (narrow: the last
  // one), source-row parity par, into window buffer `buf`
  auto issue_a_piece = [&](__amdgpu_buffer_rsrc_t r, int buf, int cc, int par_full,
                           bool narrow_pass, int k, uint32_t w) {
    int par, add;
    uint32_t qb;
    bool pad = false;
    if (!NRW || !narrow_pass) {
      par = par_full;
      add = (chunk0 + cc) * 64;
      qb = (uint32_t)aq * 16;
    } else {
      // narrow last chunk: chunk q of a row = the first 8-channel group of
      // source-row parity q (q < 2; the rest of the row is never read)
      par = aq & 1;
      pad = aq >= 2;
      add = (chunk0 + a.nchunks - 1) * 64;
      qb = 0;
    }
    const uint32_t rr = (w >> (14 * par)) & kRowPad;
    // (24-bit multiply: sample * Lx + row < 2^17, row bytes < 2^12)
    uint32_t o = __umul24((w >> 28) * (uint32_t)a.Lx + rr, (uint32_t)a.Cx * 2) + qb;
    if (pad || rr == kRowPad) o = ~0u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(
        r,
        (__attribute__((address_space(3))) void*)(ldsA + buf * ABYTES +
                                                  (k * NW + wave) * 1024),
        16, (int)o, add, 0, 0);
  };
  // window piece of pass p of the running tile (p > 0 while its predecessor runs)
  auto issue_a_cur = [&](int p, int k, uint32_t w) {
    const bool np = NRW && p >= full_passes;
    int cc = p, par = 0;
    if constexpr (R == 2) {
      if (pa.chunk_inner) {
        if (p < 4 * (nfull >> 1)) {  // a whole pair of chunks: 4 passes
          cc = 2 * (p >> 2) + (p & 1);
          par = (p >> 1) & 1;
        } else {                     // the odd last full chunk: its two parities
          cc = nfull - 1;
          par = p - 4 * (nfull >> 1);
        }
      } else {
        cc = p >> 1;
        par = p & 1;
      }
    }
    issue_a_piece(rx, (wbuf ^ 1), cc, par, np, k, w);
  };
  // piece slot K (compile-time) of pass p; returns 1 if this wave owns it
  auto issue_a_slot = [&](int p, auto k_tag) {
    constexpr int K = decltype(k_tag)::value;
    if constexpr (K < KPW) {
      if ((K * NW + wave) < pa.npa) {
        if constexpr (ALDS)
          issue_a_cur(p, K, arow_lds[K * (NW * 64) + (int)threadIdx.x]);
        else
          issue_a_cur(p, K, arow[K]);
        return 1;
      }
    }
    return 0;
  };
  // the same slot of the NEXT tile's first pass (STREAM): its row word is
  // computed on the spot -- the words kept in registers / LDS are the running
  // tile's -- and it goes into the buffer the running (last) pass does not read
  auto issue_a_next = [&](int tb0, int tu00, int toff, auto k_tag) {
    constexpr int K = decltype(k_tag)::value;
    if constexpr (K < KPW) {
      if ((K * NW + wave) < pa.npa) {
        issue_a_piece(x_rsrc(tb0), (wbuf ^ 1), 0, 0, false, K,
                      row_word(K, tb0, tu00, toff));
        return 1;
      }
    }
    return 0;
  };
  auto issue_b = [&](int slot_idx) {
    uint16_t* slot = ldsB + slot_idx * kBufB;
#pragma unroll
    for (int i = 0; i < NBW; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          rw, (__attribute__((address_space(3))) void*)(slot + (wave * NBW + i) * 512),
          16, boff[i], bstage, 0, 0);
    bstage += KS * 32 * 2;
  };
  // (one running scalar offset: + one 64-deep stage per issue)
  auto first_stage_of = [&](int wtile) { return wtile + chunk0 * a.Fp * 16; };
  // window of pass 0 and weight stages 0..2 of the tile set up last (the first
  // tile of a workgroup; every tile of the kernels that keep the tile boundary)
  auto issue_prologue = [&](const TileS& t) {
    wbuf = 1;  // (the pieces go to the buffer "after" the running one: 0)
#pragma unroll
    for (int k = 0; k < KPW; ++k)
      if ((k * NW + wave) < pa.npa) {
        if constexpr (ALDS) issue_a_cur(0, k, arow_lds[k * (NW * 64) + (int)threadIdx.x]);
        else issue_a_cur(0, k, arow[k]);
      }
    wbuf = 0;
    bstage = first_stage_of(t.wtile);
    issue_b(0);
    if (pa.total_stages > 1) issue_b(1);
    if (pa.total_stages > 2) issue_b(2);
  };

  // ---- fragment addresses ----------------------------------------------------
  // A: window byte offset of (first tile row of this lane, k-group g) at tap t,
  // swizzled, one register per tap of a pass.  A wave's 16 MT rows lie in ONE
  // segment (S >= 16 MT, checked on the host), so subtile mt is +mt KiB and the
  // other window buffer +ABYTES: immediates (the swizzle moves bit 5 by bit 8)
  int rowb0;
  {
    const int i = wm * MT * 16 + rM;
    const int seg = i >> a.log2S;
    const int ui = i & (a.S - 1);
    rowb0 = (seg * pa.WRs + ui) * 64 + g * 16;
  }
  // (absolute LDS byte addresses: the reads are inline assembly)
  const int lds0 = (int)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  // (tap t + 8 sits 512 B behind tap t -- the swizzle only looks at address bits
  // 5 and 8 --, so eight registers serve the twelve taps of a pass: the last
  // four read through tap t - 8's base + 512 in the offset field)
  static_assert(TPP <= 16, "taps 8.. of a pass reuse the bases of taps 0..");
  constexpr int NAA = TPP < 8 ? TPP : 8;
  int aaddr[NAA];
#pragma unroll
  for (int t = 0; t < NAA; ++t) aaddr[t] = lds0 + sw64(rowb0 + t * 64);
  // narrow chunk: stage s -> parity s >> 1, K-step (s & 1) * 2 + ks; its four
  // k-groups are four consecutive taps of that parity over the same 8 channels
  // (chunk slot = parity); taps past taps/2 carry zero weights (row clamped: LDS
  // may hold anything finite)
  const int half_taps = a.taps >> 1;
  auto naddr = [&](int s, int ks, int buf_off) {
    // (recomputed per use: hoisted, the eight addresses of the narrow pass sat in
    // registers -- then in scratch -- across every full pass)
    int gq = g;
    asm volatile("" : "+v"(gq));
    int idx = 4 * ((s & 1) * 2 + ks) + gq;
    idx = idx < half_taps ? idx : half_taps - 1;
    return lds0 + buf_off + sw64(rowb0 - gq * 16 + idx * 64 + (s >> 1) * 16);
  };
  // B: byte offset of this lane's fragment row in ring slot 0 per K-step
  const int swzB = (rM >> 1) & 7;
  int vboff[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
    vboff[ks] = lds0 + 2 * ABYTES +
                ((wn * 64 + rM) * kRowB + (((4 * ks + g) ^ swzB) * 8)) * 2;
  // (slot, K-step, subtile and buffer are template arguments: immediates)
  auto read_b = [&](act8(&bf)[NT], auto slot_tag, auto ks_tag) {
    constexpr int SLOT = decltype(slot_tag)::value;
    constexpr int KSI = decltype(ks_tag)::value;
    static_for<NT>([&](auto nt) {
      lds_read128<SLOT * (kBufB * 2) + decltype(nt)::value * 16 * kRowB * 2>(
          bf[decltype(nt)::value], vboff[KSI]);
    });
  };
  auto read_a = [&](act8(&af)[MT], int addr, auto extra_tag) {
    static_for<MT>([&](auto mt) {
      lds_read128<decltype(mt)::value * 1024 + decltype(extra_tag)::value>(
          af[decltype(mt)::value], addr);
    });
  };
  using X0 = std::integral_constant<int, 0>;

  // The weights are the MFMA's A operand and the window its B operand: the
  // accumulator register r of lane (g, rM) is then output column 4 g + r of tile
  // row rM -- four CONSECUTIVE channels of one row per lane, which the epilogue
  // turns into 16-byte row-contiguous stores with lane swaps alone (no LDS).
  f32x4 acc[MT][NT];
  auto mfma_step = [&](const act8(&af)[MT], const act8(&bf)[NT]) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) mfma_acc(acc[mt][nt], bf[nt], af[mt]);
  };
  // counted wait at a stage boundary: all but this wave's newest DMAs (weight
  // stage gs + 2 if it exists, the previous stage's `na` window pieces) done
  auto wait_stage = [&](bool later_b, int na) {
    if (na == 0) {
      if (later_b) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBW) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (na == 1) {
      if (later_b) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBW + 1) : "memory");
      else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    } else {
      if (later_b) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBW + 2) : "memory");
      else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    }
  };
  using std::integral_constant;
  using I0 = integral_constant<int, 0>;
  using I1 = integral_constant<int, 1>;
  act8 af0[MT], bf0[NT], af1[MT], bf1[NT];

  // One pass, NSTG stages unrolled.  aaddr[] points into the window buffer of
  // the current pass (the buffers alternate: +-ABYTES on every tap address at
  // the end of a pass -- one body instead of one per buffer).
  // Per stage: [reads of K-step 1 | MFMAs of K-step 0] counted wait + barrier
  // (publishes stage gs + 1, frees slot gs and the other window buffer)
  // [reads of the next stage's K-step 0 | DMA issue | MFMAs of K-step 1].
  // (STREAM: the tile after the running one and its window resource; stream_next:
  // there is one, so the running tile's last pass behaves like a middle pass)
  // (plain wave-uniform ints, made scalar again at every use: as a struct behind
  // the lambdas' references they went through private memory and came back as
  // per-lane values -- a waterfall loop around every DMA)
  int n_m0 = 0, n_n0 = 0, n_yoff = 0, n_b0 = 0, n_u00 = 0, n_off = 0, n_wtile = 0;
  int n_ss = 0;
  bool stream_next = false;
  // The running tile, the one after it and whether there is one.  Kernels that
  // keep the tile boundary work the next tile out INSIDE the last pass of the
  // running one -- its id and scalars in stage 0, the row words of its window
  // pieces (the running tile's are dead by then) in the later stages -- where
  // the instructions issue between MFMAs the wave would wait for anyway; at the
  // boundary they were 200-odd instructions at a quarter of the issue rate in
  // front of the next tile's DMAs (profiles/r03_swp_wave_cycles.txt).
  int lin = 0, lin_next = 0;
  bool has_next = false;
  auto plan_next = [&]() {
    lin_next = next_tile(lin + (int)gridDim.x);
    has_next = lin_next < pa.ntl;
    if (has_next) {
      const TileS t = tile_of(lin_next);
      n_m0 = t.m0, n_n0 = t.n0, n_yoff = t.y_off, n_b0 = t.b0, n_u00 = t.u00;
      n_off = t.off, n_wtile = t.wtile, n_ss = t.ss;
    }
  };
  auto run_pass = [&](auto narrow_tag, int p) {
    constexpr bool NARROW = decltype(narrow_tag)::value;
    // only in the final pass of the tile can a later weight stage be missing
    // (every other pass is followed by at least 4 stages) -- and with a next
    // tile in the stream not even there
    const bool last = p + 1 == pa.npass;
    const bool ends = last && !(STREAM && stream_next);
    constexpr int NSTG = NARROW ? 4 : NST;
    const bool next_narrow = NRW && p + 1 >= full_passes;  // (the pass after, if any)
    const int cur_off = wbuf * ABYTES;
    const int delta = wbuf ? -ABYTES : ABYTES;
    int na_prev = 0;
    static_for<NSTG>([&](auto s_tag) {
      constexpr int s = decltype(s_tag)::value;
      using SLOT = integral_constant<int, s % kSwpRing>;
      using SLOT1 = integral_constant<int, (s + 1) % kSwpRing>;
      // ---- first half -------------------------------------------------------
      read_b(bf1, SLOT{}, I1{});
      {
        constexpr int T = (s * KS + 1) % TPP;
        if constexpr (NARROW)
          read_a(af1, naddr(s, 1, cur_off), X0{});
        else
          read_a(af1, aaddr[T % NAA], integral_constant<int, (T / NAA) * 512>{});
      }
      mfma_step(af0, bf0);
      lds_wait();
      CG_TR(0);  // first half: reads of K-step 1, MFMAs of K-step 0
      // ---- stage boundary ---------------------------------------------------
      // (window pieces are only issued in stages 0 .. NSTG - 3: none can be in
      // flight at the boundaries of stage 0 and of the last stage)
      if constexpr (s == 0 || s == NSTG - 1 || NARROW)
        wait_stage(!ends || s + 2 < NSTG, 0);
      else
        wait_stage(!ends || s + 2 < NSTG, na_prev);
      CG_TR(1);  // counted vmcnt wait
      __builtin_amdgcn_s_barrier();
      CG_TR(2);  // barrier
      // ---- second half ------------------------------------------------------
      if constexpr (s + 1 < NSTG) {
        read_b(bf0, SLOT1{}, I0{});
        constexpr int T = ((s + 1) * KS) % TPP;
        if constexpr (NARROW)
          read_a(af0, naddr(s + 1, 0, cur_off), X0{});
        else
          read_a(af0, aaddr[T % NAA], integral_constant<int, (T / NAA) * 512>{});
      } else {
        // first K-step of the next pass (other window buffer, ring slot 0);
        // after the last pass a harmless read of resident LDS
        read_b(bf0, I0{}, I0{});
        read_a(af0, next_narrow ? naddr(0, 0, cur_off + delta) : aaddr[0] + delta, X0{});
      }
      // (the DMA issue sits behind the reads: in front of them the LDS pipe and
      // the matrix pipe both idle while the wave builds addresses)
      if constexpr (STREAM && s == NSTG - 3) {
        // the stream's stage s + 3 is the next tile's first
        if (last && stream_next)
          bstage = first_stage_of(to_sgpr(n_wtile));
      }
      if constexpr (R == 2 && !NARROW && s == NSTG - 3) {
        // this issue is the next pass's first stage: with the chunk innermost the
        // operand ([chunk][parity][taps]) is not walked linearly
        if (pa.chunk_inner && p < 4 * (nfull >> 1)) {
          // (chunk, parity) sits at (2 chunk + parity) PB; inside a pair the walk
          // is +0, +2, +1, +3 PB: jumps +1, -2, +1, 0 behind the linear advance
          constexpr int PB = NST * KS * 32 * 2;  // bytes of one (chunk, parity)
          const int r = p & 3;
          bstage += r == 3 ? 0 : (r == 1 ? -2 * PB : PB);
        }
      }
      if (!ends || s + 3 < NSTG) issue_b(s % kSwpRing);
      mfma_step(af1, bf1);
      lds_wait();
      CG_TR(3);  // second half: reads, weight DMA issue, MFMAs of K-step 1
      // window of the next pass: stages 0 .. NSTG - 3 of this pass, so the wait
      // of stage NSTG - 1 (which leaves only the previous stage's pieces in
      // flight) retires all of them before the first read.  (Behind the MFMA
      // block: the row word may come from LDS, and no fragment read is pending.)
      na_prev = 0;
      if constexpr (!NARROW && s <= NSTG - 3) {
        if (!last) {
          // apw (1 or 2) slots of this wave per issuing stage
          if (pa.apw == 1) {
            na_prev = issue_a_slot(p + 1, integral_constant<int, s>{});
          } else {
            na_prev = issue_a_slot(p + 1, integral_constant<int, 2 * s>{}) +
                      issue_a_slot(p + 1, integral_constant<int, 2 * s + 1>{});
          }
        } else if (STREAM && stream_next) {
          if (pa.apw == 1) {
            na_prev = issue_a_next(n_b0, n_u00, n_off, integral_constant<int, s>{});
          } else {
            na_prev =
                issue_a_next(n_b0, n_u00, n_off, integral_constant<int, 2 * s>{}) +
                issue_a_next(n_b0, n_u00, n_off, integral_constant<int, 2 * s + 1>{});
          }
        }
      }
      if constexpr (!STREAM) {
        if (last) {
          // words per stage: stages 1 .. NSTG - 1 share the KPW of them
          constexpr int WPS = (KPW + NSTG - 2) / (NSTG - 1);
          if constexpr (s == 0) {
            plan_next();
          } else {
            if (has_next) {
              static_for<WPS>([&](auto j_tag) {
                store_word(integral_constant<int, (s - 1) * WPS + decltype(j_tag)::value>{},
                           n_b0, n_u00, n_off);
              });
            }
          }
        }
      }
      CG_TR(4);  // window DMA issue (+ the next tile's row words in a last pass)
    });
#pragma unroll
    for (int t = 0; t < NAA; ++t) aaddr[t] += delta;
    wbuf ^= 1;
  };
  using False = integral_constant<bool, false>;
  using True = integral_constant<bool, true>;

  // ---- epilogue: accumulators -> lane swaps -> 16-byte row-contiguous stores ---
  // Lane (g, rM) holds, per 16 x 16 block nt, columns 4 g .. 4 g + 3 of tile row
  // rM.  v_permlane16_swap exchanges the odd 16-lane rows of block 2 p with the
  // even rows of block 2 p + 1: afterwards lane (g, rM) owns EIGHT consecutive
  // columns of row rM -- block 2 p + (g & 1), columns 8 (g >> 1) .. + 7 -- i.e.
  // one 16-byte store per lane and block pair, 64 contiguous bytes per row and
  // instruction.  No LDS round trip, no barrier.
  auto pair8 = [&](int mt, int p, float (&v)[8]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const auto sw = __builtin_amdgcn_permlane16_swap(
          __float_as_uint(acc[mt][2 * p][r]), __float_as_uint(acc[mt][2 * p + 1][r]),
          false, false);
      v[r] = __uint_as_float(sw[0]);
      v[4 + r] = __uint_as_float(sw[1]);
    }
  };
  auto epilogue = [&](int em0, int en0, int ey_off, int eb0, int ess) {
    // the matrix pipe retires the last MFMAs (inline assembly: the compiler's
    // hazard recognizer does not see them) before ordinary instructions read acc
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) asm volatile("" : "+v"(acc[mt][nt]));
    // (lane ids re-derived from an opaque copy of the thread id, as in
    // setup_tile: the epilogue keeps no register alive across the K loop)
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN;
    const int wn = wave % WN;
    const int rM = lane & 15;
    const int g = lane >> 4;
    const int cq = (g & 1) * 16 + (g >> 1) * 8;  // lane's column inside a block pair
    const int nl0 = en0 + wn * 64 + cq;  // lane's first column of block pair 0
    float bv[2][8];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const f32x4 b0v = *reinterpret_cast<const f32x4*>(bias_lds + nl0 + p * 32);
      const f32x4 b1v = *reinterpret_cast<const f32x4*>(bias_lds + nl0 + p * 32 + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        bv[p][e] = b0v[e];
        bv[p][4 + e] = b1v[e];
      }
    }
    const int mw0 = em0 + wm * MT * 16;  // first row of this wave
#ifdef CG_SWP_TRACE
    for (int p = 0; p < 2; ++p)
      for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(bv[p][e]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    CG_TR(9);  // epilogue: until the bias values have arrived
#endif
    if constexpr (LN) {
      // fused LayerNorm + LeakyReLU: a row's statistics span the two waves that
      // share its row block (wave ^ 1); the partial sums meet in a small LDS
      // table (double-buffered by subtile: one workgroup barrier per subtile)
      const float invn = 1.f / (float)a.N;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        float v[2][8];
        pair8(mt, 0, v[0]);
        pair8(mt, 1, v[1]);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int n = nl0 + p * 32 + e;
            // statistics of the STORED pre-activation, as the separate
            // cg_ln_lrelu_fwd pass sees it
            v[p][e] = n < a.N ? act2f(f2act(v[p][e] + bv[p][e])) : 0.f;
            s1 += v[p][e];
            s2 += v[p][e] * v[p][e];
          }
        s1 += __shfl_xor(s1, 16, 64);
        s2 += __shfl_xor(s2, 16, 64);
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 32, 64);
        float* pt = part + (mt & 1) * (NW * 32);
        if (g == 0)
          *reinterpret_cast<float2*>(pt + (wave * 16 + rM) * 2) = make_float2(s1, s2);
        __syncthreads();
        const float2 o2 =
            *reinterpret_cast<const float2*>(pt + ((wave ^ 1) * 16 + rM) * 2);
        const float mean = (s1 + o2.x) * invn;
        const float var = fmaxf((s2 + o2.y) * invn - mean * mean, 0.f);
        const float rstd = rsqrtf(var + a.ln_eps);
        const int m = mw0 + mt * 16 + rM;
        if (m < a.M) {
          const int b = m / a.Lu;
          const int u = m - b * a.Lu;
          const long long ridx =
              (long long)b * a.Ly + (long long)a.y_stride * u + ey_off;
          const long long rowoff = ridx * a.Cy;
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            const int n = nl0 + p * 32;
            if (n < a.Cy) {
              const int lc = n - en0;  // (en0 == 0: one column tile)
              const f32x4 g0 = *reinterpret_cast<const f32x4*>(lnp + lc);
              const f32x4 g1 = *reinterpret_cast<const f32x4*>(lnp + lc + 4);
              const f32x4 b0v = *reinterpret_cast<const f32x4*>(lnp + 128 + lc);
              const f32x4 b1v = *reinterpret_cast<const f32x4*>(lnp + 128 + lc + 4);
              float hv[8];
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                const float t =
                    (v[p][e] - mean) * rstd * (e < 4 ? g0[e] : g1[e - 4]) +
                    (e < 4 ? b0v[e] : b1v[e - 4]);
                hv[e] = fmaxf(t, a.alpha * t);
              }
              // (forward-only callers pass no statistics buffers: the
              // pre-activation is then not stored either)
              if (a.ln_mean)
                *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(a.y) +
                                          rowoff + n) =
                    make_uint4(pack2act(v[p][0], v[p][1]), pack2act(v[p][2], v[p][3]),
                               pack2act(v[p][4], v[p][5]), pack2act(v[p][6], v[p][7]));
              *reinterpret_cast<uint4*>(a.ln_h + rowoff + n) =
                  make_uint4(pack2act(hv[0], hv[1]), pack2act(hv[2], hv[3]),
                             pack2act(hv[4], hv[5]), pack2act(hv[6], hv[7]));
            }
          }
          if (a.ln_mean && wn == 0 && g == 0) {
            a.ln_mean[ridx] = mean;
            a.ln_rstd[ridx] = rstd;
          }
        }
      }
    } else {
      // Row addressing.  A wave's 16 MT rows lie in one sample (S >= 16 MT):
      // sample index and output-side phase shift are wave-uniform.
      const int bw = __builtin_amdgcn_readfirstlane(mw0 / a.Lu);
      const int uw0 = mw0 - bw * a.Lu;
      int oshift = 0;
      if (a.out_shifts && bw < a.nB) oshift = a.out_shifts[bw / a.out_seg];
      // per-sample scale in front of the epilogue (wave-uniform: a scalar load)
      float rs = 1.f;
      if (a.row_scale && bw < a.nB) rs = a.row_scale[bw];
      // output row of tile row r of this wave: element offset of its first
      // column; to_side: a reflected row of the output-side PhaseShuffle adjoint
      auto row_target = [&](int r, bool& to_side) {
        const int u = uw0 + r;
        int t = a.y_stride * u + ey_off;
        to_side = false;
        if (a.out_shifts) {
          if (oshift > 0) {
            to_side = t >= a.Ly - oshift;
            t = to_side ? t - (a.Ly - oshift) : t + oshift;
          } else {
            to_side = t < -oshift;
            t = to_side ? t : t + oshift;
          }
        }
        return ((long long)bw * (to_side ? a.side_rows : a.Ly) + t) * a.Cy;
      };
      // the LeakyReLU' mask words of every subtile are fetched up front (one
      // exposed round trip instead of one per subtile)
      uint4 mk[MT][2];
      if (a.epilogue == CG_EPI_MASK) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const int r = mt * 16 + rM;
          bool to_side;
          const long long rowoff = row_target(r, to_side);
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            mk[mt][p] = make_uint4(0u, 0u, 0u, 0u);
            if (mw0 + r < a.M && nl0 + p * 32 < a.Cy && !to_side)
              mk[mt][p] = *reinterpret_cast<const uint4*>(a.mask + rowoff + nl0 +
                                                          p * 32);
          }
        }
      }
      float ssq = 0.f;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int r = mt * 16 + rM;
        bool to_side;
        const long long rowoff = row_target(r, to_side);
        const bool row_ok = mw0 + r < a.M;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          float v[8];
          pair8(mt, p, v);  // (every lane takes part in the swap)
          const int n = nl0 + p * 32;
          if (row_ok && n < a.Cy) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (v[e] + bv[p][e]) * rs;
            if (a.epilogue == CG_EPI_LRELU) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], a.alpha * v[e]);
            } else if (a.epilogue == CG_EPI_MASK && !to_side) {
              if (a.out_shifts) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = act2f(f2act(v[e]));
              }
              const uint4 h4 = mk[mt][p];
              const uint32_t hw[4] = {h4.x, h4.y, h4.z, h4.w};
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                const uint16_t hv = (uint16_t)(hw[e >> 1] >> ((e & 1) * 16));
                v[e] *= (act2f(hv) > 0.f) ? 1.f : a.alpha;
              }
            } else if (a.epilogue == CG_EPI_SIGMOID) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = 1.f / (1.f + __expf(-v[e]));
            }
            // (columns past N are stored as zeros; only the last column tile of
            // an N that is no multiple of the tile has any: a scalar branch
            // instead of 16 compares and selects per store)
            if (en0 + TN > a.N) {
#pragma unroll
              for (int e = 0; e < 8; ++e)
                if (n + e >= a.N) v[e] = 0.f;
            }
            if (a.rowsumsq) {
#pragma unroll
              for (int e = 0; e < 8; ++e) ssq += v[e] * v[e];
            }
            if (a.out_f32) {
              float* dst = reinterpret_cast<float*>(a.y) + zsplit * a.split_stride +
                           rowoff + n;
              *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
              *reinterpret_cast<f32x4*>(dst + 4) = f32x4{v[4], v[5], v[6], v[7]};
            } else {
              uint16_t* dst =
                  (to_side ? a.side : reinterpret_cast<uint16_t*>(a.y)) + rowoff + n;
              *reinterpret_cast<uint4*>(dst) =
                  make_uint4(pack2act(v[0], v[1]), pack2act(v[2], v[3]),
                             pack2act(v[4], v[5]), pack2act(v[6], v[7]));
            }
          }
        }
      }
      if (a.rowsumsq) {
        // the whole tile belongs to one sample (nseg == 1, checked on the host):
        // one f32 atomic per workgroup
        ssq = wave_sum(ssq);
        if (lane == 0) wsum[wave] = ssq;
        __syncthreads();
        if (tid == 0 && em0 < a.M) {
          float t = 0.f;
          for (int w = 0; w < NW; ++w) t += wsum[w];
          // (ordered form: the workgroup's own slot, summed by the finishing
          // launch; else one f32 atomic per workgroup)
          if (a.ssq_ws) a.ssq_ws[(long long)eb0 * a.ssq_P + ess] = t;
          else atomicAdd(a.rowsumsq + eb0, t);
        }
      }
    }
  };

  // ---- lean epilogues (EPI != kEpiGeneric; see the list above the template) -----
  // A wave's 16 MT rows lie in one sample (S >= 16 MT), so sample, first row and
  // the PhaseShuffle shift are wave-uniform: the row base is ONE 64-bit scalar
  // address per wave (shifts of the tile's own sample / row, no division), a lane
  // adds a 32-bit offset, and out-of-range lanes are switched off by an offset
  // past num_records (the buffer form of the store drops them).
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  // (lane offset of a switched-off lane: past num_records whatever the scalar
  // offset adds, and no 32-bit wrap)
  constexpr int kOff = (int)0x80000000u;
  auto rsrc_at = [&](const void* p0, long long byte_off) {
    const unsigned long long p = reinterpret_cast<unsigned long long>(p0) +
                                 (unsigned long long)byte_off;
    const unsigned lo = (unsigned)to_sgpr((int)(unsigned)p);
    const unsigned hi = (unsigned)to_sgpr((int)(unsigned)(p >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0, 0x7fffffff,
        0x00020000);
  };
  auto epilogue_lean = [&](int en0, int ey_off, int eb0, int eu00, int ess) {
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) asm volatile("" : "+v"(acc[mt][nt]));
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN;
    const int wn = wave % WN;
    const int rM = lane & 15;
    const int g = lane >> 4;
    const int cq = (g & 1) * 16 + (g >> 1) * 8;  // lane's column inside a block pair
    en0 = to_sgpr(en0);
    const int nl0 = en0 + wn * 64 + cq;          // lane's first column of block pair 0
    const int i0 = wm * MT * 16;                 // first tile row of this wave
    const int bw = to_sgpr(eb0) + (i0 >> a.log2S);
    if (bw >= a.nB) return;                      // rows past M: the whole wave
    const int uw0 = (a.nseg > 1 ? 0 : to_sgpr(eu00)) + (i0 & (a.S - 1));
    // position of the wave's first row inside its sample, in output rows
    const int t0 = a.y_stride * uw0 + to_sgpr(ey_off);
    const int rowB = a.Cy * 2;                   // bytes per output row
    const int mt_step = 16 * a.y_stride * rowB;  // bytes between the two subtiles
    const bool cols_open = en0 + TN > a.Cy;      // the column tile overhangs the pitch
    const bool zero_tail = en0 + TN > a.N;       // ... or the real channels
    float bv[2][8];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const f32x4 b0v = *reinterpret_cast<const f32x4*>(bias_lds + nl0 + p * 32);
      const f32x4 b1v = *reinterpret_cast<const f32x4*>(bias_lds + nl0 + p * 32 + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        bv[p][e] = b0v[e];
        bv[p][4 + e] = b1v[e];
      }
    }
    // (kEpiLrelu also carries the penalty norm: the f32 sum of squares of what
    // the wave stores, before rounding -- the x^ input gradient's launch)
    float ssq = 0.f;
    auto finish8 = [&](float (&v)[8], int n) {
      if (zero_tail) {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (n + e >= a.N) v[e] = 0.f;
      }
      if constexpr (EPI == kEpiLreluSsq) {
#pragma unroll
        for (int e = 0; e < 8; ++e) ssq += v[e] * v[e];
      }
      return u32x4{pack2act(v[0], v[1]), pack2act(v[2], v[3]), pack2act(v[4], v[5]),
                   pack2act(v[6], v[7])};
    };
    auto masked8 = [&](float (&v)[8], const u32x4 h4, bool round_first) {
      const uint32_t hw[4] = {h4.x, h4.y, h4.z, h4.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (round_first) v[e] = act2f(f2act(v[e]));
        const uint16_t hv = (uint16_t)(hw[e >> 1] >> ((e & 1) * 16));
        v[e] *= (act2f(hv) > 0.f) ? 1.f : a.alpha;
      }
    };
    // Subtile mt of a lane: + mt * mt_step bytes, added on the VECTOR side.  (As
    // the scalar offset of the store it corrupted data: with a register in the
    // soffset field hipcc assumes that a 16-byte store has read its data registers
    // at issue and lets the next instructions overwrite them -- the third store of
    // this epilogue was followed by writes to its data registers two instructions
    // later, and under back-pressure the last lanes the store reads, rows 12-15
    // of a 16-row block, saw the new values: about one element in 10^4 at the
    // benchmark's shapes, none in the small kernel tests.  With soffset 0 the
    // compiler keeps its wait states; tests/test_hip_fullsize.py holds the case.)
    auto store_rows = [&](const u32x4 d, __amdgpu_buffer_rsrc_t r, int vofs, int mt) {
      // (a switched-off lane stays past num_records: kOff + mt_step < 2^32)
      __builtin_amdgcn_raw_buffer_store_b128(d, r, vofs + mt * mt_step, 0, 0);
    };
    if constexpr (EPI == kEpiLrelu || EPI == kEpiLreluSsq || EPI == kEpiMask) {
      const long long base = (((long long)bw * a.Ly + t0) * a.Cy) * 2;
      const __amdgpu_buffer_rsrc_t ry = rsrc_at(a.y, base);
      const int voff = rM * a.y_stride * rowB + nl0 * 2;
      int vo[2];
#pragma unroll
      for (int p = 0; p < 2; ++p)
        vo[p] = (cols_open && nl0 + p * 32 >= a.Cy) ? kOff : voff + p * 64;
      u32x4 mk[MT][2];
      float rs = 1.f;
      if constexpr (EPI == kEpiMask) {
        const __amdgpu_buffer_rsrc_t rm = rsrc_at(a.mask, base);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int p = 0; p < 2; ++p)
            mk[mt][p] = __builtin_amdgcn_raw_buffer_load_b128(rm, vo[p], mt * mt_step, 0);
        if (a.row_scale) rs = a.row_scale[bw];
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          float v[8];
          pair8(mt, p, v);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += bv[p][e];
          if constexpr (EPI != kEpiMask) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], a.alpha * v[e]);
          } else {
            if (a.row_scale) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] *= rs;
            }
            masked8(v, mk[mt][p], false);
          }
          store_rows(finish8(v, nl0 + p * 32), ry, vo[p], mt);
        }
      if constexpr (EPI == kEpiLreluSsq) {
        // the whole tile belongs to one sample (nseg == 1, checked on the host;
        // every wave of the workgroup is here): the workgroup's own slot of the
        // ordered sum, or one f32 atomic
        ssq = wave_sum(ssq);
        if (lane == 0) wsum[wave] = ssq;
        __syncthreads();
        if (tid == 0) {
          float t = 0.f;
          for (int w = 0; w < NW; ++w) t += wsum[w];
          if (a.ssq_ws) a.ssq_ws[(long long)bw * a.ssq_P + to_sgpr(ess)] = t;
          else atomicAdd(a.rowsumsq + bw, t);
        }
      }
    } else {
      // kEpiMaskShift: row t of the sample lands on t + shift; the |shift| rows
      // that the reflection folds back go, unmasked, to the side buffer
      // (cg_unshuffle_fixup adds them in)
      int oshift = a.out_shifts[bw / a.out_seg];
      const int tlast = t0 + a.y_stride * (16 * MT - 1);
      const bool any_side = oshift > 0 ? tlast >= a.Ly - oshift : t0 < -oshift;
      if (!any_side) {
        const long long base = (((long long)bw * a.Ly + t0 + oshift) * a.Cy) * 2;
        const __amdgpu_buffer_rsrc_t ry = rsrc_at(a.y, base);
        const __amdgpu_buffer_rsrc_t rm = rsrc_at(a.mask, base);
        const int voff = rM * a.y_stride * rowB + nl0 * 2;
        int vo[2];
#pragma unroll
        for (int p = 0; p < 2; ++p)
          vo[p] = (cols_open && nl0 + p * 32 >= a.Cy) ? kOff : voff + p * 64;
        u32x4 mk[MT][2];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int p = 0; p < 2; ++p)
            mk[mt][p] = __builtin_amdgcn_raw_buffer_load_b128(rm, vo[p], mt * mt_step, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            float v[8];
            pair8(mt, p, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += bv[p][e];
            masked8(v, mk[mt][p], true);
            store_rows(finish8(v, nl0 + p * 32), ry, vo[p], mt);
          }
      } else {
        // the one or two waves per sample whose rows reach the reflected end:
        // per-lane targets, one store instruction per destination buffer
        const __amdgpu_buffer_rsrc_t ry =
            rsrc_at(a.y, ((long long)bw * a.Ly * a.Cy) * 2);
        const __amdgpu_buffer_rsrc_t rm =
            rsrc_at(a.mask, ((long long)bw * a.Ly * a.Cy) * 2);
        const __amdgpu_buffer_rsrc_t rsd =
            rsrc_at(a.side, ((long long)bw * a.side_rows * a.Cy) * 2);
        int vy[MT][2], vs[MT][2];
        u32x4 mk[MT][2];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          int t = t0 + a.y_stride * (mt * 16 + rM);
          bool to_side;
          if (oshift > 0) {
            to_side = t >= a.Ly - oshift;
            t = to_side ? t - (a.Ly - oshift) : t + oshift;
          } else {
            to_side = t < -oshift;
            t = to_side ? t : t + oshift;
          }
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            const bool ok = !(cols_open && nl0 + p * 32 >= a.Cy);
            const int o = t * rowB + (nl0 + p * 32) * 2;
            vy[mt][p] = (ok && !to_side) ? o : kOff;
            vs[mt][p] = (ok && to_side) ? o : kOff;
            mk[mt][p] = __builtin_amdgcn_raw_buffer_load_b128(rm, vy[mt][p], 0, 0);
          }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            float v[8], vm[8];
            pair8(mt, p, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              v[e] += bv[p][e];
              vm[e] = v[e];
            }
            masked8(vm, mk[mt][p], true);
            __builtin_amdgcn_raw_buffer_store_b128(finish8(vm, nl0 + p * 32), ry,
                                                   vy[mt][p], 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(finish8(v, nl0 + p * 32), rsd,
                                                   vs[mt][p], 0, 0);
          }
      }
    }
  };

  // ---- the tile loop -----------------------------------------------------------
  lin = next_tile((int)blockIdx.x);
  if (lin >= pa.ntl) return;
#ifndef CG_SWP_NO_STAGGER
  // Start stagger.  All resident workgroups run equal tiles in lockstep, so every
  // tile boundary is a chip-wide burst (the epilogue's stores: 17 MB in 3.7 us,
  // profiles/r03_swp_wave_cycles.txt) during which no MFMA issues.  Four start
  // phases 3 200 cycles apart -- the two workgroups that share a CU (ids 256
  // apart) and neighbouring ids (bit 0) -- spread each burst while the other
  // workgroup of the CU computes; the price is the same delay once at the end
  // of the launch.  Measured on the cfg2 step (tools/ab_stagger.sh, three boxes):
  // -1.0 ... -1.7 %; less at 1 920 or 5 120 cycles per phase, a loss with eight
  // phases or with phases of a quarter tile, and nothing when only the
  // multi-tile or only the single-tile launches are staggered.
#ifndef CG_SWP_STAGGER
#define CG_SWP_STAGGER (((bx & 1) + 2 * ((bx >> 8) & 1)) * 5)
#endif
  {
    const int bx = (int)blockIdx.x;
    const int st = CG_SWP_STAGGER;  // units of s_sleep 10 = 640 cycles
    for (int i = 0; i < st; ++i) __builtin_amdgcn_s_sleep(10);
  }
#endif
  CG_TR_START;
  {
    const TileS t0 = tile_of(lin);
    setup_tile(t0, false);
    issue_prologue(t0);
  }
  CG_TR(7);  // tile set-up + prologue DMA issue
  bool landed = false;  // the tile's first window and weight stages are published
  while (true) {
    if constexpr (STREAM) {
      plan_next();
      stream_next = has_next;
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!landed) {
      // window of pass 0 and weight stages 0..2 have landed (and the stores of
      // the previous tile's epilogue have been taken)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      CG_TR(5);  // prologue wait: first window + three weight stages (+ old stores)
      __builtin_amdgcn_s_barrier();
    }
    read_b(bf0, I0{}, I0{});
    read_a(af0, (!NRW || full_passes > 0) ? aaddr[0] : naddr(0, 0, 0), X0{});
    lds_wait();
    CG_TR(6);  // prologue barrier + first fragment reads
    for (int p = 0; p < full_passes; ++p) run_pass(False{}, p);
    if constexpr (NRW) run_pass(True{}, full_passes);
    const int em0 = m0, en0 = n0, ey_off = y_off, eb0 = b0, eu00 = u00;
    if constexpr (EPI == kEpiGeneric || LN) epilogue(em0, en0, ey_off, eb0, sslot);
    else epilogue_lean(en0, ey_off, eb0, eu00, sslot);
    CG_TR(8);  // epilogue
    lin = lin_next;
    if (lin >= pa.ntl) break;
    if constexpr (STREAM) {
      // the stream already holds this tile's first window and weight stages
      // (published by the last stage boundary of the pass just run)
      TileS t;
      t.m0 = n_m0, t.n0 = n_n0, t.y_off = n_yoff, t.b0 = n_b0, t.u00 = n_u00;
      t.off = n_off, t.wtile = n_wtile, t.ss = n_ss;
      setup_tile(t, false);
      landed = true;
    } else {
      // (the tap addresses toggle window buffers per pass: back to buffer 0)
      if (wbuf) {
#pragma unroll
        for (int t = 0; t < NAA; ++t) aaddr[t] -= ABYTES;
        wbuf = 0;
      }
      // its stores still in flight, the DMAs of the next tile's prologue (every
      // wave is past the last LDS read of the K loop behind this barrier)
      __builtin_amdgcn_s_barrier();
      TileS t;
      t.m0 = n_m0, t.n0 = n_n0, t.y_off = n_yoff, t.b0 = n_b0, t.u00 = n_u00;
      t.off = n_off, t.wtile = n_wtile, t.ss = n_ss;
      setup_tile(t, true);
      issue_prologue(t);
    }
    CG_TR(7);
  }
#ifdef CG_SWP_TRACE
  if ((threadIdx.x & 63) == 0 && blockIdx.x < 1024 && blockIdx.y == 0) {
    for (int k = 0; k < kTraceParts; ++k)
      g_swp_trace[((int)blockIdx.x * 8 + (int)(threadIdx.x >> 6)) * kTraceParts + k] = tr_[k];
  }
#endif
}

// (the body is a __device__ function: with the buffer-resource builtins written
// directly in a __global__ template the host pass emits no stub for it)
// (second launch bound = waves per SIMD the register budget must allow: the
// 32-row wave tiles run 3 four-wave or 2 eight-wave workgroups per CU)
template <int R, int WM, int WN, int MT, bool LN = false, bool NRW = false,
          int EPI = kEpiGeneric>
__global__ __launch_bounds__(WM* WN * 64, MT == 2 ? (WM * WN == 8 ? 4 : 3) : 2) void
swconv_swp_kernel(SwpArgs pa) {
  swconv_swp_body<R, WM, WN, MT, LN, NRW, EPI>(pa);
}

// Workgroups of one instantiation a CU holds at `lds` bytes of dynamic LDS
// (registers, LDS and the wave slots together; asked of the runtime once).
template <int R, int WM, int WN, int MT, bool LN, bool NRW, int EPI>
int swp_occupancy(size_t lds) {
  // (per device; the LDS size of an instantiation can differ between launches:
  // the last answer is kept with the size it was asked for, packed in one word)
  static std::atomic<unsigned long long> known[kCgMaxDevices];
  const int dev = cg_device_index();
  const unsigned long long k = known[dev].load(std::memory_order_acquire);
  if ((k >> 8) == (unsigned long long)lds + 1) return (int)(k & 0xff);
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(
          &nb,
          reinterpret_cast<const void*>(
              &swconv_swp_kernel<R, WM, WN, MT, LN, NRW, EPI>),
          WM * WN * 64, lds) != hipSuccess || nb < 1)
    nb = 1;
  if (nb > 255) nb = 255;
  known[dev].store((((unsigned long long)lds + 1) << 8) | (unsigned)nb,
                   std::memory_order_release);
  return nb;
}

inline int swp_num_cus() {
  static std::atomic<int> cus[kCgMaxDevices];
  const int dev = cg_device_index();
  int c = cus[dev].load(std::memory_order_acquire);
  if (c == 0) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess ||
        prop.multiProcessorCount < 1)
      return 256;
    c = prop.multiProcessorCount;
    cus[dev].store(c, std::memory_order_release);
  }
  return c;
}

template <int R, int WM, int WN, int MT, bool LN = false, bool NRW = false,
          int EPI = kEpiGeneric>
int launch_swp(const SwpArgs& pa, unsigned gy, size_t lds, bool dry, hipStream_t s) {
  if (dry) return 0;
  static CgPerDeviceFlag attr_set;
  if (!attr_set.test()) {
    hipError_t e = hipFuncSetAttribute(
        reinterpret_cast<const void*>(
            &swconv_swp_kernel<R, WM, WN, MT, LN, NRW, EPI>),
        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set.mark();
  }
  // persistent workgroups: one resident set walks all tiles (a multiple of 8
  // workgroups, so each keeps its XCD residue over its tiles)
  long long slots =
      (long long)swp_num_cus() * swp_occupancy<R, WM, WN, MT, LN, NRW, EPI>(lds);
  slots = slots / 8 * 8;
  if (gy > 1) slots = (slots / gy) / 8 * 8;  // the split-K slices share the CUs
  if (slots < 8) slots = 8;
  const unsigned gx = (unsigned)(pa.ntl < slots ? pa.ntl : slots);
  CG_LAUNCH_PROF(CG_FAMILY_SWCONV,
                 (swconv_swp_kernel<R, WM, WN, MT, LN, NRW, EPI>), dim3(gx, gy),
                 dim3(WM * WN * 64), lds, s, pa);
  CG_LAUNCH_CHECK();
}

}  // namespace

#ifdef CG_SWP_TRACE
extern "C" int cg_debug_swp_trace(unsigned* dst, int n) {
  if (n > 1024 * 8 * kTraceParts) return CG_EINVAL;
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_swp_trace),
                                  (size_t)n * sizeof(unsigned));
}
#endif

// The specialised epilogues are OFF unless CALCIUMGAN_SWP_LEAN_EPI=1 (read once)
// or cg_debug_lean_epilogue(1): measured on the cfg2 step they cut a wave's
// boundary from ~8.5 k to ~2.6 k cycles per tile (profiles/r04_swp_wave_cycles.txt)
// and the kernels take the SAME time to 1 % on every geometry, the step 2 % MORE
// (profiles/r04_swp_lean_epilogue_ab.txt): the launches are not bound by what a
// wave does at a tile boundary.  Kept for the record and for the tests that pin
// the store hazard found on the way (store_rows).
static bool g_swp_lean_epi = [] {
  const char* e = getenv("CALCIUMGAN_SWP_LEAN_EPI");
  return e && e[0] == '1';
}();
extern "C" int cg_debug_lean_epilogue(int on) {
  const int was = g_swp_lean_epi ? 1 : 0;
  if (on >= 0) g_swp_lean_epi = on != 0;
  return was;
}

int swconv_swp_launch(const ConvArgs& a, int stride, int wm, int wn, int mt,
                      int ksplit, bool dry, hipStream_t stream) {
  // uniform 32-channel K walk, one tap per K-step; the fused LayerNorm needs
  // whole rows in the workgroup
  if (a.CK != 32 || a.taps % stride) return CG_EINVAL;
  if (a.Lx >= 0x3fff) return CG_EINVAL;  // window rows travel as 14-bit fields
  // split-K (blockIdx.y walks its share of the channel chunks, f32 partial sums
  // into the caller's workspace): whole chunks only
  if (ksplit > 1 && (a.narrow || a.epilogue == CG_EPI_LN_LRELU)) return CG_EINVAL;
  const bool ln = a.epilogue == CG_EPI_LN_LRELU;
  if (ln && (wn != 2 || stride != 1 || a.N > 128)) return CG_EINVAL;
  if (stride == 2 && !a.pmajor) return CG_EINVAL;
  const int nw = wm * wn, tn = wn * 64;
  SwpArgs pa;
  pa.c = a;
  pa.tpp = a.taps / stride;
  // whole 64-deep stages per pass, and stages 0 .. nst - 3 to issue the next
  // window in
  if (pa.tpp != kSwpTapsPerPass) return CG_EINVAL;  // the pass loop is unrolled
  pa.nst = pa.tpp / 2;
  pa.WRs = a.S + pa.tpp - 1;
  pa.wrows = a.nseg * pa.WRs;
  pa.npa = (pa.wrows * 64 + 1023) / 1024;
  pa.abytes = pa.npa * 1024;
  // a wave's piece slots k = 0 .. ceil(npa / nw) - 1 are spread over the nst - 2
  // issuing stages, apw per stage (the counted waits cover 0..2 pieces)
  const int kpw = (pa.npa + nw - 1) / nw;
  pa.apw = (kpw + (pa.nst - 2) - 1) / (pa.nst - 2);
  if (pa.apw > 2 || a.nseg > 8 || a.S < 16 * mt) return CG_EINVAL;
  const int narrow = a.narrow ? 1 : 0;
  if (narrow && (stride != 2 || a.nchunks < 2)) return CG_EINVAL;
  pa.npass = (a.nchunks - narrow) * stride + narrow;
  pa.total_stages = (a.nchunks - narrow) * stride * pa.nst + narrow * 4;
  pa.npad_rows = (a.N + 127) / 128 * 128;
  pa.inv_WRs = 1.0f / (float)pa.WRs;
  {
    static const bool off = [] {  // CALCIUMGAN_SWP_CHUNK_INNER=0: the old order (A/B)
      const char* e = getenv("CALCIUMGAN_SWP_CHUNK_INNER");
      return e && e[0] == '0';
    }();
    pa.chunk_inner = (stride == 2 && !off) ? 1 : 0;
  }
  // (window buffers at the tile's compile-time stride: KPW pieces per wave)
  const int tm = wm * mt * 16;
  const int npa_max = tm / 16 + 6;
  if (pa.npa > npa_max) return CG_EINVAL;
  // windows + weight ring + [phase shifts | rowsumsq slots]; the fused
  // LayerNorm adds its statistics table (2 x nw x 16 x 2 floats) and gamma / beta
  size_t lds = (size_t)2 * npa_max * 1024 + (size_t)kSwpRing * tn * 64 * 2 + 64;
  if (ln) lds += (size_t)2 * nw * 32 * 4 + 256 * 4;
  // (row words of the window pieces of the stride-2 8-wave 256 x 64 tile)
  if (stride == 2 && nw == 8 && wn == 1 && mt == 2)
    lds += (size_t)((tm / 16 + 6 + nw - 1) / nw) * nw * 64 * 4;
  pa.bias_off = (int)lds;
  lds += (size_t)a.gn * tn * 4;
  if (lds > 160 * 1024) return CG_EINVAL;
  pa.ntl = ((a.gm + 7) / 8) * 8 * a.gn * a.gp;
  const unsigned grid = (unsigned)(ksplit > 1 ? ksplit : 1);
  if (ln) {
    if (wm == 4 && mt == 4) return launch_swp<1, 4, 2, 4, true>(pa, grid, lds, dry, stream);
    if (wm == 2 && mt == 4) return launch_swp<1, 2, 2, 4, true>(pa, grid, lds, dry, stream);
    if (wm == 4 && mt == 2) return launch_swp<1, 4, 2, 2, true>(pa, grid, lds, dry, stream);
    return CG_EINVAL;
  }
  // lean epilogue forms: the 32-row wave tiles (what the tuner picks for nearly
  // every cfg2 geometry), bf16 rows, no split-K / f32 output / penalty norm
  int epi = kEpiGeneric;
  if (mt == 2 && !a.out_f32 && ksplit <= 1 && g_swp_lean_epi) {
    if ((a.epilogue == CG_EPI_NONE || a.epilogue == CG_EPI_LRELU) &&
        !a.out_shifts && !a.row_scale) {
      epi = a.rowsumsq ? kEpiLreluSsq : kEpiLrelu;
      if (epi == kEpiLreluSsq && stride != 1) epi = kEpiGeneric;
      if (a.epilogue == CG_EPI_NONE) pa.c.alpha = 1.f;  // max(v, v)
    } else if (a.rowsumsq) {
      epi = kEpiGeneric;
    } else if (a.epilogue == CG_EPI_MASK && !a.out_shifts) {
      epi = kEpiMask;
    } else if (a.epilogue == CG_EPI_MASK && !a.row_scale && stride == 1) {
      epi = kEpiMaskShift;
    }
    // (stride 1 carries the two forms its launches use, stride 2 likewise)
    if (stride == 1 && epi == kEpiMask) epi = kEpiGeneric;
  }
#define CG_SWP_E(RR, WM, WN, MM, NN, EE)                                     \
  if (stride == RR && wm == WM && wn == WN && mt == MM && narrow == NN &&    \
      epi == EE)                                                             \
    return launch_swp<RR, WM, WN, MM, false, NN != 0, EE>(pa, grid, lds, dry, \
                                                           stream);
#define CG_SWP(RR, WM, WN, MM, NN) CG_SWP_E(RR, WM, WN, MM, NN, kEpiGeneric)
#define CG_SWP_R(WM, WN, MM) \
  CG_SWP(1, WM, WN, MM, 0) CG_SWP(2, WM, WN, MM, 0) CG_SWP(2, WM, WN, MM, 1)
#define CG_SWP_LEAN(WM, WN, MM)                                              \
  CG_SWP_E(1, WM, WN, MM, 0, kEpiLrelu) CG_SWP_E(1, WM, WN, MM, 0, kEpiMaskShift) \
  CG_SWP_E(1, WM, WN, MM, 0, kEpiLreluSsq)                                        \
  CG_SWP_E(2, WM, WN, MM, 0, kEpiLrelu) CG_SWP_E(2, WM, WN, MM, 0, kEpiMask)  \
  CG_SWP_E(2, WM, WN, MM, 1, kEpiLrelu) CG_SWP_E(2, WM, WN, MM, 1, kEpiMask)
  CG_SWP_LEAN(4, 1, 2) CG_SWP_LEAN(8, 1, 2) CG_SWP_LEAN(4, 2, 2)
  CG_SWP_R(8, 1, 4)   // 512 x 64, 8 waves
  CG_SWP_R(4, 1, 4)   // 256 x 64, 4 waves
  CG_SWP_R(4, 2, 4)   // 256 x 128, 8 waves
  CG_SWP_R(2, 2, 4)   // 128 x 128, 4 waves
  CG_SWP_R(4, 1, 2)   // 128 x 64, 4 waves
  CG_SWP_R(8, 1, 2)   // 256 x 64, 8 waves
  CG_SWP_R(4, 2, 2)   // 128 x 128, 8 waves
#undef CG_SWP_LEAN
#undef CG_SWP_R
#undef CG_SWP
#undef CG_SWP_E
  return CG_EINVAL;
}
