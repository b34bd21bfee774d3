/* calciumgan_hip.h -- C ABI of the MI355X (gfx950) CalciumGAN hot-path kernels.
 *
 * Drop-in boundary: the reference has no FFI; the device ops
 * it executes are whatever TensorFlow dispatches for the Keras layers named
 * below.  Each entry point here replaces the device work of one reference
 * call site (file:line relative to the reference repo) and is what a
 * maintainer would bind (ctypes stub in INTEGRATION.md).
 *
 * Conventions for every function:
 *   - plain pointers and sizes only; all pointers are DEVICE pointers unless
 *     the parameter is a `const cg_*_desc*` (host struct, read at call time);
 *   - activations are channels-last bf16 [nB][L][Cp], Cp = channel pitch, a
 *     multiple of 8; channels [C, Cp) are kept zero by every kernel;
 *   - `stream` is a hipStream_t (NULL = default stream); calls are async,
 *     re-entrant per stream, allocate nothing and keep no global state;
 *   - the return value is a hipError_t (0 = success) or CG_EINVAL for a
 *     shape the kernels do not support (nothing is launched in that case).
 */
#ifndef CALCIUMGAN_HIP_H_
#define CALCIUMGAN_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CG_EINVAL 100001
#define CG_ABI_VERSION 19

/* epilogue selectors of cg_swconv */
#define CG_EPI_NONE 0     /* y = acc (+bias) */
#define CG_EPI_LRELU 1    /* y = leaky_relu(acc + bias, alpha) */
#define CG_EPI_MASK 2     /* y = acc * (mask_src > 0 ? 1 : alpha) */
#define CG_EPI_SIGMOID 3  /* y = sigmoid(acc + bias) */
#define CG_EPI_LN_LRELU 4 /* y = acc + bias; ln_h = leaky_relu(layernorm(y)) */

/* workgroup tile of cg_swconv (rows x columns of y per workgroup); the _M32
 * tiles run v_mfma_f32_32x32x16_bf16 and need CK % 32 == 0, the others
 * v_mfma_f32_16x16x32_bf16.  Every tile gives the same result up to the f32
 * summation order inside one 32-deep K-step. */
#define CG_TILE_256x64 0
#define CG_TILE_64x64 1
#define CG_TILE_128x64 2
#define CG_TILE_256x64_M32 3
#define CG_TILE_128x64_M32 4
#define CG_TILE_256x128_M32 5
#define CG_TILE_128x128_M32 6
#define CG_TILE_256x128 7
#define CG_TILE_128x128 8
/* Software-pipelined tiles (swconv_swp.hip): two waves per SIMD (one 8-wave
 * workgroup or two 4-wave workgroups per CU), fragments double-buffered in
 * registers at K-step granularity, the source window and the weight ring both
 * filled by LDS-DMA.  v_mfma_f32_16x16x32, CK == 32, at least six taps per
 * source-row parity; split-K over whole channel chunks; the LayerNorm epilogue
 * on the 128-column tiles.
 * Same results as the tiles above. */
#define CG_TILE_SWP_512x64 9
#define CG_TILE_SWP_256x64 10
#define CG_TILE_SWP_256x128 11
#define CG_TILE_SWP_128x128 12
#define CG_TILE_SWP_128x64 13  /* 4 waves x (32 x 64): short launches (more workgroups) */
#define CG_TILE_SWP_256x64_W8 14  /* 8 waves x (32 x 64): two workgroups = 4 waves per SIMD */
#define CG_TILE_SWP_128x128_W8 15 /* 8 waves x (32 x 64), 4 x 2: the same for 128 columns */
#define CG_NUM_TILES 16

int cg_abi_version(void);
/* Storage type of activations / activation gradients / packed operands this
 * build of the library computes with (everything the comments below call
 * "bf16"): libcalciumgan_hip.so is the bf16 build, libcalciumgan_hip_f16.so the
 * IEEE fp16 build (-DCG_ACT_F16=1; the reference's mixed_float16 policy,
 * main.py:22-30, with the loss scaling entry points at the end of this file).
 * Accumulators, statistics, losses, master weights and Adam state are f32 in
 * both. */
#define CG_DTYPE_BF16 0
#define CG_DTYPE_F16 1
int cg_act_dtype(void);
/* sizeof() of the descriptor structs as this library was compiled
 * (which: 0 cg_conv_desc, 1 cg_pack_desc, 2 cg_wgrad_desc; else -1): a binding
 * checks its own struct layout against it before the first launch. */
int cg_struct_size(int which);
/* rows / columns of a CG_TILE_* value (host helper; CG_EINVAL if unknown) */
int cg_tile_shape(int tile, int* rows, int* cols);

/* Measurement hook (the one piece of process-wide state in this library; not
 * for use while a stream is being captured into a hipGraph).  After
 * cg_profile_enable(n) the next n cg_swconv / cg_wgrad launches are issued with
 * a HIP event pair that carries the kernel's own begin / end timestamps;
 * cg_profile_collect waits for them, writes each launch's duration (ms) and
 * family (0 = cg_swconv, 1 = cg_wgrad) in launch order, returns the count
 * (negative hipError_t on failure) and disables the hook.  bench.py's roofline
 * leg is its only user. */
/* Development switch (process-wide, like the hook below): the specialised
 * epilogues of the 32-row software-pipelined tiles (swconv_swp.hip, kEpi*) on
 * (1) / off (0); < 0 only queries.  Returns the previous setting.  Off by
 * default (CALCIUMGAN_SWP_LEAN_EPI=1 turns them on at load): same results bit
 * for bit, no faster. */
int cg_debug_lean_epilogue(int on);
/* Form of cg_wgrad_batched for launches whose layers all take the ring-staged
 * 24-tap kernel with `partials` (round 5, ABI 18): 0 the K'-split forms, 1
 * (default; env CALCIUMGAN_WGRAD_FLEX) the flex form -- layers side by side,
 * equal contiguous K' shares -- when every team's share is worth its set-up, 2
 * the flex form whatever the size (tests).  Returns the previous mode; any
 * other argument only reads it.  Results are identical on exactly representable
 * data; in floating point the forms differ in summation order. */
int cg_debug_wgrad_flex(int mode);

int cg_profile_enable(int max_launches);
int cg_profile_collect(float* ms, int* family, int capacity);

/* ---------------------------------------------------------------------------
 * Sliding-window convolution as an implicit GEMM on MFMA (bf16 in, f32 acc).
 *
 *   y[b, y_stride*u + y_off, n] = epi( bias[n] +
 *        sum_{tap<taps} sum_{c<Cx} xs[b, stride*u + off + tap, c] * Wl[tap][c][n] )
 *
 * for u in [0, Lu), n in [0, N); rows of xs outside [0, Lx) read as zero
 * (TF 'same' padding) and xs = phase_shuffle(x, shifts[b / seg_size]) when
 * `shifts` is non-NULL (reflect gather fused into the LDS staging).
 * With nphase == 2 the launch computes two output phases z = 0,1 using
 * w + z*w_phase_stride, off + z*off_phase_step, y_off + z*yoff_phase_step
 * (the two 12-tap phases of a stride-2 transposed convolution).
 *
 * Replaces: layers.Conv1D fwd (gan/models/calciumgan.py:145-185) [stride 2,
 * taps k]; its input-gradient / Conv1DTranspose fwd (gan/models/utils.py:86-89)
 * [stride 1, taps k/2, two phases]; Conv1DTranspose input-gradient [stride 2];
 * layers.Dense on the last axis (calciumgan.py:32,96) [taps 1].
 *
 * `w` is the packed operand produced by cg_pack_weights for the same
 * (taps, Cx, CK).  Supported: stride in {1,2}; taps even when stride == 2;
 * Cx % CK == 0, CK % 8 == 0, CK >= 32; Lu a power-of-two divisor or any
 * multiple of the row tile (cg_tile_shape).
 * ------------------------------------------------------------------------- */
typedef struct cg_conv_desc {
  const void* x;        /* bf16 [nB][Lx][Cx] */
  const void* w;        /* bf16 packed [ceil(N/128)*128][Kpack] per phase */
  void* y;              /* bf16 or f32 [nB][Ly][Cy] */
  const float* bias;    /* f32 [N] or NULL */
  const void* mask_src; /* bf16, geometry of y; CG_EPI_MASK only */
  const int* shifts;    /* int32 [ceil(nB/seg_size)] or NULL */
  int nB, Lx, Cx, seg_size;
  int taps, stride, off, Lu;
  int N, Ly, Cy, y_stride, y_off;
  int CK;
  int epilogue, out_f32;
  float alpha;          /* leaky slope */
  int nphase;
  long long w_phase_stride; /* elements */
  int off_phase_step, yoff_phase_step;
  int tile;             /* CG_TILE_* */
  int stage_ksteps;     /* 0: choose; 2 or 4: MFMA K-steps per weight stage */
  float* rowsumsq;      /* optional f32 [nB]: += sum over (row, n) of y^2 per
                           sample (penalty norm, wgan_gp.py:49); needs Lu >= tile */
  int w_parity_major;   /* stride 2: `w` was packed with parity_major = 1 */
  int split_parity;     /* stride 2 + w_parity_major: stage one source-row
                           parity at a time (half the LDS window, twice the
                           staging phases; results identical) */
  /* CG_EPI_LN_LRELU only (layers.LayerNormalization + LeakyReLU after
   * Conv1DTranspose, calciumgan.py:68-70 fused into the producing launch):
   * y receives the bf16 pre-activation, ln_h = lrelu(LN(y)), ln_mean / ln_rstd
   * the per-row statistics (indexed like rows of y) cg_ln_lrelu_bwd consumes.
   * Needs N <= 128, bf16 output and a 128-column tile (cg_tile_shape).
   * ln_mean = ln_rstd = NULL: forward only -- neither the statistics nor the
   * pre-activation y are stored (y may then be any non-NULL pointer). */
  const float* ln_gamma; /* f32 [N] */
  const float* ln_beta;  /* f32 [N] */
  void* ln_h;            /* bf16, geometry of y */
  float* ln_mean;        /* f32 [nB*Ly] */
  float* ln_rstd;        /* f32 [nB*Ly] */
  float ln_eps;
  /* Output-side PhaseShuffle adjoint (input gradient of a layer whose input
   * was shuffled, calciumgan.py:117-138): output row t of sample b belongs to
   * source row r = shuffle_src(t, out_shifts[b / out_seg_size], Ly).  Rows on
   * the direct branch are stored at r (bias / mask epilogues read row r); rows
   * on the reflected branch (at most |shift| per sample) go unmasked to
   * side[b][j] and are folded in by cg_unshuffle_fixup, which also zeroes the
   * |shift| rows nothing maps to.  NULL: rows are stored where they are
   * computed.  bf16 output only. */
  int w_narrow_last;    /* `w` was packed with narrow_last = 1 (stride 2,
                           w_parity_major, CK == 32) */
  const int* out_shifts;
  int out_seg_size;
  void* side;           /* bf16 [nB][side_rows][Cy] */
  int side_rows;        /* >= max |shift| */
  /* Split-K for launches with few output tiles (the penalty's tangent chain:
   * one 128-sample segment): ksplit > 1 workgroups share an output tile, each
   * walking Cx/CK/ksplit channel chunks and storing f32 partial sums to
   * split_ws[z]; a finishing launch adds them, applies bias / LeakyReLU / mask
   * and writes the bf16 y.  Needs (Cx/CK) % ksplit == 0, bf16 output,
   * epilogue NONE / LRELU / MASK, no rowsumsq / out_shifts.  0 or 1: off. */
  int ksplit;
  float* split_ws;      /* f32 [ksplit][nB][Ly][Cy] */
  long long split_ws_elems;
  /* Per-sample scale applied before the epilogue: y = epi((acc + bias) *
   * row_scale[sample]).  The penalty's tangent chain starts from v = coef_b * g:
   * with this, its first launch reads g itself and the 134 MB pass that scaled
   * it is gone (wgan_gp.py, _critic_compute).  Software-pipelined tiles only,
   * no split-K, not with the fused LayerNorm; null: off. */
  const float* row_scale;
  /* Ordered penalty norm: with a workspace of >= cg_rowsumsq_ws_elems(d) floats
   * every workgroup STORES its share of a sample's sum of squares in its own
   * slot and a finishing launch adds a sample's slots in slot order and stores
   * rowsumsq[b] -- the same bits every run, no zeroed buffer -- instead of one
   * f32 atomic per workgroup into rowsumsq[b].  NULL: atomics (+=). */
  float* rowsumsq_ws;
  long long rowsumsq_ws_elems;
  /* 1: leave the slots in rowsumsq_ws and do NOT launch the finishing pass --
   * the caller's next launch adds them (cg_gp_loss_scale reads the slots
   * itself: one launch fewer per critic update).  Needs rowsumsq_ws. */
  int rowsumsq_defer;
} cg_conv_desc;

int cg_swconv(const cg_conv_desc* d, void* stream);
/* floats of cg_conv_desc.rowsumsq_ws for this descriptor's tile (0: no rowsumsq;
 * < 0: invalid descriptor) */
long long cg_rowsumsq_ws_elems(const cg_conv_desc* d);
/* Validate a descriptor exactly as cg_swconv would (geometry, tile, LDS budget,
 * epilogue / split-K constraints) without launching anything: 0 or CG_EINVAL.
 * The host uses it before adopting a tile choice from a saved table. */
int cg_swconv_check(const cg_conv_desc* d);

/* Streaming form of the 1-tap case with f32 output (the generator's last
 * layers.Dense + sigmoid, calciumgan.py:96-101; HBM-bound):
 *   y[r, n] = epi(bias[n] + sum_c x[r, c] * W[c][n]),  y[r, n >= N] = 0
 * x bf16 [rows][Cx], y f32 [rows][Cy]; `w` is the cg_pack_weights operand for
 * (taps 1, Cx, CK 32).  Epilogue CG_EPI_NONE or CG_EPI_SIGMOID.  Two forms:
 * Cx in {32, 64, 96, 128} and N <= Cy <= 128 (Cy % 4 == 0): every wave holds W
 * in registers; otherwise Cx in {128, 256, 384, 512}, any N <= Cy (Cy % 8 == 0):
 * a workgroup holds one 128-column panel of W in LDS (BASELINE configs[4]:
 * 512 -> 512). */
int cg_dense_rows(const void* x, const void* w, const float* bias, float* y,
                  long long rows, int Cx, int N, int Cy, int epilogue,
                  void* stream);
/* cg_dense_rows for the fake batches of ALL n critic updates of one train() -- x
 * bf16 [n * B * L][Cx], the generator's last hidden layer over n * B samples --
 * fused with cg_interp_pack.  Register form: Cx in {32, 64, 96, 128}, N <= 128, Cp =
 * 128.  LDS-panel form (ABI 19; BASELINE configs[4]): Cx in {128, 256, 384, 512}
 * with Cx > 128 or N > 128, Cp >= N a multiple of 8, B * L % 32 == 0; without x^
 * (alpha == NULL) the Dense only stores its fake segments and a second launch
 * reads `real` once for the n real segments.
 * x0[k] (bf16 [3 B][L][Cp]) receives [real | fake_k | x^_k], x^ =
 * alpha[k * B + b] * real + (1 - alpha) * fake (wgan_gp.py:38-41, interpolation
 * in f32 on the f32 Dense output as the reference does).  The f32 fake batch
 * never reaches HBM and `real` (f32 [B][L][Cr]) is read once.  L % 16 == 0,
 * n <= 8.  The bytes of x0[k] equal those of cg_dense_rows + cg_interp_pack.
 * alpha == NULL (ABI 19): the x^ segment is left untouched -- the caller forms the
 * critic's first layer on x^ with cg_lrelu_mix and never reads x^ itself. */
int cg_dense_rows_interp(const void* x, const void* w, const float* bias,
                         const float* real, const float* alpha, void* const* x0,
                         int n, int B, int L, int Cx, int N, int Cr, int Cp,
                         int epilogue, void* stream);
/* The same contraction with an activation-typed (bf16 / fp16) output, no bias,
 * no epilogue -- the input gradient of that Dense: dh[r, c] = sum_n dz[r, n] *
 * W[c][n] with `w` the operand packed from W transposed.  Cx in {128, 256, 384,
 * 512}, N <= Cy, Cy % 8 == 0; columns [N, Cy) are written as zeros. */
int cg_dense_rows_act(const void* x, const void* w, void* y, long long rows,
                      int Cx, int N, int Cy, void* stream);
/* Weight gradient of that Dense, dW f32 [Cx_real][Cg_real] row-major (the Keras
 * kernel layout), x [rows][Cx], g [rows][Cg] bf16 / fp16, pitches multiples of
 * 8.  Streaming form: 128 x 128 tiles of dW x row ranges, K'-major operands
 * through LDS transpose reads.  ws == NULL: the row ranges' partial tiles meet
 * through f32 atomics, dW[cx][cg] += sum_r x[r][cx] * g[r][cg] (the caller
 * zeroes dW).  With a workspace of cg_dense_wgrad_ws_elems(...) floats: plain
 * stores and a reducing launch that adds the ranges in a fixed order and STORES
 * dW -- the only sensible form when dW has at most four tiles (a thousand
 * workgroups' atomics on the same addresses serialise), and the deterministic
 * one for any size. */
long long cg_dense_wgrad_ws_elems(long long rows, int Cx_real, int Cg_real);
int cg_dense_wgrad(const void* x, const void* g, float* dw, long long rows,
                   int Cx, int Cg, int Cx_real, int Cg_real, float* ws,
                   long long ws_elems, void* stream);
/* elements (bf16) of one packed phase operand for (N, taps, Cx, CK) */
long long cg_packed_elems(int N, int taps, int Cx, int CK);

/* ---------------------------------------------------------------------------
 * Pack an f32 master weight (TensorFlow layout, arbitrary strides) into the
 * bf16 operand cg_swconv consumes:
 *   Wl[tap][c][n] = src[(tap0 + tap*tap_step)*s_tap + c*s_c + n*s_n]
 * for tap<taps, c<C_real, n<N_real; zero elsewhere (channel / row padding).
 * Replaces nothing in the reference (TF keeps one layout); it is the
 * MI355X-side operand preparation run after each optimizer update.
 * ------------------------------------------------------------------------- */
typedef struct cg_pack_desc {
  const float* src;
  void* dst; /* bf16, cg_packed_elems(N_real, taps, Cx, CK) elements */
  int taps, tap0, tap_step;
  long long s_tap, s_c, s_n;
  int C_real, N_real, Cx, CK;
  int parity_major; /* 0: taps in order; 1 (even taps only): the even taps first,
                       then the odd ones -- the order in which a stride-2 launch
                       with w_parity_major walks them (one source-row parity
                       at a time) */
  int narrow_last;  /* 1 (needs parity_major, CK == 32, taps <= 32, and
                       Cx - 32 < C_real <= Cx - 24: the last channel chunk
                       holds at most 8 real channels): that chunk is packed as
                       one 8-channel group per tap -- 16 slots per tap parity,
                       K = 32 groups instead of taps*4 -- for launches with
                       cg_conv_desc.w_narrow_last, which then skip the chunk's
                       all-zero channel groups */
} cg_pack_desc;
int cg_pack_weights(const cg_pack_desc* d, void* stream);
/* Batched form (one launch for all operands of a model).  Host side:
 *   blocks = cg_pack_plan_build(descs, n, NULL, 0);            (sizing pass)
 *   bytes  = cg_pack_plan_bytes(n, blocks);
 *   cg_pack_plan_build(descs, n, host_buf, bytes);             (fills host_buf)
 * copy host_buf to device memory once, then after every optimizer update
 *   cg_pack_batched(dev_plan, n, blocks, stream). */
long long cg_pack_plan_bytes(int n, long long total_blocks);
long long cg_pack_plan_build(const cg_pack_desc* descs, int n, void* host_buf,
                             long long host_bytes);
int cg_pack_batched(const void* dev_plan, int n, long long blocks, void* stream);

/* ---------------------------------------------------------------------------
 * Weight gradient of the sliding-window convolution (f32 accumulate, f32
 * atomics into dw, which the caller zeroes):
 *   dw[tap][cx][cg] += sum_{b,u} xs[b, stride*u + off + tap, cx] * g[b, u, cg]
 * dw is dense f32 [taps][Cx_real][Cg_real].  Replaces the autodiff
 * weight-gradient of Conv1D / Conv2DTranspose / Dense taken by
 * tape.gradient in gan/algorithms/optimizer.py:31-34.
 * Supported: (stride 2, even taps <= 24) or (stride 1, taps 1).
 * ------------------------------------------------------------------------- */
typedef struct cg_wgrad_desc {
  const void* x;     /* bf16 [nB][Lx][Cx] */
  const void* g;     /* bf16 [nB][Lu][Cg] */
  float* dw;         /* f32 [taps][Cx_real][Cg_real] */
  const int* shifts; /* phase shuffle applied to x rows, or NULL */
  int nB, Lx, Cx, seg_size;
  int Lu, Cg;
  int taps, stride, off;
  int Cx_real, Cg_real;
  int nsplit;        /* 0 = choose */
  int tile_rows;     /* 0 = choose; 64 or 128 rows of (b,u) per staged tile */
  int no_xcd_group;  /* 1: plain block order instead of the XCD-grouped one (A/B runs) */
  int classic_staging; /* 1: register-staged tiles instead of the LDS-DMA ring (A/B runs) */
  float* dbias;      /* optional f32 [Cg_real]: += sum of g over its first
                        bias_rows (b,u) rows -- the conv bias gradient, taken
                        from the g tiles already staged in LDS */
  long long bias_rows;
  float* partials;   /* optional workspace of >= cg_wgrad_partials_elems(d) f32:
                        the K' splits then leave their partial sums (dw tiles
                        and, with dbias, their bias column sums) with plain
                        stores and a second kernel adds them in split order (one
                        owner per element: deterministic), instead of every
                        split adding into dw / dbias with f32 atomics.  NULL:
                        atomics. */
  long long partials_elems;
  int store;         /* with `partials`: dw (and dbias) are STORED, not added to
                        (no zeroed buffer needed).  0: += as before. */
} cg_wgrad_desc;
int cg_wgrad(const cg_wgrad_desc* d, void* stream);
/* f32 elements of `partials` this descriptor needs (0: the taps == 1 form, which
 * adds each dw element once per K' split with atomics -- one split, nsplit = 1,
 * is deterministic); < 0: error */
long long cg_wgrad_partials_elems(const cg_wgrad_desc* d);
/* n weight gradients (independent layers of one backward pass) as ONE launch
 * when they are all the pipelined stride-2 form with the same tap count (the
 * accumulator flush of one layer then overlaps the next layer's main loop);
 * otherwise the same as n cg_wgrad calls.  Results as cg_wgrad (f32 atomics:
 * the summation order differs); descriptors with `partials` are reduced by one
 * extra launch for all layers. */
int cg_wgrad_batched(const cg_wgrad_desc* descs, int n, void* stream);
/* The flex form's share plan for inspection (host only, nothing is launched):
 * returns the number of ints of the table -- items [workgroups][8][6] = (layer
 * or -1, bx, by, first K' tile, tiles, slot), then per layer [gx * gy][2] =
 * (first slot, slots) -- and copies it to `out` when out_ints suffices; info
 * (>= 16 ints): team size, teams, workgroups, items, slots of layer 0..5, int
 * offsets of the layers' tile tables.  < 0: not the flex form under `mode`. */
long long cg_wgrad_flex_plan(const cg_wgrad_desc* descs, int n, int mode, int* out,
                             long long out_ints, int* info);

/* Second half of cg_conv_desc.out_shifts: per sample with shift s,
 *   delta[b, r, :] = 0 for the |s| rows r no output row maps to, and
 *   delta[b, r, :] += side[b, j, :] * (h[b, r, :] > 0 ? 1 : alpha) for the |s|
 * reflected rows.  delta / h bf16 [nB][w][Cp], side bf16 [nB][side_rows][Cp]. */
int cg_unshuffle_fixup(const void* side, const void* h, void* delta,
                       const int* shifts, int nB, int w, int Cp, int seg_size,
                       int side_rows, float alpha, void* stream);

/* ---------------------------------------------------------------------------
 * LayerNormalization(axis=-1, eps) + LeakyReLU, one wavefront per row.
 * Replaces layers.LayerNormalization + activation_fn
 * (calciumgan.py:44-46 etc.) and their autodiff backward.
 * ------------------------------------------------------------------------- */
int cg_ln_lrelu_fwd(const void* y_pre /*bf16 [rows][Cp]*/, const float* gamma,
                    const float* beta, void* h /*bf16 [rows][Cp]*/,
                    float* mean /*[rows] or NULL*/, float* rstd /*[rows] or NULL*/,
                    long long rows, int C, int Cp, float eps, float alpha,
                    void* stream);
/* Ordered reductions.  The functions below that reduce over rows take a
 * workspace `ws` of cg_reduce_ws_elems() floats (one size for every function and
 * shape; launches are stream-ordered, so one buffer serves them all).  With it,
 * every block STORES a partial row and a finishing launch adds the rows in a
 * fixed order and STORES the result: the same bits every run and no zeroed
 * output.  ws == NULL: blocks meet through f32 atomics (+= into outputs the
 * caller zeroed; the order, and so the last bits, vary from run to run). */
long long cg_reduce_ws_elems(void);
/* Deferred finishing launches (round 5): between cg_finish_defer(1) and
 * cg_finish_flush(stream) on the calling thread, the finishing launch of every
 * ordered reduction above is queued instead of issued, and the flush adds all
 * queued reductions in ONE launch (up to 12 per launch; the same sums in the same
 * order).  Every queued call must have been given its own workspace region -- its
 * partial rows are read at the flush -- and nothing between may read the outputs
 * (cg_bn_bwd does: not deferrable).  cg_finish_defer returns the previous mode. */
int cg_finish_defer(int on);
int cg_finish_flush(void* stream);
/* dy = d(loss)/d(y_pre); dgamma/dbeta (and, when dbias != NULL, the bias
 * gradient of the producing conv = column sums of dy): stored (ws) or
 * accumulated with f32 atomics (ws == NULL). */
int cg_ln_lrelu_bwd(const void* dh /*bf16*/, const void* h /*bf16*/,
                    const void* y_pre /*bf16*/, const float* mean,
                    const float* rstd, const float* gamma, void* dy /*bf16*/,
                    float* dgamma, float* dbeta, float* dbias, long long rows,
                    int C, int Cp, float alpha, float* ws, void* stream);

/* ---------------------------------------------------------------------------
 * BatchNormalization(axis=-1) of a generator block (calciumgan.py:42-43 etc.;
 * Keras defaults: momentum 0.99, epsilon 1e-3, biased batch variance) -- single
 * rank only (batch statistics would need a cross-rank reduction).  All three
 * need the workspace of cg_reduce_ws_elems() floats (ordered column sums).
 *   cg_bn_stats: mean[c] / var[c] over the rows of y; with moving_mean /
 *     moving_var (training): moving = moving * momentum + batch * (1 - momentum).
 *   cg_bn_apply: out = f((y - mean) * rsqrt(var + eps) * gamma + beta), f(t) =
 *     max(t, alpha t) (alpha = 1: no activation); inference passes the moving
 *     statistics.
 *   cg_bn_bwd: do = dout * (act ? lrelu'(h) : 1); dbeta = sum do, dgamma = sum
 *     do * xhat (both STORED); dy = gamma rstd (do - dbeta / R - xhat dgamma / R).
 * y / out / dout / h / dy bf16 [rows][Cp], Cp <= 2048.
 * ------------------------------------------------------------------------- */
int cg_bn_stats(const void* y, long long rows, int C, int Cp, float* mean,
                float* var, float* moving_mean /* or NULL */,
                float* moving_var /* or NULL */, float momentum, float* ws,
                void* stream);
int cg_bn_apply(const void* y, const float* mean, const float* var,
                const float* gamma, const float* beta, void* out, long long rows,
                int C, int Cp, float eps, float alpha, void* stream);
int cg_bn_bwd(const void* dout, const void* h /* act != 0 */, const void* y,
              const float* mean, const float* var, const float* gamma, void* dy,
              float* dgamma, float* dbeta, long long rows, int C, int Cp,
              float eps, float alpha, int act, float* ws, void* stream);

/* ---------------------------------------------------------------------------
 * Discriminator head: Flatten + Dense(1) (calciumgan.py:188-190).
 * ------------------------------------------------------------------------- */
/* h is bf16 [nB][Lt][Cp]; w is the f32 Keras kernel [Lt*C] (flatten index
 * t*C + c, calciumgan.py:188), used rounded to bf16.
 * out[b] = bias[0] + sum_{t,c<C} h[b][t][c] * bf16(w[t*C+c]) */
int cg_dense1_fwd(const void* h, const float* w, const float* bias,
                  float* out /*[nB]*/, int nB, int Lt, int C, int Cp,
                  void* stream);
/* cg_dense1_fwd and cg_dense1_bwd in one pass over h (the seed of the backward
 * chain does not depend on the head's output) */
int cg_dense1_fwd_bwd(const void* h, const float* w, const float* bias,
                      float* out /*[nB]*/, const float* coef /*device [nseg]*/,
                      void* delta /*bf16*/, int nB, int Lt, int C, int Cp,
                      int seg_size, float alpha, void* stream);
/* delta[b][t][c] = coef[b / seg_size] * bf16(w[t*C+c]) * lrelu'(h[b][t][c]) */
int cg_dense1_bwd(const float* w, const float* coef /*device [nseg]*/,
                  const void* h, void* delta /*bf16*/, int nB, int Lt, int C,
                  int Cp, int seg_size, float alpha, void* stream);
/* dw[t*C+c] += sum_b coef[b / seg_size] * x[b][t][c];
 * db[0] += sum_b bias_coef[b / seg_size]  (bias_coef may be NULL);
 * with ws: = instead of += (ordered reduction, see cg_reduce_ws_elems) */
int cg_dense1_wgrad(const void* x /*bf16 [nB][Lt][Cp]*/, const float* coef,
                    const float* bias_coef /*device [nseg]*/, float* dw,
                    float* db, int nB, int Lt, int C, int Cp, int seg_size,
                    float* ws, void* stream);

/* ---------------------------------------------------------------------------
 * Backward of LeakyReLU + PhaseShuffle between discriminator layers:
 *   delta[b][r][c] = lrelu'(h[b][r][c]) * sum_{t: src(t)=r} e[b][t][c]
 * where src is the reflect gather of PhaseShuffle (calciumgan.py:117-138).
 * ------------------------------------------------------------------------- */
int cg_unshuffle_mask(const void* e /*bf16 [nB][w][Cp]*/, const void* h,
                      void* delta, const int* shifts, int nB, int w, int Cp,
                      int seg_size, float alpha, void* stream);

/* ---------------------------------------------------------------------------
 * WGAN-GP elementwise / reduction pieces (gan/algorithms/wgan_gp.py).
 * ------------------------------------------------------------------------- */
/* x0[0:B]=bf16(real), x0[B:2B]=bf16(fake), x0[2B:3B]=bf16(alpha*real+(1-alpha)*fake)
 * (wgan_gp.py:38-41); real f32 [B][L][Cr], fake f32 [B][L][Cf] (row pitches,
 * C valid channels), x0 bf16 [3B][L][Cp].  alpha == NULL (ABI 19): the third
 * segment is left untouched (see cg_lrelu_mix). */
int cg_interp_pack(const float* real, const float* fake, const float* alpha,
                   void* x0, int B, int L, int C, int Cr, int Cf, int Cp,
                   int write_real /* 0: x0[0:B] already holds bf16(real) */,
                   void* stream);
/* dst bf16 [rows][Cp] = src f32 [rows][Cs] (C valid channels), pad zero */
int cg_cast_pad(const float* src, void* dst, long long rows, int C, int Cs,
                int Cp, void* stream);
/* norm[b] = ||g[b]||_2 over n elements (wgan_gp.py:49), g bf16 [B][n] (the
 * layer-1 input gradient, stored like every other activation gradient) */
int cg_rownorm(const void* g, float* norm, int B, long long n,
               float* ws /* ordered reduction, or NULL */, void* stream);
/* gp = mean((norm-1)^2) (wgan_gp.py:50); coef[b] = scale*2*(norm-1)/(B*norm).
 * squared != 0: `norm` holds sums of squares on entry and is replaced by their
 * square roots. */
int cg_gp_finalize(float* norm, float* gp, float* coef, int B, float scale,
                   int squared, void* stream);
/* a0 bf16 [B][n] = coef[b] * g[b], g bf16 [B][n] */
int cg_scale_rows(const void* g, const float* coef, void* a0, int B,
                  long long n, void* stream);
/* out[0] = -mean(d_out[0:B]) + mean(d_out[B:2B]) + penalty*gp[0]  (wgan_gp.py:58-61)
 * out[1] = -mean(d_out[B:2B]) (generator_loss of the same fake batch) */
int cg_critic_loss(const float* d_out, const float* gp, float penalty,
                   float* out, int B, void* stream);
/* cg_gp_finalize followed by cg_critic_loss as ONE launch: coef additionally
 * multiplied by coef_mul; loss[0] / loss[1] as cg_critic_loss's out. */
int cg_gp_critic_loss(float* norm, float* gp, float* coef, const float* d_out,
                      float* loss /*[2]*/, int B, float penalty, int squared,
                      float coef_mul, void* stream);
/* The penalty's finalisation as ONE launch (wgan_gp.py:43-62): adds the slots a
 * cg_swconv launch with rowsumsq_defer left in ssq_ws ([B][P] floats, P =
 * cg_rowsumsq_ws_elems / nB) in slot order, norm[b] = sqrt of that, gp / coef /
 * loss exactly as cg_gp_critic_loss(squared = 1), and -- g non-null -- dst[b] =
 * coef[b] * g[b] for bf16 [B][n] rows (cg_scale_rows: every workgroup forms its
 * sample's coefficient from the same slots, bit for bit the value coef[] gets). */
int cg_gp_loss_scale(const float* ssq_ws, int P, float* norm, float* gp,
                     float* coef, const float* d_out, float* loss /*[2]*/, int B,
                     float penalty, float coef_mul, const void* g, void* dst,
                     long long n, void* stream);
/* out[0] = -mean(d_out[0:B])  (wgan_gp.py:19-20) */
int cg_neg_mean(const float* d_out, float* out, int B, void* stream);

/* out f32[C] += column sums of x bf16 [rows][Cp] (bias gradients); with ws:
 * = instead of += (ordered reduction) */
int cg_colsum(const void* x, float* out, long long rows, int C, int Cp,
              float* ws, void* stream);
/* dz = dfake * s * (1 - s)  (sigmoid backward; calciumgan.py:98-99) */
int cg_sigmoid_bwd(const void* dfake /*bf16 [rows][Cp]*/,
                   const float* fake /*f32 [rows][Cf]*/,
                   void* dz /*bf16 [rows][Cp]*/, long long rows, int C, int Cf,
                   int Cp, void* stream);
/* dpre = dh * lrelu'(h), all bf16 [n] */
int cg_lrelu_bwd(const void* dh, const void* h, void* dpre, long long n,
                 float alpha, void* stream);
/* out[b] = act(mix[b] * act^-1(h_a[b]) + (1 - mix[b]) * act^-1(h_b[b])), all bf16
 * [B][n_per_sample], act(y) = max(y, alpha y) with 0 < alpha <= 1: the critic's first
 * layer on x^ = a real + (1 - a) fake (gan/algorithms/wgan_gp.py:41-47 feeding
 * gan/models/calciumgan.py:159-166) from that layer's outputs on real and fake -- the
 * convolution is linear, so the x^ segment needs no convolution of its own. */
int cg_lrelu_mix(const void* h_a, const void* h_b, const float* mix, void* out,
                 int B, long long n_per_sample, float alpha, void* stream);

/* ---------------------------------------------------------------------------
 * Keras Adam (gan/algorithms/optimizer.py:9; tf.keras.optimizers.Adam):
 *   g = grad*grad_scale; m = b1*m + (1-b1)g; v = b2*v + (1-b2)g^2;
 *   p -= lr_t * m / (sqrt(v) + eps), lr_t = lr*sqrt(1-b2^t)/(1-b1^t)
 * ------------------------------------------------------------------------- */
int cg_adam(float* p, const float* grad, float* m, float* v, long long n,
            float lr_t, float beta1, float beta2, float eps, float grad_scale,
            const float* lr_t_dev /* device scalar overriding lr_t, or NULL
                                     (lets a captured hipGraph vary the step) */,
            void* stream);

/* ---------------------------------------------------------------------------
 * Dynamic loss scaling of the mixed_float16 mode -- mixed_precision.
 * LossScaleOptimizer(Adam, 'dynamic') (gan/algorithms/optimizer.py:10-12,23-29):
 * the loss is multiplied by S before the backward pass, the gradients divided by
 * S before Adam; non-finite gradients skip the update and halve S (floor 1);
 * `growth_interval` consecutive finite updates double it [TF: initial 2^15,
 * interval 2000].  The state is four DEVICE floats, so a captured graph carries
 * it without host round trips:
 *   ls[0] = S, ls[1] = consecutive finite updates, ls[2] = applied Adam steps t,
 *   ls[3] = 1 while the gradients of the update in progress look finite.
 * One update = cg_grad_finite (clears ls[3] on any inf / nan; n % 4 == 0)
 *           -> cg_adam_scaled (no-op when ls[3] == 0; g = grad*grad_scale/S;
 *              lr_t from t = ls[2] + 1 as in cg_adam)
 *           -> cg_loss_scale_update (advances S / counters, sets ls[3] = 1).
 * ------------------------------------------------------------------------- */
int cg_grad_finite(const float* grad, long long n, float* ls, void* stream);
int cg_adam_scaled(float* p, const float* grad, float* m, float* v, long long n,
                   float lr, float beta1, float beta2, float eps,
                   float grad_scale, const float* ls, void* stream);
int cg_loss_scale_update(float* ls, int growth_interval, void* stream);

/* ---------------------------------------------------------------------------
 * GAN.metrics (gan/algorithms/gan.py:32-41, gan/utils/signals_metrics.py:9-28):
 * out[0..3] += sum over rows of squared differences of per-row (min, max,
 * mean, population-std) over channels of denormalised real vs fake; caller
 * zeroes out and divides by rows (ws == NULL); with ws the MEANS over rows are
 * stored (ordered reduction, see cg_reduce_ws_elems).
 * ------------------------------------------------------------------------- */
int cg_signal_metrics(const float* real /*[rows][Cr]*/,
                      const float* fake /*[rows][Cf]*/, float* out,
                      long long rows, int C, int Cr, int Cf, float smin,
                      float smax, float* ws, void* stream);

/* The scalars WGAN_GP.train returns (gan/algorithms/wgan_gp.py:82-95), gathered
 * by one launch: out[0] = gen_loss[0], out[1] = mean_k loss[2 k] (the critic loss
 * of update k; loss is f32 [n][2] as cg_critic_loss writes it), out[2] = mean_k
 * gp[k], out[3..6] = metrics[0..3]. */
int cg_step_outputs(const float* gen_loss, const float* loss, const float* gp,
                    const float* metrics, int n, float* out /*[7]*/,
                    void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CALCIUMGAN_HIP_H_ */
