// Kernel-side argument block of the cg_swconv launches (filled on the host by
// swconv_run in swconv.hip from a cg_conv_desc; shared by the tile kernels of
// swconv.hip and the ping-pong kernels of swconv_pp.hip).
#pragma once
#include "cg_common.h"

struct ConvArgs {
  const uint16_t* x;
  const uint16_t* w;
  void* y;
  const float* bias;
  const uint16_t* mask;
  const int* shifts;
  float* rowsumsq;
  int nB, Lx, Cx, seg_size;
  int taps, off, Lu, M;
  int N, Ly, Cy, y_stride, y_off;
  int CK, c8, nchunks, Fp, nstages;
  long long Kpack;
  int pitchA, S, log2S, nseg, WR, ldsA_elems;
  int epilogue, out_f32;
  float alpha;
  float inv_c8;
  int log2c8;  // log2(c8) when c8 is a power of two, else -1
  long long w_phase_stride;
  int off_phase_step, yoff_phase_step;
  int gm, gn, gp;  // logical grid: row tiles, column tiles, phases
  int pmajor;      // stride 2: weights packed even taps first, then odd taps
  // CG_EPI_LN_LRELU (128-column tiles, N <= 128)
  const float* ln_gamma;
  const float* ln_beta;
  uint16_t* ln_h;
  float* ln_mean;
  float* ln_rstd;
  float ln_eps;
  int narrow;      // last channel chunk packed narrow (cg_pack_desc.narrow_last)
  int ksplit;      // > 1: blockIdx.y walks its share of the channel chunks
  long long split_stride;  // f32 elements between the splits' partial outputs
  // output-side phase-shuffle adjoint (see cg_conv_desc.out_shifts)
  const int* out_shifts;
  int out_seg;
  uint16_t* side;
  int side_rows;
  const float* row_scale;  // per-sample scale in front of the epilogue, or null
  // ordered penalty norm (cg_conv_desc.rowsumsq_ws): slot
  // ((row tile in the sample) * gp + phase) * gn + column tile of sample b's
  // ssq_P slots receives the workgroup's sum with a plain store
  float* ssq_ws;
  int ssq_P;
};

// swconv_swp.hip: launch (or, dry, only validate) a software-pipelined tile of
// wm x wn waves with (16 mt) x 64 wave tiles.  `a` is the block swconv_run
// filled for a row tile of wm * mt * 16 rows; returns 0 / CG_EINVAL / a
// hipError_t.
int swconv_swp_launch(const ConvArgs& a, int stride, int wm, int wn, int mt,
                      int ksplit, bool dry, hipStream_t stream);

