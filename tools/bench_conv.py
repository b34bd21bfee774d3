#!/usr/bin/env python
"""Micro-benchmark of single cg_swconv / cg_wgrad launches (development tool).

  python tools/bench_conv.py conv  R taps nB Lx Cx N [CK] [small] [epi] [f32]
                                   [ksteps] [rowsumsq] [sp]
(stride-2 operands are packed parity-major; sp = 1: split-parity staging)
  python tools/bench_conv.py wgrad R taps nB Lx Cx Cg [nsplit] [tile_rows] [no_xcd_group] [classic_staging]
Lx is the source length; outputs Lu = Lx/2 (R=2) or Lx (R=1, 2 phases when
taps > 1)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from calciumgan_amd import _lib, geometry as geo, nets

BF16 = torch.bfloat16


def timeit(fn, iters=int(os.environ.get('BENCH_ITERS', 20))):
  # (BENCH_ITERS=8000: long enough for the power-limited clock to settle)
  fn()
  torch.cuda.synchronize()
  s = torch.cuda.Event(enable_timing=True)
  e = torch.cuda.Event(enable_timing=True)
  s.record()
  for _ in range(iters):
    fn()
  e.record()
  torch.cuda.synchronize()
  return s.elapsed_time(e) / iters * 1e-3


def main():
  kind = sys.argv[1]
  a = [int(v) for v in sys.argv[2:]]
  dev = 'cuda'
  if kind == 'conv':
    R, taps, nB, Lx, Cx, N = a[:6]
    CK = a[6] if len(a) > 6 and a[6] > 0 else None
    small = a[7] if len(a) > 7 else -1
    epi = a[8] if len(a) > 8 else 0
    f32 = a[9] if len(a) > 9 else 0
    ksteps = a[10] if len(a) > 10 else 0
    ssq = torch.zeros(nB, device=dev) if len(a) > 11 and a[11] else None
    sp = a[12] if len(a) > 12 else 0
    nphase = 2 if (R == 1 and taps > 1) else 1
    Lu = Lx // 2 if R == 2 else Lx
    Ly = Lu * (2 if nphase == 2 else 1)
    Cy = geo.pitch(N)
    if CK is None:
      CK = nets._ck_for(Cx, R, taps, Lu)
    x = torch.randn(nB, Lx, Cx, device=dev).to(BF16)
    W = torch.randn(taps, Cx, N, device=dev)
    # BENCH_CREAL=<real channels> (< Cx): zero-padded source channels, e.g. 102
    # in a 128 pitch -- the narrow-last-chunk operand of the critic's first layer
    creal = int(os.environ.get('BENCH_CREAL', Cx))
    if creal < Cx:
      x[:, :, creal:] = 0
      W = W[:, :creal].contiguous()
    op = nets.PackedOperand(W, [(0, 1, creal * N, N, 1)] * nphase, creal, N, Cx,
                            CK, taps, parity_major=R == 2)
    op.repack()
    y = torch.zeros(nB, Ly, Cy, device=dev,
                    dtype=torch.float32 if f32 else BF16)
    bias = torch.zeros(N, device=dev)
    d = nets._conv_desc(x, op.buf, y, nB, Lx, Cx, taps, R, 0 if taps == 1 else
                        -(taps // R - 1) // 2, Lu, N, Ly, Cy, CK,
                        y_stride=nphase, bias=bias, epilogue=epi,
                        mask_src=y if epi == 2 else None, out_f32=bool(f32),
                        nphase=nphase, w_phase_stride=op.elems,
                        off_phase_step=1, yoff_phase_step=1, rowsumsq=ssq,
                        w_parity_major=R == 2, w_narrow_last=op.narrow_last)
    if small >= 0:
      d.tile = small
      d.stage_ksteps = ksteps
      d.split_parity = sp
      d.ksplit = 0
    st = nets._stream()
    t = timeit(lambda: _lib.call('cg_swconv', ctypes.byref(d), st))
    fl = 2.0 * nB * Lu * N * taps * Cx * nphase
    print('conv R%d taps%d nB%d Lu%d Cx%d N%d CK%d small%d ks%d: %.1f us  %.1f TF/s'
          % (R, taps, nB, Lu, Cx, N, CK, d.tile, ksteps, t * 1e6,
             fl / t / 1e12))
  else:
    R, taps, nB, Lx, Cx, Cg = a[:6]
    nsplit = a[6] if len(a) > 6 else 0
    Lu = Lx // 2 if R == 2 else Lx
    x = torch.randn(nB, Lx, Cx, device=dev).to(BF16)
    g = torch.randn(nB, Lu, Cg, device=dev).to(BF16)
    dw = torch.zeros(taps, Cx, Cg, device=dev)
    d = nets._wgrad_desc(x, g, dw, nB, Lx, Cx, Lu, Cg, taps, R,
                         0 if taps == 1 else -(taps - 2) // 2, Cx, Cg, slot=0)
    d.nsplit = nsplit
    d.tile_rows = a[7] if len(a) > 7 else 0
    d.no_xcd_group = a[8] if len(a) > 8 else 0
    d.classic_staging = a[9] if len(a) > 9 else 0
    st = nets._stream()
    t = timeit(lambda: _lib.call('cg_wgrad', ctypes.byref(d), st))
    fl = 2.0 * nB * Lu * Cg * taps * Cx
    print('wgrad R%d taps%d nB%d Lu%d Cx%d Cg%d nsplit%d tile%d plain%d classic%d: %.1f us  %.1f TF/s'
          % (R, taps, nB, Lu, Cx, Cg, nsplit, d.tile_rows, d.no_xcd_group, d.classic_staging,
             t * 1e6, fl / t / 1e12))


if __name__ == '__main__':
  main()
