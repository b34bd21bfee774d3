#!/usr/bin/env python
"""Where a wave of the software-pipelined conv tiles spends its cycles.

  bash tools/swp_trace.sh          (builds the -DCG_SWP_TRACE library, runs this)
  python tools/swp_trace.py R taps nB Lx Cx N tile [creal]

Needs CALCIUMGAN_HIP_LIB = a library built with -DCG_SWP_TRACE: every wave adds
up shader-clock cycles per part of the tile loop (swconv_swp.hip, CG_TR) and
cg_debug_swp_trace reads the table back.  Prints, averaged over all waves, the
cycles per part, their share of the wave's life and the cycles per K-step stage;
the same geometry is timed with the product kernel for scale (the stamps cost
the traced kernel 10-25 %)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from calciumgan_amd import _lib, geometry as geo, nets

PARTS = ['first half (reads k1 | MFMA k0)', 'counted vmcnt wait', 'stage barrier',
         'second half (reads | W DMA | MFMA k1)', 'window DMA issue',
         'prologue vmcnt(0)', 'prologue barrier + reads', 'tile set-up + DMA issue',
         'epilogue after the bias arrived', 'epilogue: bias load round trip']


def main():
  R, taps, nB, Lx, Cx, N, tile = [int(v) for v in sys.argv[1:8]]
  creal = int(sys.argv[8]) if len(sys.argv) > 8 else Cx
  dev = 'cuda'
  nphase = 2 if (R == 1 and taps > 1) else 1
  Lu = Lx // 2 if R == 2 else Lx
  Ly = Lu * nphase
  Cy = geo.pitch(N)
  CK = nets._ck_for(Cx, R, taps, Lu)
  x = torch.randn(nB, Lx, Cx, device=dev).to(torch.bfloat16)
  if creal < Cx:
    x[:, :, creal:] = 0
  W = torch.randn(taps, creal, N, device=dev)
  op = nets.PackedOperand(W, [(0, 1, creal * N, N, 1)] * nphase, creal, N, Cx, CK,
                          taps, parity_major=R == 2)
  op.repack()
  y = torch.zeros(nB, Ly, Cy, device=dev, dtype=torch.bfloat16)
  saved = nets._AUTOTUNE
  nets._AUTOTUNE = False
  d = nets._conv_desc(x, op.buf, y, nB, Lx, Cx, taps, R,
                      0 if taps == 1 else -(taps // R - 1) // 2, Lu, N, Ly, Cy,
                      CK, y_stride=nphase, bias=torch.zeros(N, device=dev),
                      epilogue=1, nphase=nphase, w_phase_stride=op.elems,
                      off_phase_step=1, yoff_phase_step=1,
                      w_parity_major=R == 2, w_narrow_last=op.narrow_last)
  nets._AUTOTUNE = saved
  d.tile, d.stage_ksteps, d.split_parity, d.ksplit = tile, 2, 0, 0
  st = nets._stream()
  lib = _lib.load()
  for _ in range(3):
    _lib.call('cg_swconv', ctypes.byref(d), st)
  torch.cuda.synchronize()
  s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  s.record()
  for _ in range(10):
    _lib.call('cg_swconv', ctypes.byref(d), st)
  e.record()
  torch.cuda.synchronize()
  us = s.elapsed_time(e) / 10 * 1e3
  nparts = 10
  tm, tn = _lib.SWP_TILES[tile]
  buf = np.zeros(1024 * 8 * nparts, np.uint32)
  lib.cg_debug_swp_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
  rc = lib.cg_debug_swp_trace(buf.ctypes.data, buf.size)
  assert rc == 0, rc
  t = buf.reshape(1024, 8, nparts).astype(np.float64)
  live = t[:, :, :10].sum(-1) > 0
  w = t[live]                       # (waves, parts)
  tot = w[:, :10].sum(-1)
  nchunks = Cx // CK
  narrow = 1 if op.narrow_last else 0
  stages_tile = (nchunks - narrow) * R * 6 + narrow * 4
  ntiles = (nB * Lu // tm) * ((N + tn - 1) // tn) * nphase
  print('conv R%d taps%d nB%d Lu%d Cx%d(%d) N%d tile %d (%dx%d): traced launch %.1f us; '
        '%d waves in %d workgroups, %d tiles, %d stages per tile'
        % (R, taps, nB, Lu, Cx, creal, N, tile, tm, tn, us, len(w), live.any(1).sum(),
           ntiles, stages_tile))
  print('  wave life %.0f cycles (min %.0f max %.0f)' % (tot.mean(), tot.min(), tot.max()))
  tiles_per_wg = ntiles / live.any(1).sum()
  for k, name in enumerate(PARTS):
    per = w[:, k].mean()
    line = '  %-40s %9.0f cycles  %5.1f %%' % (name, per, 100 * per / tot.mean())
    if k < 5:
      line += '   %6.0f per stage' % (per / (tiles_per_wg * stages_tile))
    else:
      line += '   %6.0f per tile' % (per / tiles_per_wg)
    print(line)
  print('  (16 MFMAs per stage and wave = 256 matrix-pipe cycles; x waves per SIMD)')
  # who ends the launch (round 5): life per WORKGROUP (mean of its waves), by XCD,
  # by resident slot of the CU (ids 256 apart share a CU) and by start phase
  nwg = int(live.any(1).sum())
  wl = np.array([t[i][live[i]][:, :10].sum(-1).mean() for i in range(nwg)])
  ids = np.arange(nwg)
  print('  workgroup life: mean %.0f k, min %.0f k, max %.0f k (max / mean %.3f)' % (
      wl.mean() / 1e3, wl.min() / 1e3, wl.max() / 1e3, wl.max() / wl.mean()))
  print('    by XCD (id & 7):      ', ' '.join('%4.0f' % (wl[(ids & 7) == x].mean() / 1e3)
                                               for x in range(8)))
  print('    by CU slot (id >> 8): ', ' '.join('%4.0f' % (wl[(ids >> 8) == x].mean() / 1e3)
                                               for x in range((nwg + 255) // 256)))
  print('    by id bit 0:          ', ' '.join('%4.0f' % (wl[(ids & 1) == x].mean() / 1e3)
                                               for x in range(2)))
  print('    by 32-id block:       ', ' '.join('%4.0f' % (wl[i:i + 32].mean() / 1e3)
                                               for i in range(0, nwg, 32)))
  k_loop = np.array([t[i][live[i]][:, :5].sum(-1).mean() for i in range(nwg)])
  print('    K loop part only:      mean %.0f k, min %.0f k, max %.0f k' % (
      k_loop.mean() / 1e3, k_loop.min() / 1e3, k_loop.max() / 1e3))


if __name__ == '__main__':
  main()
