#!/usr/bin/env python
"""Micro-benchmark of the HBM-bound kernels at the cfg2 shapes (development
tool): prints us per launch and achieved TB/s against the algorithmic bytes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from calciumgan_amd import _lib, nets

BF16 = torch.bfloat16
P = nets._p


def timeit(fn, iters=30):
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
  dev = 'cuda'
  st = nets._stream()
  B = int(os.environ.get('BENCH_B', 128))
  rws = torch.empty(_lib.load().cg_reduce_ws_elems(), device=dev)
  # generator LayerNorm layers: (rows, C, Cp)
  for L, C in ((128, 320), (256, 256), (512, 192), (1024, 128), (2048, 102)):
    rows, Cp = B * L, max(32, (C + 31) // 32 * 32)
    y = torch.randn(rows, Cp, device=dev).to(BF16)
    h = torch.empty_like(y)
    g = torch.ones(C, device=dev)
    b = torch.zeros(C, device=dev)
    mean = torch.empty(rows, device=dev)
    rstd = torch.empty(rows, device=dev)
    t = timeit(lambda: _lib.call('cg_ln_lrelu_fwd', P(y), P(g), P(b), P(h),
                                 P(mean), P(rstd), rows, C, Cp, 1e-3, 0.3, st))
    nbytes = rows * Cp * 4 + rows * 8
    print('ln_fwd rows %7d C %3d: %6.1f us  %.2f TB/s' % (rows, C, t * 1e6,
                                                          nbytes / t / 1e12))
    dh = torch.randn(rows, Cp, device=dev).to(BF16)
    dy = torch.empty_like(y)
    dg = torch.zeros(C, device=dev)
    db = torch.zeros(C, device=dev)
    dbias = torch.zeros(C, device=dev)
    nbytes = rows * Cp * 8 + rows * 8
    for label, ws in (('atomics', None), ('ordered', rws)):
      t = timeit(lambda: _lib.call('cg_ln_lrelu_bwd', P(dh), P(h), P(y), P(mean),
                                   P(rstd), P(g), P(dy), P(dg), P(db), P(dbias),
                                   rows, C, Cp, 0.3, P(ws), st))
      print('ln_bwd (%s) rows %7d C %3d: %6.1f us  %.2f TB/s' % (
          label, rows, C, t * 1e6, nbytes / t / 1e12))
  # discriminator unshuffle + mask (3B batch)
  for w, C in ((1024, 64), (512, 128), (256, 192), (128, 256)):
    nB = 3 * B
    e = torch.randn(nB, w, C, device=dev).to(BF16)
    h = torch.randn(nB, w, C, device=dev).to(BF16)
    d = torch.empty_like(e)
    sh = torch.tensor([3, -7, 10], dtype=torch.int32, device=dev)
    t = timeit(lambda: _lib.call('cg_unshuffle_mask', P(e), P(h), P(d), P(sh), nB,
                                 w, C, B, 0.3, st))
    nbytes = nB * w * C * 6
    print('unshuffle nB %d w %4d C %3d: %6.1f us  %.2f TB/s' % (
        nB, w, C, t * 1e6, nbytes / t / 1e12))
  # generator-step signal metrics and the output Dense bias gradient
  rows, C, Cp = B * 2048, 102, 128
  real = torch.rand(rows, C, device=dev)
  fake = torch.rand(rows, Cp, device=dev)
  out = torch.zeros(4, device=dev)
  dz = torch.randn(rows, Cp, device=dev).to(BF16)
  cs = torch.zeros(C, device=dev)
  for label, ws in (('atomics', None), ('ordered', rws)):
    t = timeit(lambda: _lib.call('cg_signal_metrics', P(real), P(fake), P(out),
                                 rows, C, C, Cp, 0.0, 1.0, P(ws), st))
    print('signal_metrics (%s) rows %d: %6.1f us  %.2f TB/s' % (
        label, rows, t * 1e6, rows * (C + Cp) * 4 / t / 1e12))
    t = timeit(lambda: _lib.call('cg_colsum', P(dz), P(cs), rows, C, Cp, P(ws),
                                 st))
    print('colsum (%s) rows %d: %6.1f us  %.2f TB/s' % (
        label, rows, t * 1e6, rows * Cp * 2 / t / 1e12))


if __name__ == '__main__':
  main()
