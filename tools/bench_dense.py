#!/usr/bin/env python
"""Micro-benchmark of the streaming per-timestep Dense kernels (development tool):
  python tools/bench_dense.py rows Cx N [f16]
times cg_dense_rows (f32 out + sigmoid) and cg_dense_rows_act (act-typed out)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from calciumgan_amd import _lib, geometry as geo, nets

rows, Cx, N = (int(v) for v in sys.argv[1:4])
if len(sys.argv) > 4:
  _lib.use('f16')
dt = nets.act_dtype()
x = torch.randn(rows, Cx, device='cuda').to(dt)
W = torch.randn(Cx, N, device='cuda') * 0.05
op = nets.PackedOperand(W, [(0, 1, 0, N, 1)], Cx, N, Cx, 32, 1)
op.repack()
cf = (N + 7) // 8 * 8
y = torch.zeros(rows, cf, device='cuda')
ya = torch.zeros(rows, geo.pitch(N), device='cuda', dtype=dt)
b = torch.zeros(N, device='cuda')
st = nets._stream()


def timeit(fn, iters=10):
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


fl = 2.0 * rows * Cx * N
t = timeit(lambda: _lib.call('cg_dense_rows', nets._p(x), nets._p(op.buf), nets._p(b),
                             nets._p(y), rows, Cx, N, cf, 3, st))
print('dense_rows     rows %d Cx %d N %d: %.1f us  %.1f TF/s  %.2f TB/s' % (
    rows, Cx, N, t * 1e6, fl / t / 1e12, rows * (Cx * 2 + cf * 4) / t / 1e12))
if Cx >= 128:
  t = timeit(lambda: _lib.call('cg_dense_rows_act', nets._p(x), nets._p(op.buf),
                               nets._p(ya), rows, Cx, N, geo.pitch(N), st))
  print('dense_rows_act rows %d Cx %d N %d: %.1f us  %.1f TF/s  %.2f TB/s' % (
      rows, Cx, N, t * 1e6, fl / t / 1e12,
      rows * (Cx * 2 + geo.pitch(N) * 2) / t / 1e12))
g = torch.randn(rows, geo.pitch(N), device='cuda').to(dt)
dw = torch.zeros(Cx, N, device='cuda')
need = _lib.load().cg_dense_wgrad_ws_elems(rows, Cx, N)
ws = torch.empty(max(need, 1), device='cuda')
t = timeit(lambda: _lib.call('cg_dense_wgrad', nets._p(x), nets._p(g), nets._p(dw), rows,
                             Cx, geo.pitch(N), Cx, N, nets._p(ws), need, st))
print('dense_wgrad    rows %d Cx %d N %d: %.1f us  %.1f TF/s  %.2f TB/s' % (
    rows, Cx, N, t * 1e6, fl / t / 1e12, rows * (Cx * 2 + geo.pitch(N) * 2) / t / 1e12))
