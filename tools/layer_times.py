#!/usr/bin/env python
"""Per-launch timing of the MFMA kernels inside one train() at the bench
config (HIP events around every cg_swconv / cg_wgrad call), grouped by launch
geometry, with achieved TFLOP/s per group.  Development tool (GPU only)."""
import argparse
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch

import bench
from calciumgan_amd import nets


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--batch', type=int, default=128)
  ap.add_argument('--steps', type=int, default=3)
  args = ap.parse_args()
  from calciumgan_amd.gan.algorithms import get_algorithm
  from calciumgan_amd.gan.models import get_models
  hp = bench.make_hparams(2048, 102, 64, 10)
  gen, dis = get_models(hp, None)
  gan = get_algorithm(hp, gen, dis, None)
  gan._use_graph = False
  real = torch.rand(args.batch, 2048, 102, device=gan.device)
  gan.train(real)
  recs = []
  # monkeypatch the timed launcher to keep the descriptor
  orig = nets._timed

  def timed(name, family, d, st):
    s = torch.cuda.Event(enable_timing=True)
    e = torch.cuda.Event(enable_timing=True)
    s.record()
    nets._lib.call(name, nets.ctypes.byref(d), st)
    e.record()
    recs.append((family, d, s, e))

  nets._timed = timed
  torch.cuda.synchronize()
  for _ in range(args.steps):
    gan.train(real)
  torch.cuda.synchronize()
  nets._timed = orig
  agg = collections.OrderedDict()
  for fam, d, s, e in recs:
    ms = s.elapsed_time(e)
    if fam == 'swconv':
      key = ('swconv', d.stride, d.taps, d.nB, d.Lu, d.Cx, d.N, d.CK, d.nphase,
             d.tile, d.epilogue)
      flops = 2.0 * d.nB * d.Lu * d.N * d.taps * d.Cx * d.nphase
    else:
      key = ('wgrad', d.stride, d.taps, d.nB, d.Lu, d.Cx, d.Cg, 0, 0, 0, 0)
      flops = 2.0 * d.nB * d.Lu * d.Cg * d.taps * d.Cx
    a = agg.setdefault(key, [0, 0.0, flops])
    a[0] += 1
    a[1] += ms
  tot = sum(v[1] for v in agg.values())
  print('%-7s %2s %4s %5s %5s %4s %4s %4s %2s %2s %2s | %5s %9s %8s %6s' %
        ('kind', 'R', 'taps', 'nB', 'Lu', 'Cx', 'N', 'CK', 'ph', 'sm', 'ep',
         'calls', 'avg_us', 'TF/s', 'share'))
  for k, (c, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print('%-7s %2d %4d %5d %5d %4d %4d %4d %2d %2d %2d | %5d %9.1f %8.1f %5.1f%%'
          % (k + (c, ms / c * 1e3, fl / (ms / c * 1e-3) / 1e12, 100 * ms / tot)))
  print('total MFMA-kernel ms per step: %.2f' % (tot / args.steps))


if __name__ == '__main__':
  main()
