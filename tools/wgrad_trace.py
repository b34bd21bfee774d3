#!/usr/bin/env python
"""Where a wave of the batched cg_wgrad launch spends its cycles (critic pass
at the benchmark's shapes).  Needs CALCIUMGAN_HIP_LIB = a library built with
-DCG_WGRAD_TRACE (tools/wgrad_trace.sh)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench
from calciumgan_amd import _lib
from calciumgan_amd.gan.algorithms import get_algorithm
from calciumgan_amd.gan.models import get_models

PARTS = ['item set-up + ring fill + first fragments', "K' loop",
         'bias column sums + accumulator flush', 'items without a share',
         'barrier between items']


def main():
  hp = bench.make_hparams(2048, 102, 64, 10, False)
  hp.verbose = 0
  gen, dis = get_models(hp, None)
  gan = get_algorithm(hp, gen, dis, None)
  real = torch.rand(128, 2048, 102, device=gan.device)
  for _ in range(3):
    gan._critic_compute(real)
  torch.cuda.synchronize()
  lib = _lib.load()
  n = 5
  buf = np.zeros(256 * 8 * n, np.uint32)
  lib.cg_debug_wgrad_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
  assert lib.cg_debug_wgrad_trace(buf.ctypes.data, buf.size) == 0
  t = buf.reshape(256, 8, n).astype(np.float64)
  w = t[t.sum(-1) > 0]
  tot = w.sum(-1)
  print('batched cg_wgrad, critic pass (5 layers): %d waves, wave life %.0f cycles '
        '(min %.0f max %.0f)' % (len(w), tot.mean(), tot.min(), tot.max()))
  for k, name in enumerate(PARTS):
    print('  %-44s %9.0f cycles  %5.1f %%   %7.0f per layer'
          % (name, w[:, k].mean(), 100 * w[:, k].mean() / tot.mean(), w[:, k].mean() / 5))
  per_wg(t)
  per_layer(gan, t)


def per_wg(t):
  life = t.sum(-1).mean(-1)  # per workgroup
  print('  wave life by workgroup id (mean of 16 ids, k cycles):')
  print('   ', ' '.join('%4.0f' % (life[i:i + 16].mean() / 1e3) for i in range(0, 256, 16)))
  kl = t[:, :, 1].mean(-1)
  print("  K' loop only:")
  print('   ', ' '.join('%4.0f' % (kl[i:i + 16].mean() / 1e3) for i in range(0, 256, 16)))


def per_layer(gan, t):
  """Flex form (round 5): the wave life of the workgroups by the layer their
  share lies in (workgroups whose items are all of one layer), and the K' loop
  cycles per K' tile of that layer -- what the planner's per-tile costs (16 per
  128-row tile, 9 per 64-row tile) should be proportional to."""
  lib = _lib.load()
  descs = gan._get_state(128)['critic'].wgrad
  arr = (_lib.WgradDesc * len(descs))(*descs)
  info = (ctypes.c_int * 16)()
  n = lib.cg_wgrad_flex_plan(arr, len(descs), 1, None, 0, info)
  if n < 0:
    print('  (not the flex form)')
    return
  out = (ctypes.c_int * n)()
  lib.cg_wgrad_flex_plan(arr, len(descs), 1, out, n, info)
  nwg = info[2]
  items = np.array(out[:nwg * 48]).reshape(nwg, 8, 6)
  life = t.sum(-1).mean(-1)
  kloop = t[:, :, 1].mean(-1)
  print('  flex form: teams of %d, %d items; by the layer a workgroup\'s share lies in:' % (
      info[0], info[3]))
  for li in range(len(descs)):
    sel, tiles = [], []
    for w in range(min(nwg, 256)):
      its = [r for r in items[w] if r[0] >= 0]
      if its and all(r[0] == li for r in its):
        sel.append(w)
        tiles.append(sum(int(r[4]) for r in its))
    if not sel:
      continue
    sel = np.array(sel)
    print('    layer %d: %3d workgroups, life %6.0f k cycles (min %.0f max %.0f), K\' loop '
          '%6.0f k = %5.0f cycles per K\' tile (%d tiles each)' % (
              li + 1, len(sel), life[sel].mean() / 1e3, life[sel].min() / 1e3,
              life[sel].max() / 1e3, kloop[sel].mean() / 1e3,
              kloop[sel].mean() / np.mean(tiles), int(np.mean(tiles))))


  # the workgroups of cx block 0 also sum the bias columns of their g tiles
  b0 = [w for w in range(min(nwg, 256)) if any(r[0] >= 0 and r[1] == 0 for r in items[w])]
  bn = [w for w in range(min(nwg, 256)) if any(r[0] >= 0 for r in items[w]) and w not in b0]
  print('    workgroups with a cx-block-0 item (bias column sums): %d, life %.0f k; the others: %d, '
        'life %.0f k' % (len(b0), life[b0].mean() / 1e3, len(bn), life[bn].mean() / 1e3))
  # leader effect: life by the workgroup's position in its team (slot (id >> 3) % S),
  # all workgroups / only those without a cx-block-0 item
  S = info[0]
  pos = (np.arange(min(nwg, 256)) >> 3) % S
  print('    life by team position (k cycles), all:', ' '.join(
      '%4.0f' % (life[:len(pos)][pos == m].mean() / 1e3) for m in range(S)),
        '| without bias sums:', ' '.join(
            '%4.0f' % (np.mean([life[w] for w in bn if pos[w] == m] or [0]) / 1e3)
            for m in range(S)))
  # who is slow: by XCD (id & 7), and the eight longest-lived workgroups
  ids = np.arange(min(nwg, 256))
  print('    life by XCD (k cycles):', ' '.join(
      '%4.0f' % (life[ids[(ids & 7) == x]].mean() / 1e3) for x in range(8)))
  for w in np.argsort(-life[:min(nwg, 256)])[:8]:
    its = [tuple(int(v) for v in r) for r in items[w] if r[0] >= 0]
    print('    slowest: wg %3d (xcd %d) life %4.0f k  items (layer, bx, by, k0, kn, slot) %s' % (
        w, w & 7, life[w] / 1e3, its))


if __name__ == '__main__':
  main()
