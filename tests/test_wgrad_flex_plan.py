"""Host logic of the flex form of cg_wgrad_batched (calciumgan_amd/csrc/wgrad.hip,
plan_flex; round 5): the share plan of a backward pass is inspected through
cg_wgrad_flex_plan -- no device call, so this runs without a GPU.

The weight gradients replace what tape.gradient computes for the Conv1D /
Conv1DTranspose kernels in gan/algorithms/optimizer.py:31-34 (autodiff of
gan/models/calciumgan.py:145-185); the plan only decides WHICH workgroup sums
WHICH (layer, output tile, K' range), so the properties checked are exact:
every K' tile of every output tile is summed exactly once, the partial sums of
an output tile sit in consecutive slots in ascending K' order (the reducing
launch's fixed order), the members of a team walk identical K' ranges (operand
tiles are shared in one XCD's L2), and the shares are balanced."""
import ctypes

import numpy as np
import pytest

from calciumgan_amd import _lib
from calciumgan_amd import geometry as geo

K = 24
ITEMS, INTS = 8, 6


def _desc(nB, Lx, Cin, Cout):
  """Conv1D(k=24, s=2) weight-gradient descriptor with placeholder pointers (the
  planner reads geometry only)."""
  d = _lib.WgradDesc()
  d.x = d.g = d.dw = 0x1000
  d.partials, d.partials_elems = 0x1000, 1 << 40
  d.nB, d.Lx, d.Cx, d.seg_size = nB, Lx, geo.pitch(Cin), 1
  d.Lu, d.Cg = Lx // 2, geo.pitch(Cout)
  d.taps, d.stride, d.off = K, 2, -11
  d.Cx_real, d.Cg_real = Cin, Cout
  d.store = 1
  return d


def _critic(nB, L, C, U):
  chans = [C, U, 2 * U, 3 * U, 4 * U, 5 * U]
  descs = [_desc(nB, L >> i, chans[i], chans[i + 1]) for i in range(5)]
  for d in descs[1:]:  # PhaseShuffle in front of layers 2-5 (calciumgan.py:117-138)
    d.shifts, d.seg_size = 0x1000, max(1, nB // 3)
  return descs


def _generator(nB, L, C, U):
  # Conv1DTranspose weight gradient: "x" is the (longer) output gradient, "g" the input
  chans = [5 * U, 4 * U, 3 * U, 2 * U, U, C]
  return [_desc(nB, L >> (4 - i), chans[i + 1], chans[i]) for i in range(5)]


def _plan(descs, mode):
  lib = _lib.load()
  arr = (_lib.WgradDesc * len(descs))(*descs)
  info = (ctypes.c_int * 16)()
  n = lib.cg_wgrad_flex_plan(arr, len(descs), mode, None, 0, info)
  if n < 0:
    return None
  out = (ctypes.c_int * n)()
  assert lib.cg_wgrad_flex_plan(arr, len(descs), mode, out, n, info) == n
  return np.array(out[:], dtype=np.int64), list(info)


def _check(descs, mode):
  res = _plan(descs, mode)
  assert res is not None
  tab, info = res
  S, nteams, nwg, nitems = info[:4]
  items = tab[:nwg * ITEMS * INTS].reshape(nwg, ITEMS, INTS)
  grids = [((d.Cx_real + 31) // 32, (d.Cg_real + 63) // 64) for d in descs]
  ntiles = [d.nB * d.Lu // (128 if d.Lu % 128 == 0 else 64) for d in descs]
  per_tile = {}  # (layer, bx, by) -> [(k0, kn, slot)]
  cost = np.zeros(nwg)
  count = 0
  for w in range(nwg):
    ended = False
    for li, bx, by, k0, kn, slot in items[w]:
      if li < 0:
        ended = True
        continue
      assert not ended, 'items of a workgroup are a prefix of its row'
      assert 0 <= li < len(descs) and 0 <= bx < grids[li][0] and 0 <= by < grids[li][1]
      assert kn > 0 and k0 >= 0 and k0 + kn <= ntiles[li]
      per_tile.setdefault((li, bx, by), []).append((k0, kn, slot))
      # (the planner's costs: 64 per 128-row tile -- 62 without a PhaseShuffle in
      # front of the layer --, 38 per 64-row tile)
      cost[w] += kn * ((64 if descs[li].shifts else 62) if descs[li].Lu % 128 == 0
                       else 38)
      count += 1
  assert count == nitems
  slots = [0] * len(descs)
  for li, (gx, gy) in enumerate(grids):
    tl = tab[info[10 + li]:info[10 + li] + 2 * gx * gy].reshape(gy * gx, 2)
    for by in range(gy):
      for bx in range(gx):
        its = sorted(per_tile[(li, bx, by)])
        # every K' tile exactly once ...
        pos = 0
        for k0, kn, _ in its:
          assert k0 == pos, 'gap or overlap in the K\' cover'
          pos += kn
        assert pos == ntiles[li]
        # ... and the tile's slots consecutive, in ascending K' order
        p0, cnt = tl[by * gx + bx]
        assert cnt == len(its)
        assert [s for _, _, s in its] == list(range(p0, p0 + cnt))
        slots[li] += cnt
    assert slots[li] == info[4 + li]
    # the slots of a layer do not overlap
    allslots = sorted(s for (l, _, _), its in per_tile.items() if l == li
                      for _, _, s in its)
    assert allslots == list(range(slots[li]))
  # team members (same XCD residue, consecutive slots) walk the same K' ranges
  for t in range(nteams):
    ids = [((t >> 3) * S + m) * 8 + (t & 7) for m in range(S)]
    ranges = [[(int(r[0]), int(r[3]), int(r[4])) for r in items[w] if r[0] >= 0]
              for w in ids]
    live = [r for r in ranges if r]
    for r in live[1:]:
      # (a member without a tile in some column skips that column)
      assert set(r) <= set(live[0]) or set(live[0]) <= set(r)
  return S, nteams, nwg, nitems, cost, info


def test_cfg2_critic_pass_plan():
  """BASELINE configs[1], critic backward: 3 x 128 samples, five layers."""
  S, nteams, nwg, nitems, cost, info = _check(_critic(384, 2048, 102, 64), 1)
  assert (S, nteams, nwg) == (4, 64, 256)
  # ~1.4 accumulator flushes per workgroup instead of 3 (halves) / 5 (plain):
  # partial sums of a pass <= 80 MB (was 148 MB)
  assert nitems <= 400
  assert nitems * 3 * 8 * 2048 * 4 <= 80e6
  # balanced to a few K' tiles
  assert cost.max() <= 1.03 * cost.mean()
  assert cost.min() >= 0.97 * cost.mean()


def test_cfg2_generator_pass_plan():
  S, nteams, nwg, nitems, cost, _ = _check(_generator(128, 2048, 102, 64), 1)
  assert nwg == 256
  live = cost[cost > 0]
  assert live.max() <= 1.06 * live.mean()


def test_cfg5_critic_pass_plan():
  """BASELINE configs[4]: L = 8192, 512 neurons, 3 x 256 samples."""
  S, nteams, nwg, nitems, cost, _ = _check(_critic(768, 8192, 512, 64), 1)
  assert nwg == 256
  assert cost.max() <= 1.03 * cost.mean()


@pytest.mark.parametrize('nB,L,C,U', [(6, 1024, 40, 24), (4, 256, 16, 8),
                                      (12, 512, 102, 16), (2, 2048, 102, 64)])
def test_small_launches_plan_in_forced_mode(nB, L, C, U):
  """Mode 2 (tests) plans whatever the size, with fewer teams; mode 1 declines a
  launch whose shares would not be worth their set-up."""
  descs = [d for d in _critic(nB, L, C, U) if d.Lu % 64 == 0][:4]
  if len(descs) < 2:
    pytest.skip('no two ring-form layers at this shape')
  _check(descs, 2)


def test_mode1_declines_tiny_launches():
  descs = _critic(2, 512, 16, 8)[:2]
  assert _plan(descs, 1) is None
  assert _plan(descs, 2) is not None
