"""Run-to-run reproducibility of the HIP path.

The reference's arithmetic is a fixed TensorFlow graph; this path's used to
differ from run to run in the last bits because several sums met through f32
atomics (penalty norm, bias / LayerNorm / head gradients, metrics) and because
the tile tuner picks per process.  With the ordered reductions (default,
calciumgan_amd.nets.DETERMINISTIC) and a fixed tile table
(the static choice: the default since round 5, the tuner is opt-in) two
PROCESSES must produce the same bits: weights, Adam moments and every returned
scalar of every step."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(steps, shape, extra_env=None):
  env = dict(os.environ)
  env.pop('CALCIUMGAN_AUTOTUNE', None)  # the shipped default: static tiles
  env.pop('CALCIUMGAN_TILE_CACHE', None)
  env.update(extra_env or {})
  out = subprocess.run(
      [sys.executable, os.path.join(ROOT, 'tests', 'determinism_worker.py'),
       str(steps)] + [str(v) for v in shape],
      env=env, capture_output=True, text=True, timeout=900)
  assert out.returncode == 0, out.stderr[-2000:]
  return json.loads(out.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize('shape,steps', [((256, 16, 8, 32), 200),
                                         ((2048, 102, 16, 4), 200),
                                         ((2048, 102, 64, 128), 20)],
                         ids=['cfg1', 'cfg2_layers_b4', 'cfg2_full'])
def test_two_processes_train_to_identical_bits(shape, steps):
  """200 train() calls (1000 critic + 200 generator updates; two eager calls,
  then hipGraph replays) in two fresh processes: identical weights, moments and
  per-step outputs.  cfg1 is BASELINE configs[0]; the second case has cfg2's
  sequence length and neuron count (fused penalty norm, narrow channel chunk,
  split-K candidates off the table as everywhere with static tiles); the third
  is BASELINE configs[1] itself -- the benchmark's shapes, widths and batch --
  for 20 calls."""
  a = _run(steps, shape)
  b = _run(steps, shape)
  assert a['graph'] and b['graph']
  assert a['outputs'] == b['outputs'], (a['last'], b['last'])
  assert a['weights'] == b['weights']


def test_atomics_form_is_still_available_and_close():
  """CALCIUMGAN_DETERMINISTIC=0 (f32 atomics onto zeroed buffers) stays a
  working configuration: ten steps end within rounding of the ordered form."""
  a = _run(10, (256, 16, 8, 32))
  b = _run(10, (256, 16, 8, 32), {'CALCIUMGAN_DETERMINISTIC': '0'})
  import numpy as np
  np.testing.assert_allclose(a['last'], b['last'], rtol=5e-2, atol=5e-2)


@pytest.mark.parametrize('knob', [
    'CALCIUMGAN_WGRAD_PARTIALS=0', 'CALCIUMGAN_GRAPH=0', 'CALCIUMGAN_BATCH_G=0',
    'CALCIUMGAN_FUSE_LN=0', 'CALCIUMGAN_FUSE_UNSHUFFLE=0',
    'CALCIUMGAN_NARROW_LAST=0', 'CALCIUMGAN_FOLD_SCALE=0',
    'CALCIUMGAN_SWP_TILES=0', 'CALCIUMGAN_WGRAD_XCD=0', 'CALCIUMGAN_SPLIT_K=0',
    'CALCIUMGAN_WGRAD_HALVES=0',
    'CALCIUMGAN_SPLIT_SEGMENTS=1', 'CALCIUMGAN_SWP_LEAN_EPI=1',
    # round 5
    'CALCIUMGAN_WGRAD_FLEX=0', 'CALCIUMGAN_SWP_CHUNK_INNER=0',
    'CALCIUMGAN_FUSE_INTERP=0', 'CALCIUMGAN_LN_POW2=1', 'CALCIUMGAN_AUTOTUNE=1',
    'CALCIUMGAN_DEFER_FINISH=0', 'CALCIUMGAN_L1_LINEAR=0'])
def test_every_documented_switch_is_a_working_configuration(knob):
  """README's switches select older / alternative forms of the same arithmetic.
  Each must still train: the first train() call (five critic updates + one
  generator update) at cfg2's layer shapes returns what the default returns, up to
  rounding and -- for the switches that draw z in another order -- the noise of a
  batch of 4; three calls stay finite.  (Round 4 found
  CALCIUMGAN_WGRAD_PARTIALS=0 adding its atomics onto gradients the ordered mode
  no longer zeroes: dis_loss 2651 for -178 from the second critic update on.)"""
  import numpy as np
  shape = (2048, 102, 16, 4)
  name, value = knob.split('=')
  a = _run(3, shape)
  b = _run(3, shape, {name: value})
  assert np.isfinite(b['last']).all(), b
  fa, fb = np.asarray(a['first']), np.asarray(b['first'])
  # gen_loss, dis_loss, gradient penalty
  assert (np.abs(fb[:3] - fa[:3]) <= 0.25 * np.abs(fa[:3]) + 1.0).all(), (fa, fb)
  # signal metrics of the generated batch
  np.testing.assert_allclose(fb[3:], fa[3:], rtol=0.1, atol=0.02)


@pytest.mark.parametrize('knob', ['', 'CALCIUMGAN_BATCH_G=0',
                                  'CALCIUMGAN_SPLIT_SEGMENTS=1',
                                  'CALCIUMGAN_FUSE_INTERP=0', 'CALCIUMGAN_GRAPH=0',
                                  'CALCIUMGAN_FOLD_SCALE=0'])
def test_layer1_mix_under_every_schedule(knob):
  """The critic's first layer on x^ mixed from its outputs on real and fake
  (round 5) applies from 16 384 layer-1 rows per segment, so the switch tests above
  (batch 4) run the convolving plan.  Here the mix is forced at that shape
  (CALCIUMGAN_L1_LINEAR_MIN_ROWS=0) under every schedule that feeds it differently:
  the batched generator pass, one pass per update, the data-parallel segments, the
  separate interpolate + pack launches (which no longer write x^), eager launches,
  v formed by its own pass.  Against the convolving default: the first train()
  within rounding (and z-order noise), three calls finite; the separate
  interpolate + pack launches end on the same bits as the forced default."""
  import numpy as np
  shape = (2048, 102, 16, 4)
  force = {'CALCIUMGAN_L1_LINEAR_MIN_ROWS': '0'}
  a = _run(3, shape)
  extra = dict(force)
  if knob:
    name, value = knob.split('=')
    extra[name] = value
  b = _run(3, shape, extra)
  assert np.isfinite(b['last']).all(), b
  fa, fb = np.asarray(a['first']), np.asarray(b['first'])
  assert (np.abs(fb[:3] - fa[:3]) <= 0.25 * np.abs(fa[:3]) + 1.0).all(), (fa, fb)
  np.testing.assert_allclose(fb[3:], fa[3:], rtol=0.1, atol=0.02)
  if knob == 'CALCIUMGAN_FUSE_INTERP=0':
    c = _run(3, shape, force)
    assert b['outputs'] == c['outputs'], (b['last'], c['last'])
    assert b['weights'] == c['weights']


def test_fused_penalty_launch_equals_the_three_launches_bit_for_bit():
  """cg_gp_loss_scale (slot sums + gp / coef / loss + v's scale) against
  rowsumsq_finish -> cg_gp_critic_loss -> cg_scale_rows (CALCIUMGAN_FUSE_GP=0):
  the same arithmetic in the same order, so ten steps end on identical bits."""
  shape = (2048, 102, 16, 4)
  a = _run(10, shape)
  b = _run(10, shape, {'CALCIUMGAN_FUSE_GP': '0'})
  assert a['outputs'] == b['outputs'], (a['last'], b['last'])
  assert a['weights'] == b['weights']


@pytest.mark.parametrize('extra', [{}, {'CALCIUMGAN_BATCH_G': '0'},
                                   {'CALCIUMGAN_SPLIT_SEGMENTS': '1'}],
                         ids=['batched_pass', 'pass_per_update', 'dp_segments'])
def test_fused_interpolation_equals_the_separate_launches_bit_for_bit(extra):
  """cg_dense_rows_interp (round 5: the generator pass writes the critic's
  [real | fake | x^] itself -- all updates' at once from the batched pass, one
  update's from a per-update pass as under data parallelism) against
  cg_dense_rows + cg_interp_pack (CALCIUMGAN_FUSE_INTERP=0): the same arithmetic
  on the same draws, so ten steps at cfg2's layer shapes end on identical bits."""
  shape = (2048, 102, 16, 4)
  a = _run(10, shape, dict(extra))
  b = _run(10, shape, dict(extra, CALCIUMGAN_FUSE_INTERP='0'))
  assert a['outputs'] == b['outputs'], (a['last'], b['last'])
  assert a['weights'] == b['weights']


def test_deferred_finishing_launches_equal_the_separate_ones_bit_for_bit():
  """cg_finish_defer / cg_finish_flush (round 5: the seven finishing launches of
  the generator backward's ordered reductions as one) against one finishing launch
  per reduction (CALCIUMGAN_DEFER_FINISH=0): the same sums in the same order."""
  shape = (2048, 102, 16, 4)
  a = _run(10, shape)
  b = _run(10, shape, {'CALCIUMGAN_DEFER_FINISH': '0'})
  assert a['outputs'] == b['outputs'], (a['last'], b['last'])
  assert a['weights'] == b['weights']
