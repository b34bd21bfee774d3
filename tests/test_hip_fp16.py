"""mixed_float16 mode of the hot path (BASELINE.json configs[4]; reference
main.py:22-30, gan/algorithms/optimizer.py:10-12,23-34): fp16 activations and
MFMA operands (libcalciumgan_hip_f16.so) with dynamic loss scaling around both
optimizers, against the oracle with fp16 storage emulated.

Tolerances: fp16 keeps 11 significand bits (bf16: 8), so forward values sit
closer to the f32 oracle than in the bf16 tests: 2e-3 against the emulating
oracle, 1e-2 against f32; gradients by the same noise-floor rule as
test_hip_step._check_grads (the fixture measures the emulation's own distance
from f32).
"""
import numpy as np
import pytest
import torch

import oracle as O
from test_hip_step import CONFIGS, _check_grads, _flat

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _back_to_bf16():
  """Other test modules run on the default (bf16) build."""
  yield
  from calciumgan_amd import _lib
  _lib.use('bf16')


def _build(name, scale=None):
  from calciumgan_amd.gan.algorithms import get_algorithm
  from calciumgan_amd.gan.models import get_models
  L, C, U, k, m, B, ln = CONFIGS[name]
  hp = O.make_hparams(L, C, U, kernel_size=k, m=m, layer_norm=ln)
  hp.verbose = 0
  hp.mixed_precision = True
  gen, dis = get_models(hp, None)
  gan = get_algorithm(hp, gen, dis, None)
  assert gan.precision == 'f16' and gen.net.h_dtype == torch.float16
  rng = np.random.RandomState(42)
  gw, dw = gen.get_weights(), dis.get_weights()
  for w in gw + dw:
    if w.ndim == 1:
      w += rng.randn(*w.shape).astype(np.float32) * 0.05
  gen.set_weights(gw)
  dis.set_weights(dw)
  if scale is not None:
    gan.dis_optimizer.loss_scale_state[0] = scale
    gan.gen_optimizer.loss_scale_state[0] = scale
  real = rng.uniform(0, 1, (B, L, C)).astype(np.float32)
  return hp, gen, dis, gan, real, B


def test_loss_scale_kernels_follow_the_tf_state_machine():
  """cg_grad_finite / cg_adam_scaled / cg_loss_scale_update against
  oracle.DynamicLossScale + keras_adam: unscale inside Adam, skip + halve on a
  non-finite gradient (applied-step count untouched), double after `interval`
  finite updates, floor at 1."""
  from calciumgan_amd import _lib, nets
  _lib.use('f16')
  dev = 'cuda'
  n = 1024
  rng = np.random.RandomState(0)
  p0 = rng.randn(n).astype(np.float32)
  params = nets.FlatParams([(n,)], dev)
  params.set_weights([p0])
  ls = nets.new_loss_scale_state(dev)
  ref_p = torch.tensor(p0.copy())
  ref_m, ref_v = torch.zeros(n), torch.zeros(n)
  ref = O.DynamicLossScale(increment_period=3)
  ref_t = 0
  lr = 1e-3
  seq = ['ok', 'ok', 'inf', 'ok', 'ok', 'ok', 'nan', 'inf', 'ok']
  for kind in seq:
    g = rng.randn(n).astype(np.float32)
    S = float(ls[0])
    assert S == ref.scale
    scaled = g * S
    if kind == 'inf':
      scaled[17] = np.inf
    elif kind == 'nan':
      scaled[900] = np.nan
    params.grad.copy_(torch.tensor(scaled))
    nets.adam_update_scaled(params, lr, ls, interval=3)
    gt = torch.tensor(scaled / S)
    if ref.update([gt]):
      ref_t += 1
      O.keras_adam(ref_p, gt, ref_m, ref_v, ref_t, lr)
    torch.cuda.synchronize()
    assert float(ls[2]) == ref_t and float(ls[1]) == ref.good_steps
    assert float(ls[3]) == 1.0
    np.testing.assert_allclose(params.data.cpu().numpy(), ref_p.numpy(),
                               rtol=2e-6, atol=1e-7)
  assert float(ls[0]) == ref.scale
  # floor
  ls[0] = 1.0
  params.grad.fill_(float('inf'))
  nets.adam_update_scaled(params, lr, ls, interval=3)
  assert float(ls[0]) == 1.0


def _oracle_critic(hp, gen, dis, real, r, q):
  gw = [torch.tensor(w) for w in gen.get_weights()]
  dw = [torch.tensor(w) for w in dis.get_weights()]
  return O.d_step_grads(gw, dw, torch.tensor(real), torch.tensor(r['z']),
                        torch.tensor(r['alpha']), r['shifts_real'],
                        r['shifts_fake'], r['shifts_inter'], hp, q, q)


class _Unscaled(object):
  """grad views divided by the loss scale they were computed under."""

  def __init__(self, views, S):
    self.views, self.S = views, S

  def __iter__(self):
    return iter([v / self.S for v in self.views])


@pytest.mark.parametrize('name', ['tiny', 'mid', 'odd_c', 'long'])
def test_critic_step_fp16_matches_oracle(name):
  hp, gen, dis, gan, real, B = _build(name, scale=1024.0)
  r = O.draw_randomness(hp, B, seed=7)['critic'][0]
  emu = _oracle_critic(hp, gen, dis, real, r, O.f16_round)
  f32 = _oracle_critic(hp, gen, dis, real, r, lambda x: x)
  loss, gp = gan._train_discriminator(real, r, slot=0)
  torch.cuda.synchronize()
  st = gan._get_state(B)
  assert float(gan.dis_optimizer.loss_scale_state[2]) == 1.0, 'update skipped'
  d_out = st['dws'].d_out.cpu().numpy()
  for res, tol in ((emu, 2e-3), (f32, 1e-2)):
    np.testing.assert_allclose(d_out[:B], res['real_out'][:, 0].numpy(),
                               rtol=tol, atol=tol * 0.1)
    np.testing.assert_allclose(d_out[B:2 * B], res['fake_out'][:, 0].numpy(),
                               rtol=tol, atol=tol * 0.1)
    # the penalty's inner gradient is NOT loss-scaled (its own tape,
    # wgan_gp.py:45-48): stored in fp16 its small entries (1e-3 and below) lose
    # bits to the subnormal range, which the oracle's f32 backward does not
    # model -- 1e-2 on the norm and what derives from it
    np.testing.assert_allclose(st['norm_out'].cpu().numpy(), res['norm'].numpy(),
                               rtol=1e-2)
    np.testing.assert_allclose(float(gp), float(res['gp']), rtol=2e-2)
    np.testing.assert_allclose(float(loss), float(res['loss']), rtol=2e-2,
                               atol=1e-3)
  _check_grads(list(_Unscaled(dis.net.params.grad_views, 1024.0)), emu['grads'],
               f32['grads'], 'fp16 critic ' + name, floor=2e-2)


@pytest.mark.parametrize('name', ['tiny', 'mid', 'odd_c'])
def test_generator_step_fp16_matches_oracle(name):
  hp, gen, dis, gan, real, B = _build(name, scale=1024.0)
  r = O.draw_randomness(hp, B, seed=8)['gen']
  gw = [torch.tensor(w) for w in gen.get_weights()]
  dw = [torch.tensor(w) for w in dis.get_weights()]
  zt = torch.tensor(r['z'])
  emu = O.g_step_grads(gw, dw, zt, r['shifts'], hp, O.f16_round, O.f16_round)
  f32 = O.g_step_grads(gw, dw, zt, r['shifts'], hp)
  loss, metrics = gan._train_generator(real, r)
  torch.cuda.synchronize()
  np.testing.assert_allclose(float(loss), float(emu['loss']), rtol=2e-3,
                             atol=2e-4)
  np.testing.assert_allclose(float(loss), float(f32['loss']), rtol=1e-2,
                             atol=1e-3)
  fake = gan._get_state(B)['gws'].fake[:, :, :hp.num_channels].cpu().numpy()
  np.testing.assert_allclose(fake, emu['fake'].numpy(), atol=1e-3)
  np.testing.assert_allclose(fake, f32['fake'].numpy(), atol=4e-3)
  _check_grads(list(_Unscaled(gen.net.params.grad_views, 1024.0)), emu['grads'],
               f32['grads'], 'fp16 generator ' + name, floor=2e-2)


def test_overflowing_scale_skips_the_update_and_halves():
  """A loss scale far too large for fp16 backward tensors: the scaled chain
  overflows to infinity, LossScaleOptimizer drops the update (weights, Adam
  moments and the applied-step count stay), the scale halves; training carries
  on from the smaller scale."""
  hp, gen, dis, gan, real, B = _build('tiny', scale=2.0**40)
  w0 = _flat(dis.get_weights())
  r = O.draw_randomness(hp, B, seed=3)['critic'][0]
  gan._train_discriminator(real, r, slot=0)
  torch.cuda.synchronize()
  ls = gan.dis_optimizer.loss_scale_state.cpu().numpy()
  assert ls[0] == 2.0**39 and ls[1] == 0 and ls[2] == 0 and ls[3] == 1
  np.testing.assert_array_equal(_flat(dis.get_weights()), w0)
  assert float(dis.net.params.m.abs().max()) == 0.0
  assert gan.dis_optimizer.iterations == 0
  # a sane scale again: the next update is applied
  gan.dis_optimizer.loss_scale_state[0] = 256.0
  gan._train_discriminator(real, r, slot=0)
  torch.cuda.synchronize()
  assert gan.dis_optimizer.iterations == 1
  assert np.abs(_flat(dis.get_weights()) - w0).max() > 0


def test_train_fp16_tracks_oracle_and_replays_as_graph():
  """train() under mixed precision: three injected-randomness steps follow the
  fp16-emulating oracle (with its DynamicLossScale), then the free-running
  steps replay as hipGraphs with the loss-scale state living on the device."""
  hp, gen, dis, gan, real, B = _build('tiny', scale=512.0)
  orc = O.OracleGAN(hp, gen.get_weights(), dis.get_weights(), emulate_f16=True,
                    loss_scaling=True)
  for step in range(3):
    rand = O.draw_randomness(hp, B, seed=100 + step)
    got = gan.train(real, rand)
    ref = orc.train(real, rand)
    torch.cuda.synchronize()
    np.testing.assert_allclose([float(got[0]), float(got[1]), float(got[2])],
                               ref[:3], rtol=1e-2, atol=1e-3)
  assert gan.dis_optimizer.iterations == 15 == orc.dis_steps
  assert gan.gen_optimizer.iterations == 3 == orc.gen_steps
  for _ in range(5):
    out = gan.train(real)
  torch.cuda.synchronize()
  assert gan._get_state(B).get('graph') is not None
  assert np.isfinite([float(out[0]), float(out[1]), float(out[2])]).all()
  assert gan.dis_optimizer.iterations == 40 and gan.gen_optimizer.iterations == 8
  assert float(gan.dis_optimizer.loss_scale_state[1]) == 40.0


def test_cfg5_shapes_fp16_smoke():
  """BASELINE.json configs[4] layer shapes (L=8192, 512 neurons, num_units 64,
  m 10) in mixed_float16, small batch: two train() steps run, stay finite and
  keep or lower the loss scale."""
  from calciumgan_amd.gan.algorithms import get_algorithm
  from calciumgan_amd.gan.models import get_models
  hp = O.make_hparams(8192, 512, 64, kernel_size=24, m=10)
  hp.verbose = 0
  hp.mixed_precision = True
  gen, dis = get_models(hp, None)
  gan = get_algorithm(hp, gen, dis, None)
  real = torch.rand(8, 8192, 512, device=gan.device)
  for _ in range(2):
    out = gan.train(real)
  torch.cuda.synchronize()
  assert np.isfinite([float(out[0]), float(out[1]), float(out[2])]).all()
  assert 1.0 <= float(gan.dis_optimizer.loss_scale_state[0]) <= 2.0**15
  assert gan.gen_optimizer.iterations + gan.dis_optimizer.iterations > 0
