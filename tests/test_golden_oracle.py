"""The oracle reproduces its committed golden vectors (tests/golden, made by
tests/make_golden.py).  Upstream parity is UNPINNED (no reference tests, no
TensorFlow); these fixtures pin the oracle against drift and give the GPU tests
a fixed target."""
import os

import numpy as np
import torch

import oracle as O

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def load_step():
  d = np.load(os.path.join(GOLD, 'wgan_gp_step_tiny.npz'))
  gw = [d['gw%02d' % i] for i in range(24)]
  dw = [d['dw%02d' % i] for i in range(12)]
  return d, gw, dw


def test_oracle_reproduces_step_golden():
  d, gw, dw = load_step()
  hp = O.make_hparams(64, 6, 8, m=2)
  gt = [torch.tensor(w) for w in gw]
  dt = [torch.tensor(w) for w in dw]
  crit = O.d_step_grads(gt, dt, torch.tensor(d['real']), torch.tensor(d['z']),
                        torch.tensor(d['alpha']), d['shifts_real'],
                        d['shifts_fake'], d['shifts_inter'], hp)
  np.testing.assert_allclose(crit['fake'].numpy(), d['fake'], rtol=1e-5,
                             atol=1e-6)
  np.testing.assert_allclose(crit['norm'].numpy(), d['norm'], rtol=1e-4)
  np.testing.assert_allclose(float(crit['gp']), d['gp'], rtol=1e-4)
  np.testing.assert_allclose(float(crit['loss']), d['dis_loss'], rtol=1e-4)
  for i, g in enumerate(crit['grads']):
    np.testing.assert_allclose(g.numpy(), d['dgrad%02d' % i], rtol=2e-3,
                               atol=1e-6)
  gen = O.g_step_grads(gt, dt, torch.tensor(d['gen_z']), d['gen_shifts'], hp)
  np.testing.assert_allclose(float(gen['loss']), d['gen_loss'], rtol=1e-4,
                             atol=1e-6)
  for i, g in enumerate(gen['grads']):
    np.testing.assert_allclose(g.numpy(), d['ggrad%02d' % i], rtol=2e-3,
                               atol=1e-7)


def test_oracle_train_reproduces_golden_losses():
  d, gw, dw = load_step()
  hp = O.make_hparams(64, 6, 8, m=2)
  gan = O.OracleGAN(hp, gw, dw)
  out = gan.train(d['real'], O.draw_randomness(hp, 4, seed=5))
  np.testing.assert_allclose(out[:3], d['train_out'], rtol=1e-4, atol=1e-6)
  np.testing.assert_allclose([out[3][k] for k in sorted(out[3])],
                             d['train_metrics'], rtol=1e-4)
  # 5 critic Adam updates + 1 generator update land on the stored weights
  for i, w in enumerate(gan.dis):
    np.testing.assert_allclose(w.numpy(), d['dw_after%02d' % i], rtol=1e-3,
                               atol=2e-5)
  for i, w in enumerate(gan.gen):
    np.testing.assert_allclose(w.numpy(), d['gw_after%02d' % i], rtol=1e-3,
                               atol=2e-5)


def test_fresh_model_loss_sanity():
  """a freshly initialised model has gp ~ 0.9-1.0 and
  dis_loss ~ 9-10 at lambda = 10."""
  hp = O.make_hparams(256, 16, 32, m=2)
  rng = np.random.RandomState(0)
  gan = O.OracleGAN(hp, O.init_generator(hp, rng), O.init_discriminator(hp, rng))
  real = rng.uniform(0, 1, (8, 256, 16)).astype(np.float32)
  r = O.draw_randomness(hp, 8, 0)['critic'][0]
  res = gan.train_discriminator(real, r)
  assert 0.7 < float(res['gp']) < 1.0
  assert 7.0 < float(res['loss']) < 10.5
