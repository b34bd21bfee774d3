"""Index-level known-answer tests that pin the oracle's layer semantics to the
definitions in SURVEY.md Appendix A (the reference has no tests of its own:
PARITY UNPINNED upstream).  Every expected value here is computed by explicit
python/numpy loops written from the formulas, independent of torch's conv."""
import numpy as np
import pytest
import torch

import oracle as O


def _conv_same_loops(x, w, b, s):
  B, L, Ci = x.shape
  k, _, Co = w.shape
  Lo = -(-L // s)
  total = max((Lo - 1) * s + k - L, 0)
  left = total // 2
  y = np.zeros((B, Lo, Co), np.float64)
  for t in range(Lo):
    for kk in range(k):
      i = s * t + kk - left
      if 0 <= i < L:
        y[:, t, :] += x[:, i, :] @ w[kk]
  return y + b


def _convT_same_loops(x, w, b, s):
  B, L, Ci = x.shape
  k, _, Co, _ = w.shape
  Lo = L * s
  total = max((L - 1) * s + k - Lo, 0)
  left = total // 2
  y = np.zeros((B, Lo, Co), np.float64)
  for i in range(L):
    for kk in range(k):
      o = s * i + kk - left
      if 0 <= o < Lo:
        y[:, o, :] += x[:, i, :] @ w[kk, 0].T  # (Ci)->(Co): W[kk,0,co,ci]
  return y + b


@pytest.mark.parametrize('L,k,s', [(32, 24, 2), (16, 8, 2), (10, 5, 2),
                                   (9, 4, 1)])
def test_conv1d_same_matches_definition(L, k, s):
  rng = np.random.RandomState(1)
  x = rng.randn(2, L, 3)
  w = rng.randn(k, 3, 5)
  b = rng.randn(5)
  got = O.conv1d_same(
      torch.tensor(x), torch.tensor(w), torch.tensor(b), s).numpy()
  np.testing.assert_allclose(got, _conv_same_loops(x, w, b, s), atol=1e-10)


def test_same_padding_k24_s2_is_11_11():
  assert O.same_padding(2048, 24, 2) == (1024, 11, 11)
  assert O.same_padding(64, 24, 2) == (32, 11, 11)


@pytest.mark.parametrize('L,k,s', [(16, 24, 2), (8, 8, 2), (4, 24, 2)])
def test_conv1d_transpose_matches_definition(L, k, s):
  rng = np.random.RandomState(2)
  x = rng.randn(2, L, 3)
  w = rng.randn(k, 1, 5, 3)
  b = rng.randn(5)
  got = O.conv1d_transpose_same(
      torch.tensor(x), torch.tensor(w), torch.tensor(b), s).numpy()
  assert got.shape == (2, L * s, 5)
  np.testing.assert_allclose(got, _convT_same_loops(x, w, b, s), atol=1e-10)


def test_conv_transpose_is_adjoint_of_conv():
  """Conv2DTranspose 'same' is the input-gradient of the 'same' conv."""
  rng = np.random.RandomState(3)
  k, s, Ci, Co, L = 24, 2, 3, 4, 32
  w = torch.tensor(rng.randn(k, Ci, Co))
  x = torch.tensor(rng.randn(1, L, Ci))
  dy = torch.tensor(rng.randn(1, L // s, Co))
  lhs = (O.conv1d_same(x, w, None, s) * dy).sum()
  # transpose kernel layout (k,1,Co_T,Ci_T) with Co_T=Ci, Ci_T=Co
  wt = w.permute(0, 1, 2)[:, None]  # (k,1,Ci,Co): out channels=Ci, in=Co
  rhs = (O.conv1d_transpose_same(dy, wt, None, s) * x).sum()
  np.testing.assert_allclose(lhs.item(), rhs.item(), rtol=1e-10)


def test_phase_shuffle_known_answers():
  # w=6, values 0..5; tf.pad reflect excludes the edge sample
  x = torch.arange(6.0).reshape(1, 6, 1)
  np.testing.assert_array_equal(
      O.phase_shuffle(x, 2)[0, :, 0].numpy(), [2, 3, 4, 5, 4, 3])
  np.testing.assert_array_equal(
      O.phase_shuffle(x, -2)[0, :, 0].numpy(), [2, 1, 0, 1, 2, 3])
  np.testing.assert_array_equal(
      O.phase_shuffle(x, 0)[0, :, 0].numpy(), [0, 1, 2, 3, 4, 5])
  np.testing.assert_array_equal(
      O.phase_shuffle(x, 1)[0, :, 0].numpy(), [1, 2, 3, 4, 5, 4])
  np.testing.assert_array_equal(
      O.phase_shuffle(x, -1)[0, :, 0].numpy(), [1, 0, 1, 2, 3, 4])


@pytest.mark.parametrize('shift', range(-5, 6))
def test_phase_shuffle_equals_reflect_pad_and_slice(shift):
  """Restates calciumgan.py:126-137 literally with torch's reflect pad."""
  w = 16
  x = torch.randn(2, w, 3, dtype=torch.float64)
  xt = x.transpose(1, 2)
  if shift > 0:
    ref = torch.nn.functional.pad(xt, (0, shift), mode='reflect')[:, :,
                                                                  shift:w +
                                                                  shift]
  else:
    ref = torch.nn.functional.pad(xt, (-shift, 0), mode='reflect')[:, :, 0:w]
  np.testing.assert_array_equal(
      O.phase_shuffle(x, shift).numpy(), ref.transpose(1, 2).numpy())


def test_layer_norm_known_answer():
  x = torch.tensor([[[1.0, 2.0, 3.0, 6.0]]], dtype=torch.float64)
  g = torch.tensor([1.0, 2.0, 1.0, 1.0], dtype=torch.float64)
  b = torch.tensor([0.0, 0.0, 1.0, 0.0], dtype=torch.float64)
  mean, var = 3.0, (4 + 1 + 0 + 9) / 4.0
  exp = (np.array([1, 2, 3, 6.0]) - mean) / np.sqrt(var + 1e-3)
  exp = exp * g.numpy() + b.numpy()
  np.testing.assert_allclose(
      O.layer_norm(x, g, b)[0, 0].numpy(), exp, rtol=1e-12)


def test_leaky_relu_alpha():
  x = torch.tensor([-2.0, 0.0, 3.0])
  np.testing.assert_allclose(O.leaky_relu(x).numpy(), [-0.6, 0.0, 3.0])


def test_param_counts_match_survey():
  """cfg1 G 1 091 456 / D 997 089; cfg2 G 4 375 740 /
  D 4 110 273."""
  rng = np.random.RandomState(0)
  hp = O.make_hparams(256, 16, 32)
  assert O.count_params(O.init_generator(hp, rng)) == 1091456
  assert O.count_params(O.init_discriminator(hp, rng)) == 997089
  hp = O.make_hparams(2048, 102, 64)
  g, d = O.init_generator(hp, rng), O.init_discriminator(hp, rng)
  assert O.count_params(g) == 4375740 and len(g) == 24
  assert O.count_params(d) == 4110273 and len(d) == 12


def test_noise_shape_requires_divisibility():
  assert O.calculate_noise_shape((2048, 102), 32, 5, 2) == (64, 32)
  with pytest.raises(ValueError):
    O.calculate_noise_shape((100, 4), 32, 5, 2)


def test_discriminator_flatten_is_time_major():
  """Flatten index t*C+c: a dense kernel that picks flat index j
  must read activation [t=j//C, c=j%C]."""
  hp = O.make_hparams(64, 3, 1, m=0)
  rng = np.random.RandomState(5)
  d = [torch.tensor(w, dtype=torch.float64) for w in O.init_discriminator(hp, rng)]
  x = torch.tensor(rng.rand(1, 64, 3))
  # recompute layer-5 activation by hand
  h = x
  for l in range(5):
    h = O.leaky_relu(O.conv1d_same(h, d[2 * l], d[2 * l + 1], 2))
  assert h.shape == (1, 2, 5)
  for j in range(10):
    dw = torch.zeros(10, 1, dtype=torch.float64)
    dw[j, 0] = 1.0
    dd = d[:10] + [dw, torch.zeros(1, dtype=torch.float64)]
    out = O.discriminator_forward(dd, x, [0, 0, 0, 0], hp)
    np.testing.assert_allclose(out.item(), h[0, j // 5, j % 5].item())


def test_keras_adam_known_answer():
  """One and two steps by hand (eps outside the bias correction)."""
  p = torch.tensor([1.0], dtype=torch.float64)
  m = torch.zeros(1, dtype=torch.float64)
  v = torch.zeros(1, dtype=torch.float64)
  g = torch.tensor([0.5], dtype=torch.float64)
  O.keras_adam(p, g, m, v, 1, 1e-3)
  lr_t = 1e-3 * np.sqrt(1 - 0.999) / (1 - 0.9)
  exp1 = 1.0 - lr_t * 0.05 / (np.sqrt(0.00025) + 1e-7)
  np.testing.assert_allclose(p.item(), exp1, rtol=1e-12)
  g2 = torch.tensor([-1.0], dtype=torch.float64)
  O.keras_adam(p, g2, m, v, 2, 1e-3)
  m2 = 0.9 * 0.05 + 0.1 * -1.0
  v2 = 0.999 * 0.00025 + 0.001 * 1.0
  lr_2 = 1e-3 * np.sqrt(1 - 0.999**2) / (1 - 0.9**2)
  np.testing.assert_allclose(
      p.item(), exp1 - lr_2 * m2 / (np.sqrt(v2) + 1e-7), rtol=1e-12)


def test_gradient_penalty_matches_finite_differences():
  """GP = mean((||dD/dx||-1)^2): check the inner gradient against central
  differences of D (float64) on a tiny model."""
  hp = O.make_hparams(32, 2, 1, kernel_size=4, m=1)
  rng = np.random.RandomState(7)
  d = [torch.tensor(w, dtype=torch.float64) for w in O.init_discriminator(hp, rng)]
  real = torch.tensor(rng.rand(2, 32, 2))
  fake = torch.tensor(rng.rand(2, 32, 2))
  alpha = torch.tensor([0.25, 0.75], dtype=torch.float64)
  shifts = [1, -1, 0, 1]
  gp, norm, grad = O.gradient_penalty(d, real, fake, alpha, shifts, hp)
  inter = O.interpolation(real, fake, alpha)
  eps = 1e-6
  fd = torch.zeros_like(inter)
  for idx in [(0, 0, 0), (0, 5, 1), (1, 31, 0), (1, 16, 1)]:
    e = torch.zeros_like(inter)
    e[idx] = eps
    up = O.discriminator_forward(d, inter + e, shifts, hp)[idx[0], 0]
    dn = O.discriminator_forward(d, inter - e, shifts, hp)[idx[0], 0]
    np.testing.assert_allclose(
        grad[idx].item(), ((up - dn) / (2 * eps)).item(), rtol=1e-5, atol=1e-9)
  exp = ((grad.reshape(2, -1).norm(dim=1) - 1)**2).mean()
  np.testing.assert_allclose(gp.item(), exp.item(), rtol=1e-12)


def test_signal_metrics_known_answer():
  real = torch.tensor([[[0.0, 1.0], [0.5, 0.5]]])
  fake = torch.tensor([[[1.0, 1.0], [0.0, 1.0]]])
  m = O.signal_metrics(real, fake, 0.0, 2.0, True)
  # denorm x2: real rows (0,2),(1,1); fake rows (2,2),(0,2)
  np.testing.assert_allclose(m['signals_metrics/min'].item(), (4 + 1) / 2)
  np.testing.assert_allclose(m['signals_metrics/max'].item(), (0 + 1) / 2)
  np.testing.assert_allclose(m['signals_metrics/mean'].item(), (1 + 0) / 2)
  np.testing.assert_allclose(m['signals_metrics/std'].item(), (1 + 1) / 2)


def test_dynamic_loss_scale_state_machine():
  """tf DynamicLossScale under LossScaleOptimizer (optimizer.py:10-12,23-34;
  SURVEY A.8): initial 2**15, doubled after `increment_period` consecutive
  finite updates, halved (floor 1) on a non-finite one, which is skipped."""
  ls = O.DynamicLossScale()
  assert ls.scale == 2.0**15 and ls.period == 2000
  ok, bad = [torch.ones(3)], [torch.tensor([1.0, float('nan')])]
  for _ in range(1999):
    assert ls.update(ok)
  assert ls.scale == 2.0**15 and ls.good_steps == 1999
  assert ls.update(ok) and ls.scale == 2.0**16 and ls.good_steps == 0
  assert not ls.update(bad) and ls.scale == 2.0**15 and ls.good_steps == 0
  assert not ls.update([torch.tensor([float('inf')])]) and ls.scale == 2.0**14
  ls.scale = 1.0
  assert not ls.update(bad) and ls.scale == 1.0
  # fp16 emulation: 11 significand bits, overflow to infinity past 65504
  x = torch.tensor([1.0 + 2.0**-11, 65520.0, 1e-8])
  r = O.f16_round(x)
  assert float(r[0]) == 1.0 and torch.isinf(r[1]) and float(r[2]) == 0.0
  # the scaler skips exactly the non-finite update of an oracle step
  hp = O.make_hparams(64, 6, 8, m=2)
  rng = np.random.RandomState(0)
  gan = O.OracleGAN(hp, O.init_generator(hp, rng), O.init_discriminator(hp, rng),
                    emulate_f16=True, loss_scaling=True)
  real = rng.uniform(0, 1, (2, 64, 6)).astype(np.float32)
  gan.train(real, O.draw_randomness(hp, 2, 0))
  assert gan.dis_steps == hp.n_critic and gan.gen_steps == 1
  assert gan.dis_scale.good_steps == hp.n_critic


def test_gram_operator_identities_of_the_penalty_first_layer():
  """DESIGN section 9's costed lead, pinned before anybody builds it: the input
  gradient g of the critic's first Conv1D enters the step only through ||g||^2, the
  tangent's first layer conv(c g) and layer 1's weight gradient -- all three are
  functions of delta_1 through G = W^T W (23 lags, Co x Co; truncated kernels in the
  first / last six rows of a sample).  float64, the oracle's Conv1D as ground truth
  (tools/probe/gram_layer1.py)."""
  import importlib.util
  import os
  path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                      'tools', 'probe', 'gram_layer1.py')
  spec = importlib.util.spec_from_file_location('gram_layer1', path)
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  assert mod.check(T=20, ci=5, co=4, seed=0) < 1.0
  mod.check(T=13, ci=3, co=6, seed=1)
