"""The whole WGAN-GP step at the FULL layer shapes and batch of the benchmark
(BASELINE.json configs[1]: L=2048, C=102, U=64, k=24, m=10, batch 128) against
the committed oracle fixture tests/golden/wgan_gp_cfg2.npz (written by
tests/make_golden_cfg2.py: f32 oracle and the oracle with bf16 storage
emulated, same seeds for weights / inputs / draws).

Launches use the static tile choice, which since round 5 is also what bench.py
and main.py run by default (the tuner is opt-in, CALCIUMGAN_AUTOTUNE=1): same
batch, same launch geometries, so the kernels bench.py times are the kernels
checked here.

Tolerances: forward values / losses 1e-2 against the emulating
oracle, 3e-2 against f32; gradients per tensor by norm and by sampled elements
(bars below, set from the measured bf16 noise floor of the emulation itself).
"""
import importlib.util
import os

import numpy as np
import pytest
import torch

import oracle as O

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLDEN = os.path.join(HERE, 'golden', 'wgan_gp_cfg2.npz')


def _maker():
  spec = importlib.util.spec_from_file_location(
      'make_golden_cfg2', os.path.join(HERE, 'make_golden_cfg2.py'))
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  return mod


@pytest.fixture(scope='module')
def cfg2():
  """(maker module, golden arrays, hparams, gen, dis, gan, real): models hold
  the fixture's initial weights; tiles are the static (default) choice."""
  from calciumgan_amd import nets
  from calciumgan_amd.gan.algorithms import get_algorithm
  from calciumgan_amd.gan.models import get_models
  assert not nets._AUTOTUNE or os.environ.get('CALCIUMGAN_AUTOTUNE') == '1'
  M = _maker()
  gold = np.load(GOLDEN)
  hp, gw, dw, real = M.build()
  assert int(gold['batch']) == real.shape[0] == 128
  hp.verbose = 0
  gen, dis = get_models(hp, None)
  gan = get_algorithm(hp, gen, dis, None)
  return M, gold, hp, gen, dis, gan, real, gw, dw


def _reset(gen, dis, gan, gw, dw):
  gen.set_weights(gw)
  dis.set_weights(dw)
  for net in (gen.net, dis.net):
    net.params.m.zero_()
    net.params.v.zero_()
    net.repack()
  gan.gen_optimizer.iterations = 0
  gan.dis_optimizer.iterations = 0


def _check_grad_samples(M, gold, prefix, views, what):
  """Per tensor: the gradient norm against the f32 oracle, and the sampled
  elements against both oracles.  bf16 storage moves a critic / generator
  gradient tensor by 5-15 % (relative L2) from f32 -- the fixture's own
  emu-vs-f32 distance, measured on the same samples, sets the bar: HIP is no
  further from f32 than 2x that (+1e-2) and within 1.5x of it (+1e-2) of the
  emulation."""
  n_f32 = gold['f32_' + prefix + 'norms']
  bad = []
  for i, g in enumerate(views):
    g = g.detach().cpu().numpy().reshape(-1).astype(np.float64)
    idx = M.sample_index(i, g.size)
    ref = gold['f32_%sg%02d' % (prefix, i)].astype(np.float64)
    emu = gold['emu_%sg%02d' % (prefix, i)].astype(np.float64)
    if n_f32[i] < 1e-12:
      assert np.abs(g).max() < 1e-6, (what, i)
      continue
    # sampled elements carry the tensor's per-element noise: distances relative
    # to the samples' own norm
    den = np.linalg.norm(ref) + 1e-30
    e_hf = np.linalg.norm(g[idx] - ref) / den
    e_ef = np.linalg.norm(emu - ref) / den
    e_he = np.linalg.norm(g[idx] - emu) / den
    nr = np.linalg.norm(g) / n_f32[i]
    if (e_hf > 2.0 * e_ef + 1e-2 or e_he > 1.5 * e_ef + 1e-2 or
        abs(nr - 1) > max(5e-2, 2.0 * e_ef)):
      bad.append((i, round(e_hf, 4), round(e_ef, 4), round(e_he, 4),
                  round(nr, 4)))
  assert not bad, '%s: (tensor, hip-f32, emu-f32, hip-emu, norm ratio) %s' % (
      what, bad)
  tot = np.sqrt(sum(float(np.linalg.norm(
      g.detach().cpu().numpy().astype(np.float64)))**2 for g in views))
  assert abs(tot / np.linalg.norm(n_f32) - 1) < 2e-2, what


def test_critic_update_matches_oracle_at_benchmark_shapes(cfg2):
  M, gold, hp, gen, dis, gan, real, gw, dw = cfg2
  _reset(gen, dis, gan, gw, dw)
  B = real.shape[0]
  r = O.draw_randomness(hp, B, seed=M.SEED_R)['critic'][0]
  loss, gp = gan._train_discriminator(real, r, slot=0)
  torch.cuda.synchronize()
  st = gan._get_state(B)
  d_out = st['dws'].d_out.cpu().numpy()
  fake = st['gws'].fake  # not kept by the forward-only pass
  for tag, tol in (('emu_', 1e-2), ('f32_', 3e-2)):
    np.testing.assert_allclose(d_out[:B], gold[tag + 'real_out'], rtol=tol,
                               atol=tol * 0.1)
    np.testing.assert_allclose(d_out[B:2 * B], gold[tag + 'fake_out'], rtol=tol,
                               atol=tol * 0.1)
    np.testing.assert_allclose(st['norm_out'].cpu().numpy(), gold[tag + 'norm'],
                               rtol=tol)
    np.testing.assert_allclose(float(gp), float(gold[tag + 'gp']), rtol=tol)
    np.testing.assert_allclose(float(loss), float(gold[tag + 'dis_loss']),
                               rtol=tol)
  _check_grad_samples(M, gold, 'd_', dis.net.params.grad_views, 'critic')


def test_generator_update_matches_oracle_at_benchmark_shapes(cfg2):
  M, gold, hp, gen, dis, gan, real, gw, dw = cfg2
  _reset(gen, dis, gan, gw, dw)
  B = real.shape[0]
  r = O.draw_randomness(hp, B, seed=M.SEED_R)['gen']
  loss, metrics = gan._train_generator(real, r)
  torch.cuda.synchronize()
  st = gan._get_state(B)
  np.testing.assert_allclose(float(loss), float(gold['emu_gen_loss']),
                             rtol=1e-2, atol=1e-3)
  np.testing.assert_allclose(float(loss), float(gold['f32_gen_loss']),
                             rtol=3e-2, atol=3e-3)
  np.testing.assert_allclose(st['dws'].d_out.cpu().numpy()[:B],
                             gold['emu_gen_fake_out'], rtol=1e-2, atol=1e-3)
  _check_grad_samples(M, gold, 'g_', gen.net.params.grad_views, 'generator')


def test_generated_batch_matches_oracle_at_benchmark_shapes(cfg2):
  """G(z) of the first critic update: the fixture keeps every 8th sample,
  64th time step and 6th channel."""
  M, gold, hp, gen, dis, gan, real, gw, dw = cfg2
  _reset(gen, dis, gan, gw, dw)
  B = real.shape[0]
  r = O.draw_randomness(hp, B, seed=M.SEED_R)['critic'][0]
  fake = gan._critic_generate(real, r)
  torch.cuda.synchronize()
  got = fake[:, :, :hp.num_channels].cpu().numpy()[::8, ::64, ::6]
  np.testing.assert_allclose(got, gold['emu_fake_slice'], atol=4e-3)
  np.testing.assert_allclose(got, gold['f32_fake_slice'], atol=2e-2)


def test_train_step_matches_oracle_at_benchmark_shapes(cfg2):
  """One whole train() (5 critic updates + 1 generator update, Keras Adam):
  losses, signal metrics and how far every weight tensor moved."""
  M, gold, hp, gen, dis, gan, real, gw, dw = cfg2
  _reset(gen, dis, gan, gw, dw)
  B = real.shape[0]
  out = gan.train(real, O.draw_randomness(hp, B, seed=M.SEED_R))
  torch.cuda.synchronize()
  got = [float(out[0]), float(out[1]), float(out[2])]
  np.testing.assert_allclose(got, gold['emu_train_out'], rtol=3e-2, atol=3e-3)
  np.testing.assert_allclose(got, gold['f32_train_out'], rtol=5e-2, atol=5e-3)
  np.testing.assert_allclose([float(out[3][k]) for k in sorted(out[3])],
                             gold['emu_train_metrics'], rtol=2e-2)
  # Adam's first steps move every element by ~lr * sign(g): the per-tensor
  # movement norm is the robust summary (a wrong gradient SIGN pattern, missing
  # update or wrong step size shows here)
  for now, init, ref in ((dis.get_weights(), dw, gold['emu_train_dmove']),
                         (gen.get_weights(), gw, gold['emu_train_gmove'])):
    mv = np.array([np.linalg.norm(a.astype(np.float64) - b) for a, b in
                   zip(now, init)])
    np.testing.assert_allclose(mv, ref, rtol=0.1, atol=1e-7)


def test_training_dynamics_follow_oracle_at_benchmark_shapes(cfg2):
  """Ten train() steps on the same batch with seeded draws: the loss
  trajectory of the f32 oracle at these shapes grows by an order of magnitude
  within ten steps (generator loss 0 -> tens, critic loss -> -20...-55; the
  reference graph's own dynamics on a single repeated batch).  A bf16 path
  cannot track a chaotic trajectory element-wise; it must stay in the same
  regime step by step."""
  M, gold, hp, gen, dis, gan, real, gw, dw = cfg2
  _reset(gen, dis, gan, gw, dw)
  B = real.shape[0]
  traj = gold['traj']
  got = []
  for s in range(len(traj)):
    o = gan.train(real, O.draw_randomness(hp, B, seed=100 + s))
    got.append([float(o[0]), float(o[1]), float(o[2])])
  torch.cuda.synchronize()
  got = np.array(got)
  assert np.isfinite(got).all()
  # the first two steps are still deterministic enough to compare closely
  np.testing.assert_allclose(got[:2], traj[:2], rtol=0.1, atol=0.05)
  # later: same sign and scale of the critic loss, penalty of the same order
  late = slice(4, None)
  assert (got[late, 1] < 0).all() and (traj[late, 1] < 0).all()
  assert 0.3 < np.abs(got[late, 1]).mean() / np.abs(traj[late, 1]).mean() < 3.0
  assert 0.2 < got[late, 2].mean() / traj[late, 2].mean() < 5.0
  assert np.abs(got[:, 0]).max() < 10 * np.abs(traj[:, 0]).max() + 10
