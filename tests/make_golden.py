"""Generate tests/golden/*.npz from the CPU oracle (run from the repo root:
``python tests/make_golden.py``).  The reference ships no golden vectors and
cannot run here (no TensorFlow), so these pin the ORACLE's own outputs: any
later edit of oracle/ that changes its arithmetic shows up as a diff, and the
GPU tests replay the same inputs through the HIP path.

Also pins the synthetic-input generator against the one piece of the
reference that IS importable here (numpy/scipy only):
/root/reference/dataset/dg/dichot_gauss.py -- see make_dg_golden().
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle as O  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')


def make_step_golden():
  hp = O.make_hparams(64, 6, 8, m=2)
  rng = np.random.RandomState(2024)
  gw = O.init_generator(hp, rng)
  dw = O.init_discriminator(hp, rng)
  for w in gw + dw:
    if w.ndim == 1:
      w += rng.randn(*w.shape).astype(np.float32) * 0.05
  B = 4
  real = rng.uniform(0, 1, (B, 64, 6)).astype(np.float32)
  rand = O.draw_randomness(hp, B, seed=5)
  r = rand['critic'][0]
  gt = [torch.tensor(w) for w in gw]
  dt = [torch.tensor(w) for w in dw]
  crit = O.d_step_grads(gt, dt, torch.tensor(real), torch.tensor(r['z']),
                        torch.tensor(r['alpha']), r['shifts_real'],
                        r['shifts_fake'], r['shifts_inter'], hp)
  gen = O.g_step_grads(gt, dt, torch.tensor(rand['gen']['z']),
                       rand['gen']['shifts'], hp)
  gan = O.OracleGAN(hp, gw, dw)
  out = gan.train(real, rand)
  d = dict(real=real, z=r['z'], alpha=r['alpha'],
           shifts_real=r['shifts_real'], shifts_fake=r['shifts_fake'],
           shifts_inter=r['shifts_inter'], gen_z=rand['gen']['z'],
           gen_shifts=rand['gen']['shifts'],
           fake=crit['fake'].numpy(), real_out=crit['real_out'].numpy(),
           fake_out=crit['fake_out'].numpy(), norm=crit['norm'].numpy(),
           gp=np.float32(crit['gp']), dis_loss=np.float32(crit['loss']),
           gen_loss=np.float32(gen['loss']),
           train_out=np.array(out[:3], np.float64),
           train_metrics=np.array([out[3][k] for k in sorted(out[3])]))
  for i, w in enumerate(gw):
    d['gw%02d' % i] = w
  for i, w in enumerate(dw):
    d['dw%02d' % i] = w
  for i, g in enumerate(crit['grads']):
    d['dgrad%02d' % i] = g.numpy()
  for i, g in enumerate(gen['grads']):
    d['ggrad%02d' % i] = g.numpy()
  for i, w in enumerate(gan.dis):
    d['dw_after%02d' % i] = w.numpy()
  for i, w in enumerate(gan.gen):
    d['gw_after%02d' % i] = w.numpy()
  np.savez_compressed(os.path.join(OUT, 'wgan_gp_step_tiny.npz'), **d)


def make_dg_golden():
  """Outputs of the REFERENCE's DichotGauss / DGOptimise (imported from
  /root/reference/dataset, numpy+scipy only) on seeded inputs."""
  ref = '/root/reference/dataset'
  if not os.path.isdir(ref):
    print('reference not present; keeping existing dg golden')
    return
  sys.path.insert(0, ref)
  from dg.dichot_gauss import DichotGauss
  from dg.optim_dichot_gauss import DGOptimise
  np.random.seed(1234)
  n, T = 6, 4000
  rates = np.array([0.02, 0.05, 0.1, 0.2, 0.4, 0.6])
  from scipy.stats import norm
  gamma = norm.ppf(rates)[None, :]
  rho = 0.05
  corr = (1 - rho) * np.eye(n) + rho * np.ones((n, n))
  dg = DichotGauss(n, mean=gamma, corr=corr, make_pd=True)
  spikes = dg.sample(repeats=T)  # (1, T, n)
  opt = DGOptimise(np.transpose(spikes, (1, 0, 2)).reshape(1, T, n))
  np.savez_compressed(
      os.path.join(OUT, 'dg_reference.npz'), rates=rates, gamma=gamma, rho=rho,
      spike_mean=spikes.mean(axis=(0, 1)),
      spike_cov=np.cov(spikes[0].T),
      gauss_mean=np.asarray(opt.gauss_mean))


if __name__ == '__main__':
  os.makedirs(OUT, exist_ok=True)
  make_step_golden()
  make_dg_golden()
  print('wrote', os.listdir(OUT))
