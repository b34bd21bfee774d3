"""Generate tests/golden/wgan_gp_cfg2.npz: the CPU oracle at the FULL layer
shapes of BASELINE.json configs[1] (L=2048, C=102, U=64, k=24, m=10, layer
norm) at the benchmark batch (128), for tests/test_hip_cfg2.py.  Run from the repo root:

  python tests/make_golden_cfg2.py

Weights and inputs are not stored (8.5 M parameters): both this script and the
test rebuild them from the same seeds through oracle.init_* (numpy
RandomState: stable across machines).  Stored are the ORACLE'S OUTPUTS -- f32
and with bf16 storage emulated -- for one critic update, one generator update,
one full train() and a 10-step loss trajectory: losses, penalty, per-sample
norms and critic outputs, per-tensor gradient norms, a few hundred sampled
gradient elements per tensor, a strided slice of the generated batch.

Like tests/make_golden.py this pins the oracle's own arithmetic (the reference
ships no vectors and TensorFlow is absent): parity with the reference stays
"unpinned"; what the fixture buys is a full-shape, tuned-tile check of the HIP
path that does not need the oracle's minutes of CPU time on the GPU box.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle as O  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden', 'wgan_gp_cfg2.npz')
SHAPE = dict(L=2048, C=102, U=64, k=24, m=10)
B = 128
SEED_W, SEED_X, SEED_R = 2025, 2026, 7
N_SAMPLE = 256  # sampled elements per gradient tensor
TRAJ_STEPS = 10


def build():
  """(hp, generator weights, discriminator weights, real batch): shared with
  the test."""
  hp = O.make_hparams(SHAPE['L'], SHAPE['C'], SHAPE['U'],
                      kernel_size=SHAPE['k'], m=SHAPE['m'])
  rng = np.random.RandomState(SEED_W)
  gw = O.init_generator(hp, rng)
  dw = O.init_discriminator(hp, rng)
  for w in gw + dw:  # biases / LN parameters away from their trivial values
    if w.ndim == 1:
      w += rng.randn(*w.shape).astype(np.float32) * 0.05
  xr = np.random.RandomState(SEED_X)
  # smooth positive traces in [0, 1] (closer to calcium signals than white
  # noise: neighbouring time steps correlate)
  base = xr.uniform(0, 1, (B, SHAPE['L'] // 16 + 1, SHAPE['C']))
  real = np.repeat(base, 16, axis=1)[:, :SHAPE['L']]
  real = (0.8 * real + 0.2 * xr.uniform(0, 1, real.shape)).astype(np.float32)
  return hp, gw, dw, real


def sample_index(i, n):
  """The sampled flat positions of gradient tensor i (n elements)."""
  r = np.random.RandomState(1000 + i)
  return np.sort(r.choice(n, size=min(N_SAMPLE, n), replace=False))


def _pack(prefix, d, grads, out):
  out[prefix + 'norms'] = np.array(
      [float(g.double().norm()) for g in grads], np.float64)
  for i, g in enumerate(grads):
    flat = g.reshape(-1).numpy()
    out[prefix + 'g%02d' % i] = flat[sample_index(i, flat.size)]


def main():
  hp, gw, dw, real = build()
  rand = O.draw_randomness(hp, B, seed=SEED_R)
  out = dict(batch=np.int64(B))
  for tag, emu in (('f32_', False), ('emu_', True)):
    q = O.bf16_round if emu else (lambda x: x)
    gt = [torch.tensor(w) for w in gw]
    dt = [torch.tensor(w) for w in dw]
    r = rand['critic'][0]
    kw = dict(q=q, wq=q) if emu else {}
    crit = O.d_step_grads(gt, dt, torch.tensor(real), torch.tensor(r['z']),
                          torch.tensor(r['alpha']), r['shifts_real'],
                          r['shifts_fake'], r['shifts_inter'], hp, **kw)
    gen = O.g_step_grads(gt, dt, torch.tensor(rand['gen']['z']),
                         rand['gen']['shifts'], hp, **kw)
    out[tag + 'dis_loss'] = np.float64(crit['loss'])
    out[tag + 'gp'] = np.float64(crit['gp'])
    out[tag + 'norm'] = crit['norm'].numpy()
    out[tag + 'real_out'] = crit['real_out'].numpy().reshape(-1)
    out[tag + 'fake_out'] = crit['fake_out'].numpy().reshape(-1)
    out[tag + "fake_slice"] = crit["fake"].numpy()[::8, ::64, ::6]
    out[tag + 'gen_loss'] = np.float64(gen['loss'])
    out[tag + 'gen_fake_out'] = gen['fake_out'].numpy().reshape(-1)
    _pack(tag + 'd_', crit, crit['grads'], out)
    _pack(tag + 'g_', gen, gen['grads'], out)
    # one full train(): 5 critic updates + 1 generator update
    gan = O.OracleGAN(hp, gw, dw, emulate_bf16=emu)
    o = gan.train(real, rand)
    out[tag + 'train_out'] = np.array(o[:3], np.float64)
    out[tag + 'train_metrics'] = np.array([o[3][k] for k in sorted(o[3])])
    out[tag + 'train_dmove'] = np.array(
        [float((a - torch.tensor(b)).double().norm())
         for a, b in zip(gan.dis, dw)])
    out[tag + 'train_gmove'] = np.array(
        [float((a - torch.tensor(b)).double().norm())
         for a, b in zip(gan.gen, gw)])
    print(tag, 'critic', float(crit['loss']), float(crit['gp']), 'gen',
          float(gen['loss']), 'train', o[:3], flush=True)
  # loss trajectory of the f32 oracle (training dynamics at full shapes)
  gan = O.OracleGAN(hp, gw, dw)
  traj = []
  for s in range(TRAJ_STEPS):
    o = gan.train(real, O.draw_randomness(hp, B, seed=100 + s))
    traj.append(o[:3])
    print('traj', s, o[:3], flush=True)
  out['traj'] = np.array(traj, np.float64)
  np.savez_compressed(OUT, **out)
  print('wrote', OUT, os.path.getsize(OUT), 'bytes')


if __name__ == '__main__':
  main()
