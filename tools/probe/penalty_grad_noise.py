"""Companion of penalty_norm_bias.py: the PENALTY TERM's weight gradients (the hand-
derived second backward's target) in both bf16 emulations -- x^ convolved, or layer
1 mixed from the real / fake outputs -- against the f32 oracle: relative L2 error
and norm ratio over 12 draws of 4 samples (L 1024, 102 neurons, num_units 64,
critic kernels x 1.6).  CPU only.  python3 tools/probe/penalty_grad_noise.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import oracle as O
from oracle import calciumgan_oracle as OM
torch.set_num_threads(8)
L, C, U, B = 1024, 102, 64, 4
hp = O.make_hparams(L, C, U, kernel_size=24, m=10, layer_norm=True)
rng = np.random.RandomState(0)
gw = [torch.tensor(w) for w in O.init_generator(hp, rng)]
dw0 = [torch.tensor(w) * (1.6 if w.ndim == 3 else 1.0) + (torch.tensor(rng.randn(*w.shape).astype(np.float32)) * 0.05 if w.ndim == 1 else 0) for w in O.init_discriminator(hp, rng)]
res = {'conv': [], 'mix': []}
cosd = {'conv': [], 'mix': []}
for draw in range(12):
  r = O.draw_randomness(hp, B, seed=100 + draw)['critic'][0]
  real = torch.tensor(np.random.RandomState(draw).uniform(0, 1, (B, L, C)).astype(np.float32))
  with torch.no_grad():
    fake = O.generator_forward(gw, torch.tensor(r['z']), hp)
  alpha = torch.tensor(r['alpha'])
  g = {}
  for name, q, mix in (('f32', None, False), ('conv', O.bf16_round, False), ('mix', O.bf16_round, True)):
    OM.EMULATE_LAYER1_MIX = mix
    dw = [w.clone().requires_grad_(True) for w in dw0]
    kw = {} if q is None else dict(q=q, wq=q)
    gp, norm, grad = O.gradient_penalty(dw, real, fake, alpha, r['shifts_inter'], hp, **kw)
    gr = torch.autograd.grad(10.0 * gp, dw, allow_unused=True)
    g[name] = np.concatenate([(x if x is not None else torch.zeros_like(w)).numpy().reshape(-1) for x, w in zip(gr, dw)]).astype(np.float64)
  for name in ('conv', 'mix'):
    res[name].append(np.linalg.norm(g[name] - g['f32']) / np.linalg.norm(g['f32']))
    cosd[name].append(np.linalg.norm(g[name]) / np.linalg.norm(g['f32']) - 1)
for name in ('conv', 'mix'):
  v = np.array(res[name]); w = np.array(cosd[name])
  print('%s: penalty-term weight gradient vs f32, relative L2 error mean %.4f sd %.4f (n = %d); norm ratio - 1: %+.4f +- %.4f' % (name, v.mean(), v.std(ddof=1), len(v), w.mean(), w.std(ddof=1) / np.sqrt(len(w))))
