"""Does forming the critic's first layer on x^ from its outputs on real / fake BIAS the
penalty norm?  CPU only: the oracle's bf16 emulation in both forms (conv: x^ rounded
and convolved; mix: oracle.layer1_mix_pre) against the f32 oracle, signed relative
error of ||grad_x^ D|| per sample over 40 draws x 4 samples at L 1024, 102 neurons,
num_units 64, critic kernels scaled by 1.6 (norms of order 1, as in a trained
critic).  python3 tools/probe/penalty_norm_bias.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import oracle as O
from oracle import calciumgan_oracle as OM
torch.set_num_threads(8)
L, C, U, B = 1024, 102, 64, 4
hp = O.make_hparams(L, C, U, kernel_size=24, m=10, layer_norm=True)
rng = np.random.RandomState(0)
gw = [torch.tensor(w) for w in O.init_generator(hp, rng)]
dw = [torch.tensor(w) * (1.6 if w.ndim == 3 else 1.0) + (torch.tensor(rng.randn(*w.shape).astype(np.float32)) * 0.05 if w.ndim == 1 else 0) for w in O.init_discriminator(hp, rng)]
rows = []
t0 = time.time()
for draw in range(40):
  r = O.draw_randomness(hp, B, seed=100 + draw)['critic'][0]
  real = torch.tensor(np.random.RandomState(draw).uniform(0, 1, (B, L, C)).astype(np.float32))
  with torch.no_grad():
    fake = O.generator_forward(gw, torch.tensor(r['z']), hp)
  alpha = torch.tensor(r['alpha'])
  out = {}
  for name, q, mix in (('f32', None, False), ('conv', O.bf16_round, False), ('mix', O.bf16_round, True)):
    OM.EMULATE_LAYER1_MIX = mix
    kw = {} if q is None else dict(q=q, wq=q)
    gp, norm, grad = O.gradient_penalty([w.clone().requires_grad_(True) for w in dw], real, fake, alpha, r['shifts_inter'], hp, create_graph=False, **kw)
    out[name] = norm.detach().numpy().astype(np.float64)
  for b in range(B):
    rows.append(((out['conv'][b] - out['f32'][b]) / out['f32'][b], (out['mix'][b] - out['f32'][b]) / out['f32'][b]))
  pass
rows = np.array(rows)
for i, name in enumerate(('conv', 'mix')):
  v = rows[:, i]
  print('%s: signed relative error of ||g|| vs f32: mean %+.5f +- %.5f (s.e.), sd %.5f, n = %d' % (name, v.mean(), v.std(ddof=1) / np.sqrt(len(v)), v.std(ddof=1), len(v)))
