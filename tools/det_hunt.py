#!/usr/bin/env python
"""Which step / tensor first differs between two processes (development tool of
tests/test_determinism.py).  Prints one line per train() call: short hashes of
the step's outputs, of every gradient buffer and of the weights."""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import oracle as O


def h(t):
  return hashlib.sha256(t.detach().cpu().numpy().tobytes()).hexdigest()[:8]


def main():
  steps = int(sys.argv[1])
  L, C, U, B = (int(v) for v in sys.argv[2:6])
  from calciumgan_amd.gan.algorithms import get_algorithm
  from calciumgan_amd.gan.models import get_models
  hp = O.make_hparams(L, C, U, m=2)
  hp.verbose = 0
  gen, dis = get_models(hp, None)
  gan = get_algorithm(hp, gen, dis, None)
  rng = np.random.RandomState(7)
  real = torch.from_numpy(rng.uniform(0, 1, (B, L, C)).astype(np.float32)).cuda()
  for s in range(steps):
    out = gan.train(real)
    torch.cuda.synchronize()
    vals = torch.stack([out[0], out[1], out[2]] + list(out[3].values()))
    dg = dis.net.params.grad
    gg = gen.net.params.grad
    parts = ['out ' + h(vals)]
    parts.append('Dgrad ' + ' '.join(h(v) for v in dis.net.params.grad_views))
    parts.append('Ggrad ' + ' '.join(h(v) for v in gen.net.params.grad_views))
    parts.append('Dw ' + h(dis.net.params.data) + ' Gw ' + h(gen.net.params.data))
    print('step %d | %s' % (s, ' | '.join(parts)), flush=True)


if __name__ == '__main__':
  main()
