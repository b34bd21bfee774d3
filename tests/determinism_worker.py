"""Worker of tests/test_determinism.py: a fresh process trains the HIP path for
N train() calls (two eager, the rest hipGraph replays) from fixed weights on a
fixed batch with the process's own seeded draws and prints one line: the
SHA-256 of every weight tensor, of the Adam moments and of the returned
scalars of every step.  Tiles are the static choice (CALCIUMGAN_AUTOTUNE=0 is
set by the test), reductions the ordered ones (the default)."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import oracle as O


def main():
  steps = int(sys.argv[1])
  L, C, U, B = (int(v) for v in sys.argv[2:6])
  from calciumgan_amd.gan.algorithms import get_algorithm
  from calciumgan_amd.gan.models import get_models
  np.random.seed(1234)
  torch.manual_seed(1234)
  hp = O.make_hparams(L, C, U, m=2)
  hp.verbose = 0
  gen, dis = get_models(hp, None)
  gan = get_algorithm(hp, gen, dis, None)
  rng = np.random.RandomState(7)
  real = torch.from_numpy(rng.uniform(0, 1, (B, L, C)).astype(np.float32)).cuda()
  h = hashlib.sha256()
  first = None
  for _ in range(steps):
    out = gan.train(real)
    vals = torch.stack([out[0], out[1], out[2]] + list(out[3].values()))
    h.update(vals.cpu().numpy().tobytes())
    if first is None:
      first = [float(v) for v in vals.cpu()]
  torch.cuda.synchronize()
  hw = hashlib.sha256()
  for w in gen.get_weights() + dis.get_weights():
    hw.update(np.ascontiguousarray(w).tobytes())
  for net in (gen.net, dis.net):
    hw.update(net.params.m.cpu().numpy().tobytes())
    hw.update(net.params.v.cpu().numpy().tobytes())
  print(json.dumps({'weights': hw.hexdigest(), 'outputs': h.hexdigest(),
                    'last': [float(v) for v in vals.cpu()], 'first': first,
                    'graph': gan._get_state(B).get('graph') is not None}))


if __name__ == '__main__':
  main()
