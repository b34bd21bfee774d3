"""How close the critic step's penalty norms / losses sit to the oracles over several
draws of the injected randomness (tests/test_hip_step.py::test_critic_step_matches_oracle
runs ONE draw, seed 7): per configuration and seed, the worst relative distance of
the per-sample norms to the bf16-emulating oracle and to the f32 oracle.
  CALCIUMGAN_L1_LINEAR=0|1 python3 tools/probe/critic_noise.py [seeds [config,config...]]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import conftest  # noqa: F401,E402
import oracle as O  # noqa: E402
import test_hip_step as T  # noqa: E402

seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
names = sys.argv[2].split(',') if len(sys.argv) > 2 else list(T.CONFIGS)
from oracle import calciumgan_oracle as OM  # noqa: E402
from calciumgan_amd import nets  # noqa: E402

# both sides in the same form: the HIP plan forced to mix at any size, and the
# emulating oracle following it -- or both convolving x^
OM.EMULATE_LAYER1_MIX = os.environ.get('CALCIUMGAN_L1_LINEAR', '1') != '0'
nets._L1_LINEAR_MIN_ROWS = 0
print('layer 1 of x^ from the other two segments (product and emulation):',
      OM.EMULATE_LAYER1_MIX)
for name in names:
  row = []
  for seed in range(1, seeds + 1):
    hp, gen, dis, gan, real, B = T._build(name)
    r = O.draw_randomness(hp, B, seed=seed)['critic'][0]
    emu = T._oracle_critic(hp, gen, dis, real, r, O.bf16_round)
    f32 = T._oracle_critic(hp, gen, dis, real, r, lambda x: x)
    loss, gp = gan._train_discriminator(real, r, slot=0)
    torch.cuda.synchronize()
    st = gan._get_state(B)
    n = st['norm_out'].cpu().numpy()
    e = lambda res: float(np.max(np.abs(n - res['norm'].numpy()) / res['norm'].numpy()))
    ee = float(np.max(np.abs(emu['norm'].numpy() - f32['norm'].numpy()) / f32['norm'].numpy()))
    row.append((e(emu), e(f32), ee))
  print('%-10s hip-emu %s | hip-f32 %s | emu-f32 %s' % (
      name, ' '.join('%.4f' % t[0] for t in row), ' '.join('%.4f' % t[1] for t in row),
      ' '.join('%.4f' % t[2] for t in row)), flush=True)
