import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from calciumgan_amd.gan.algorithms import get_algorithm
from calciumgan_amd.gan.models import get_models
hp = bench.make_hparams(8192, 512, 64, 10, mixed_precision=True)
gen, dis = get_models(hp, None)
gan = get_algorithm(hp, gen, dis, None)
B = 256
gws = gen.net.workspace(5 * B, forward_only=True)
print('streaming_out', gen.net.streaming_out, 'Cp', gen.net.Cp, 'L', gen.net.L,
      'can_interp', gws.can_interp(5), 'fuse', gan._can_fuse_interp(B, 5),
      'cinp', dis.net.layers[0].cinp)
