"""Worker of tests/test_parallel_gpu.py: one rank of a 2-rank data-parallel run
of the real HIP train() (collectives over gloo so that both ranks can share the
single GPU of the test box; the production backend is 'nccl' = RCCL and uses
the same code).  Writes one JSON record per rank into $DP_WORKER_OUT/rank<r>.json."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

import oracle as O
from calciumgan_amd import parallel


def main():
  # DP_BACKEND=nccl: RCCL, one GPU per rank (test_parallel_gpu.py launches it
  # only when the box has one per rank)
  backend = os.environ.get('DP_BACKEND', 'gloo')
  parallel.init_process_group(backend)
  rank, world = parallel.rank(), parallel.world_size()
  from calciumgan_amd.gan.algorithms import get_algorithm
  from calciumgan_amd.gan.models import get_models
  hp = O.make_hparams(256, 16, 32, kernel_size=24, m=2, layer_norm=True)
  hp.verbose = 0
  gen, dis = get_models(hp, None)          # same seed -> same initial weights
  gan = get_algorithm(hp, gen, dis, None)
  assert gan._sync.world == world
  rng = np.random.RandomState(7)
  full = rng.uniform(0, 1, (8 * world, 256, 16)).astype(np.float32)
  mine = torch.tensor(full[rank::world]).to(gan.device)
  losses = []
  for _ in range(5):                        # 2 eager calls, then graph replays
    gl, dl, gp, metrics = gan.train(mine)
    losses.append([float(gl), float(dl), float(gp)])
  torch.cuda.synchronize()
  steps = int(os.environ.get('DP_TIMED_STEPS', '0'))
  ms_per_step = None
  if steps:                                 # tools/dp_overlap.sh
    import time
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
      gan.train(mine)
    torch.cuda.synchronize()
    dist.barrier()
    ms_per_step = (time.perf_counter() - t0) / steps * 1e3
  # collectives of the checks below: device tensors over RCCL, host over gloo
  cdev = gan.device if backend == 'nccl' else torch.device('cpu')
  # logged scalars are means over the ranks: identical everywhere
  logged = [torch.empty(3, dtype=torch.float64, device=cdev)
            for _ in range(world)]
  dist.all_gather(logged, torch.tensor(losses[-1], dtype=torch.float64,
                                       device=cdev))
  # every rank launches rank 0's tile choices
  from calciumgan_amd import nets
  import zlib
  tiles = zlib.crc32(repr(sorted(nets._TILE_CACHE.items())).encode())
  tile_ids = [torch.empty(1, dtype=torch.int64, device=cdev)
              for _ in range(world)]
  dist.all_gather(tile_ids, torch.tensor([tiles], dtype=torch.int64,
                                         device=cdev))
  flat = torch.cat([gan.generator.net.params.data,
                    gan.discriminator.net.params.data]).to(cdev)
  gathered = [torch.empty_like(flat) for _ in range(world)]
  dist.all_gather(gathered, flat)
  same = all(torch.equal(gathered[0], g) for g in gathered)
  z_other = [torch.empty(4, device=cdev) for _ in range(world)]
  dist.all_gather(z_other, gan.get_noise(1)[0, :4].to(cdev))
  rec = json.dumps(dict(
      rank=rank, world=world, backend=backend, weights_identical=bool(same),
      logged_identical=bool(all(torch.equal(logged[0], l) for l in logged)),
      tiles_identical=bool(all(int(t) == int(tile_ids[0]) for t in tile_ids)),
      ms_per_step=ms_per_step,
      finite=bool(np.isfinite(np.array(losses)).all()),
      graphed=bool(gan._state[mine.shape[0]].get('graph') is not None),
      segments=len(gan._state[mine.shape[0]]['graph']['graphs']),
      noise_differs=bool(not torch.equal(z_other[0], z_other[1])),
      moved=float((flat - gathered[0]).abs().max()), losses=losses[-1]))
  # one file per rank: the ranks' stdout streams interleave (gloo's own
  # connection messages land in the middle of lines)
  out_dir = os.environ.get('DP_WORKER_OUT')
  if out_dir:
    with open(os.path.join(out_dir, 'rank{}.json'.format(rank)), 'w') as f:
      f.write(rec)
  print(rec, flush=True)
  dist.barrier()
  dist.destroy_process_group()


if __name__ == '__main__':
  main()
