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


EQUIV = dict(L=256, C=16, U=16, B=8, seed=11)   # DP_MODE=equiv, shared with the test


def equiv_inputs(hp, world):
  """Global batch + draws of the DP == single-rank-on-the-global-batch check:
  (real (B, L, C), critic draws, generator draws)."""
  rng = np.random.RandomState(7)
  real = rng.uniform(0, 1, (EQUIV['B'], EQUIV['L'], EQUIV['C'])).astype(np.float32)
  r = O.draw_randomness(hp, EQUIV['B'], seed=EQUIV['seed'])
  return real, r['critic'][0], r['gen']


def shard_draws(r, rank, world):
  out = {}
  for k, v in r.items():
    v = np.asarray(v)
    # per-sample draws (z, alpha) are sharded like the batch; the phase shifts
    # are one draw per layer for the whole global batch
    out[k] = v if k.startswith('shifts') else v[rank::world]
  return out


def equiv(backend):
  """HIP data parallel == HIP single rank on the global batch: every rank
  computes the critic / generator gradients of its shard with ITS rows of the
  injected draws; after the all-reduce and the 1/world of grad_scale they are
  the global-batch gradients (test_parallel_gpu.py computes those itself)."""
  rank, world = parallel.rank(), parallel.world_size()
  from calciumgan_amd.gan.algorithms import get_algorithm
  from calciumgan_amd.gan.models import get_models
  hp = O.make_hparams(EQUIV['L'], EQUIV['C'], EQUIV['U'], kernel_size=24, m=2,
                      layer_norm=True)
  hp.verbose = 0
  gen, dis = get_models(hp, None)
  gan = get_algorithm(hp, gen, dis, None)
  real, rc, rg = equiv_inputs(hp, world)
  mine = torch.tensor(real[rank::world]).to(gan.device)
  gan._critic_compute(mine, shard_draws(rc, rank, world), slot=0)
  gan._sync.all_reduce(dis.net.params.grad)
  d_grad = (dis.net.params.grad * gan._sync.grad_scale).cpu().numpy()
  st = gan._get_state(mine.shape[0])
  local = torch.stack([st['gp'][0], st['loss'][0, 0]]).double().cpu()
  dist.all_reduce(local)
  gan._gen_compute(mine, shard_draws(rg, rank, world))
  gan._sync.all_reduce(gen.net.params.grad)
  g_grad = (gen.net.params.grad * gan._sync.grad_scale).cpu().numpy()
  out_dir = os.environ['DP_WORKER_OUT']
  np.savez(os.path.join(out_dir, 'equiv_rank{}.npz'.format(rank)),
           d_grad=d_grad, g_grad=g_grad, gp_loss=(local / world).numpy())
  dist.barrier()
  dist.destroy_process_group()


def overflow(backend):
  """mixed_float16 under data parallelism: a non-finite gradient on ONE rank
  must skip the update on EVERY rank -- the finite check runs on the reduced
  gradients (optimizer.py:23-34 after the all-reduce), never per rank."""
  rank, world = parallel.rank(), parallel.world_size()
  from calciumgan_amd.gan.algorithms import get_algorithm
  from calciumgan_amd.gan.models import get_models
  hp = O.make_hparams(64, 6, 8, kernel_size=24, m=2, layer_norm=True)
  hp.mixed_precision = True
  hp.verbose = 0
  gen, dis = get_models(hp, None)
  gan = get_algorithm(hp, gen, dis, None)
  rng = np.random.RandomState(3)
  real = torch.tensor(rng.uniform(0, 1, (4, 64, 6)).astype(np.float32)).to(
      gan.device)
  rec = {}
  for tag, poison in (('clean', False), ('one_rank_inf', True)):
    w0 = dis.net.params.data.clone()
    scale0 = float(gan.dis_optimizer.loss_scale_state[0])
    it0 = gan.dis_optimizer.iterations
    gan._critic_compute(real, None, slot=0)
    if poison and rank == world - 1:
      dis.net.params.grad[5] = float('inf')
    gan._sync.all_reduce(dis.net.params.grad)
    gan._critic_apply()
    torch.cuda.synchronize()
    rec[tag] = dict(
        moved=float((dis.net.params.data - w0).abs().max()),
        scale_before=scale0,
        scale_after=float(gan.dis_optimizer.loss_scale_state[0]),
        applied=gan.dis_optimizer.iterations - it0,
        finite=bool(torch.isfinite(dis.net.params.data).all()))
  with open(os.path.join(os.environ['DP_WORKER_OUT'],
                         'overflow_rank{}.json'.format(rank)), 'w') as f:
    json.dump(rec, f)
  dist.barrier()
  dist.destroy_process_group()


def main():
  # DP_BACKEND=nccl: RCCL, one GPU per rank (test_parallel_gpu.py launches it
  # only when the box has one per rank)
  backend = os.environ.get('DP_BACKEND', 'gloo')
  parallel.init_process_group(backend)
  mode = os.environ.get('DP_MODE', 'train')
  if mode == 'equiv':
    return equiv(backend)
  if mode == 'overflow':
    return overflow(backend)
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
