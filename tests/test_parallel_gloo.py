"""Data-parallel path on CPU: 2 processes, gloo backend (the GPU path uses the
same code with backend 'nccl' = RCCL).  Compute in these tests is the oracle;
what is under test is calciumgan_amd.parallel (gradient all-reduce + scale,
random streams) and the sharding rule of SURVEY 8(e)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle as O
from calciumgan_amd import parallel


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  p = s.getsockname()[1]
  s.close()
  return p


def _worker(rank, world, port, fn, ret):
  os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                    WORLD_SIZE=str(world), RANK=str(rank),
                    LOCAL_RANK=str(rank))
  torch.set_num_threads(2)
  parallel.init_process_group('gloo')
  try:
    ret[rank] = fn(rank, world)
  finally:
    dist.barrier()
    dist.destroy_process_group()


def _run(fn, world=2):
  ctx = mp.get_context('spawn')
  ret = ctx.Manager().dict()
  port = _free_port()
  procs = [ctx.Process(target=_worker, args=(r, world, port, fn, ret))
           for r in range(world)]
  for p in procs:
    p.start()
  for p in procs:
    p.join(300)
    assert p.exitcode == 0
  return dict(ret)


def _allreduce_case(rank, world):
  sync = parallel.GradSync()
  g = torch.full((1000,), float(rank + 1))
  sync.all_reduce(g)
  mean = g * sync.grad_scale
  s = torch.tensor([float(rank), 2.0])
  sync.mean_scalars(s)
  return mean[:3].tolist(), s.tolist(), sync.world


def test_grad_sync_averages_over_ranks():
  out = _run(_allreduce_case)
  for r in (0, 1):
    mean, s, world = out[r]
    assert world == 2
    np.testing.assert_allclose(mean, [1.5, 1.5, 1.5])
    np.testing.assert_allclose(s, [0.5, 2.0])


def _streams_case(rank, world):
  st = parallel.RandomStreams(1234, torch.device('cpu'), m=10)
  sh = [st.shifts(3).tolist() for _ in range(3)]
  z = st.noise(4, 8)
  a = st.alpha(4)
  return sh, z.tolist(), a.tolist()


def test_random_streams_shared_shifts_private_noise():
  out = _run(_streams_case)
  assert out[0][0] == out[1][0]  # identical phase shifts on every rank
  assert out[0][1] != out[1][1] and out[0][2] != out[1][2]
  sh = np.array(out[0][0])
  assert sh.shape == (3, 4, 3) and sh.min() >= -10 and sh.max() <= 10


HP = dict(sequence_length=64, num_channels=4, num_units=8, m=2)


def _dp_case(rank, world):
  hp = O.make_hparams(**HP)
  rng = np.random.RandomState(0)
  gw = [torch.tensor(w) for w in O.init_generator(hp, rng)]
  dw = [torch.tensor(w) for w in O.init_discriminator(hp, rng)]
  data = np.random.RandomState(1).uniform(0, 1, (4, 64, 4)).astype(np.float32)
  r = O.draw_randomness(hp, 4, seed=3)['critic'][0]
  lo, hi = rank * 2, rank * 2 + 2  # shard by sample
  res = O.d_step_grads(gw, dw, torch.tensor(data[lo:hi]),
                       torch.tensor(r['z'][lo:hi]),
                       torch.tensor(r['alpha'][lo:hi]), r['shifts_real'],
                       r['shifts_fake'], r['shifts_inter'], hp)
  sync = parallel.GradSync()
  flat = torch.cat([g.reshape(-1) for g in res['grads']])
  sync.all_reduce(flat)
  flat *= sync.grad_scale
  return flat.numpy()


def test_sharded_critic_gradients_equal_global_batch():
  """Local-mean losses + all-reduce(avg) of gradients == gradient of the
  global-batch loss (the DP rule of SURVEY 8(e)), incl. the penalty term."""
  out = _run(_dp_case)
  hp = O.make_hparams(**HP)
  rng = np.random.RandomState(0)
  gw = [torch.tensor(w) for w in O.init_generator(hp, rng)]
  dw = [torch.tensor(w) for w in O.init_discriminator(hp, rng)]
  data = np.random.RandomState(1).uniform(0, 1, (4, 64, 4)).astype(np.float32)
  r = O.draw_randomness(hp, 4, seed=3)['critic'][0]
  res = O.d_step_grads(gw, dw, torch.tensor(data), torch.tensor(r['z']),
                       torch.tensor(r['alpha']), r['shifts_real'],
                       r['shifts_fake'], r['shifts_inter'], hp)
  ref = torch.cat([g.reshape(-1) for g in res['grads']]).numpy()
  np.testing.assert_allclose(out[0], out[1], rtol=0, atol=0)
  np.testing.assert_allclose(out[0], ref, rtol=2e-4, atol=1e-7)


def _world8_case(rank, world):
  """cfg3's partition on eight ranks (CPU, gloo): global batch 1024 -> 8 x 128,
  a ragged 1021-sample batch, gather of the generated shards, the shared shift
  stream over several steps, the gradient all-reduce + scalar mean."""
  out = {}
  batch = torch.arange(1024 * 3, dtype=torch.float32).reshape(1024, 3)
  mine = parallel.shard_batch(batch)
  out['n'] = len(mine)
  out['first'] = mine[:2, 0].tolist()
  gathered = parallel.gather_batch(mine * 2.0)
  out['gathered_ok'] = (None if gathered is None else
                        bool(torch.equal(gathered, batch * 2.0)))
  ragged = parallel.shard_batch(batch[:1021])
  out['ragged_n'] = len(ragged)
  g2 = parallel.gather_batch(ragged)
  out['ragged_gathered'] = (None if g2 is None else
                            bool(torch.equal(g2, batch[:1016])))
  out['tiny'] = parallel.shard_batch(batch[:5]) is None
  st = parallel.RandomStreams(1234, torch.device('cpu'), m=10)
  out['shifts'] = [st.shifts(3).tolist() for _ in range(4)]
  out['z0'] = st.noise(2, 4)[0].tolist()
  sync = parallel.GradSync()
  flat = torch.full((4110273 // 64,), float(rank))   # (a slice of the 16.4 MB buffer)
  h = sync.all_reduce_async(flat)
  h.wait()
  out['grad_mean'] = float(flat[0] * sync.grad_scale)
  out['scalars'] = sync.mean_scalars(torch.tensor([float(rank), 1.0])).tolist()
  out['choice'] = parallel.broadcast_object((14, 2, 0, 1) if rank == 0 else None)
  return out


def test_world8_partition_of_cfg3():
  """BASELINE configs[2] (8 x MI355X, global batch 1024) rehearsed on eight CPU
  ranks: what the first RCCL run will execute around its kernels."""
  out = _run(_world8_case, world=8)
  assert sorted(out) == list(range(8))
  for r in range(8):
    o = out[r]
    assert o['n'] == 128 and o['first'] == [3.0 * r, 3.0 * (r + 8)]
    assert o['ragged_n'] == 127 and o['tiny']
    assert o['gathered_ok'] is (True if r == 0 else None)
    assert o['ragged_gathered'] is (True if r == 0 else None)
    assert o['shifts'] == out[0]['shifts']            # one draw for all ranks
    assert r == 0 or o['z0'] != out[0]['z0']          # private noise
    np.testing.assert_allclose(o['grad_mean'], 3.5)   # mean of 0..7
    np.testing.assert_allclose(o['scalars'], [3.5, 1.0])
    assert tuple(o['choice']) == (14, 2, 0, 1)        # rank 0's tile choice
