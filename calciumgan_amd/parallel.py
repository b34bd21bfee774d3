"""Data parallelism for the WGAN-GP step: one process per GPU, batch sharded
by sample, ONE RCCL all-reduce per optimizer update over the model's flat
gradient buffer: 5 x 16.4 MB (critic) + 1 x 17.5 MB (generator)
per train() at cfg2.  The reference has no distributed code; this is new.

Randomness: noise z and interpolation alpha are per-rank streams; the
PhaseShuffle shifts come from a generator seeded identically on every rank so
"one shift per layer per call for the whole (global) batch" is preserved.
"""
import os

import torch
import torch.distributed as dist


def env_world():
  return int(os.environ.get('WORLD_SIZE', '1'))


def env_rank():
  return int(os.environ.get('RANK', '0'))


def env_local_rank():
  return int(os.environ.get('LOCAL_RANK', '0'))


def init_process_group(backend=None):
  """Initialise torch.distributed from the torchrun environment (backend
  'nccl' is RCCL on ROCm; 'gloo' for CPU tests).  No-op for world size 1."""
  if env_world() <= 1 or dist.is_initialized():
    return
  if backend is None:
    backend = 'nccl' if torch.cuda.is_available() else 'gloo'
  if torch.cuda.is_available():
    # one GPU per rank; the modulo only matters for single-GPU rehearsals of
    # the multi-rank path with the gloo backend
    torch.cuda.set_device(env_local_rank() % torch.cuda.device_count())
  dist.init_process_group(backend=backend)


def world_size():
  return dist.get_world_size() if dist.is_initialized() else 1


def rank():
  return dist.get_rank() if dist.is_initialized() else 0


class GradSync(object):
  """Sum-all-reduce of a flat gradient buffer; the division by world size is
  folded into the optimizer's grad_scale (no extra pass over the gradients)."""

  def __init__(self):
    self.world = world_size()
    self.grad_scale = 1.0 / self.world

  def all_reduce(self, flat_grad):
    if self.world > 1:
      dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
    return flat_grad

  def all_reduce_async(self, flat_grad):
    """Start the sum-all-reduce and return its work handle (None for one
    rank): `handle.wait()` orders the current stream after the collective, so
    launches issued in between overlap it."""
    if self.world > 1:
      return dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, async_op=True)
    return None

  def mean_scalars(self, t):
    """Average a small tensor of logged scalars across ranks (in place)."""
    if self.world > 1:
      dist.all_reduce(t, op=dist.ReduceOp.SUM)
      t.mul_(1.0 / self.world)
    return t


class RandomStreams(object):
  """Per-rank streams for z / alpha, a rank-shared stream for phase shifts."""

  def __init__(self, seed, device, m):
    self.m = int(m)
    self.device = device
    r = rank()
    if device.type == 'cuda':
      self.local = torch.Generator(device=device)
    else:
      self.local = torch.Generator()
    self.local.manual_seed(seed * 1000003 + 7919 * (r + 1))
    self.shared = torch.Generator()  # CPU, same on all ranks
    self.shared.manual_seed(seed)

  def noise(self, batch_size, noise_dim):
    return torch.randn(batch_size, noise_dim, generator=self.local,
                       device=self.device, dtype=torch.float32)

  def alpha(self, batch_size):
    return torch.rand(batch_size, generator=self.local, device=self.device,
                      dtype=torch.float32)

  def shifts(self, nseg):
    """int32 (4, nseg): U{-m..m} per layer per discriminator call."""
    return torch.randint(-self.m, self.m + 1, (4, nseg), generator=self.shared,
                         dtype=torch.int32)


def barrier():
  if dist.is_initialized():
    dist.barrier()


def shard_batch(batch, r=None, world=None):
  """Rank r's samples of a global batch: r::world of the largest prefix that
  divides evenly, so every rank holds the SAME number of samples and the
  (1/world) * sum of local-mean gradients is the global mean.  Up to world - 1
  samples of a ragged last batch are dropped; returns None when the batch is
  smaller than the world (every rank then skips it -- all ranks see the same
  batch, so the decision is consistent and no collective is left hanging)."""
  world = world_size() if world is None else world
  r = rank() if r is None else r
  if world <= 1:
    return batch
  n = (len(batch) // world) * world
  if n == 0:
    return None
  return batch[:n][r::world]


def gather_batch(local, dst=0):
  """Inverse of shard_batch for equal shards: the global batch (sample order
  restored) on rank dst, None elsewhere.  Used for the generated samples of a
  validation pass (files are written by rank 0 only)."""
  if not dist.is_initialized() or dist.get_world_size() == 1:
    return local
  world = dist.get_world_size()
  parts = [torch.empty_like(local) for _ in range(world)]
  dist.all_gather(parts, local.contiguous())
  if dist.get_rank() != dst:
    return None
  out = torch.stack(parts, dim=1)  # (n_local, world, ...): sample i*world + r
  return out.reshape((-1,) + tuple(local.shape[1:]))


def broadcast_object(obj, src=0):
  """A small picklable object from rank src to every rank (tile choices)."""
  if not dist.is_initialized() or dist.get_world_size() == 1:
    return obj
  box = [obj]
  dist.broadcast_object_list(box, src=src)
  return box[0]
