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
