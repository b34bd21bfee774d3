#!/usr/bin/env python
"""What an HBM-bound kernel costs when it runs BESIDE the step instead of inside
it (development probe, GPU only).  Times K graph-replayed train() calls at cfg2
(a) alone, (b) with N device-to-device copies of `mb` MB each enqueued on a side
stream before every replay (they overlap the step's MFMA kernels), (c) the same
copies alone.  If (b) - (a) is well below (c), moving interpolate + pack (5 x 364
MB per step) off the critical path would pay; if it is about (c), the step is
as HBM / power bound as the sum says.
  python tools/overlap_probe.py [steps] [copies per step] [MB read+written per copy]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench


def main():
  steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
  ncopy = int(sys.argv[2]) if len(sys.argv) > 2 else 4
  mb = int(sys.argv[3]) if len(sys.argv) > 3 else 364
  from calciumgan_amd.gan.algorithms import get_algorithm
  from calciumgan_amd.gan.models import get_models
  hp = bench.make_hparams(2048, 102, 64, 10)
  gen, dis = get_models(hp, None)
  gan = get_algorithm(hp, gen, dis, None)
  real = gan.batch_buffer(128)
  real.uniform_(0, 1)
  for _ in range(5):
    gan.train(real)
  torch.cuda.synchronize()
  src = torch.empty(mb * (1 << 20) // 2, dtype=torch.uint8, device='cuda')
  dst = torch.empty_like(src)
  side = torch.cuda.Stream()

  def run(with_step, with_copies):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
      if with_copies:
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
          for _ in range(ncopy):
            dst.copy_(src)
      if with_step:
        gan.train(real)
      if with_copies:
        torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3

  for rep in range(2):
    a = run(True, False)
    b = run(True, True)
    c = run(False, True)
    print('step alone %.3f ms | step + %d x %d MB copies beside it %.3f ms (+%.3f) | '
          'the copies alone %.3f ms' % (a, ncopy, mb, b, b - a, c), flush=True)


if __name__ == '__main__':
  main()
