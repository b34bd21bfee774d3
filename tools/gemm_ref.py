#!/usr/bin/env python
"""Context for the swconv roofline fraction: what the vendor GEMM (torch.matmul
-> hipBLASLt) reaches on this device for plain bf16 GEMMs with the M/N/K of the
CalciumGAN layers (no im2col cost, no epilogue).  Development tool."""
import torch


def run(M, N, K, iters=20):
  a = torch.randn(M, K, device='cuda').to(torch.bfloat16)
  b = torch.randn(K, N, device='cuda').to(torch.bfloat16)
  torch.matmul(a, b)
  torch.cuda.synchronize()
  s = torch.cuda.Event(enable_timing=True)
  e = torch.cuda.Event(enable_timing=True)
  s.record()
  for _ in range(iters):
    torch.matmul(a, b)
  e.record()
  torch.cuda.synchronize()
  t = s.elapsed_time(e) / iters * 1e-3
  print('gemm M%d N%d K%d: %.1f us  %.1f TF/s' % (M, N, K, t * 1e6,
                                                   2.0 * M * N * K / t / 1e12))


if __name__ == '__main__':
  run(384 * 1024, 64, 24 * 128)     # D conv 1 (3B batch)
  run(384 * 512, 128, 24 * 64)      # D conv 2
  run(384 * 256, 192, 24 * 128)     # D conv 3
  run(384 * 128, 256, 24 * 192)     # D conv 4
  run(384 * 64, 320, 24 * 256)      # D conv 5
  run(384 * 256, 128, 12 * 192)     # dgrad phase
  run(8192, 8192, 8192)             # square reference
