"""Helpers for the GPU parity tests: call single kernels of the C ABI on torch
tensors and build small-integer (bf16-exact) test data."""
import ctypes

import numpy as np
import torch

from calciumgan_amd import _lib
from calciumgan_amd import geometry as geo
from calciumgan_amd import nets

DEV = 'cuda'
BF16 = torch.bfloat16


def stream():
  return nets._stream()


def p(t):
  return nets._p(t)


def to_pitch(x, cp, dtype=BF16):
  """(B, L, C) f32 cpu -> (B, L, Cp) device tensor, zero padded channels."""
  B, L, C = x.shape
  out = torch.zeros(B, L, cp, dtype=dtype, device=DEV)
  out[:, :, :C] = x.to(DEV).to(dtype)
  return out


def int_tensor(rng, shape, lo=-3, hi=3, scale=1.0):
  """Small integers (times a power-of-two scale): exact in bf16, products and
  moderate sums exact in f32 -> kernels can be checked bit-for-bit."""
  return torch.tensor(
      rng.randint(lo, hi + 1, size=shape).astype(np.float32) * scale)


def pack(src_dev, phases, C_real, N_real, Cx, CK, taps, parity_major=False):
  op = nets.PackedOperand(src_dev, phases, C_real, N_real, Cx, CK, taps,
                          parity_major=parity_major)
  op.repack()
  return op


def numpy_pack(wl, Cx, CK):
  """Reference packing of a logical operand Wl[tap][c][n] (numpy f32) into the
  layout documented in include/calciumgan_hip.h / swconv.hip."""
  taps, C, N = wl.shape
  c8 = CK // 8
  nchunks = Cx // CK
  Fp = (taps * c8 + 15) // 16 * 16
  Npad = (N + 63) // 64 * 64
  out = np.zeros((Npad, nchunks, Fp, 8), np.float32)
  for tap in range(taps):
    for c in range(C):
      cc, r = divmod(c, CK)
      q8, e = divmod(r, 8)
      out[:N, cc, tap * c8 + q8, e] = wl[tap, c, :]
  return out.reshape(Npad, -1)


def swconv(x, op, y, *, nB, Lx, Cx, taps, stride, off, Lu, N, Ly, Cy, **kw):
  d = nets._conv_desc(x, op.buf, y, nB, Lx, Cx, taps, stride, off, Lu, N, Ly,
                      Cy, op.CK, **kw)
  small = kw.pop('force_small', None)
  _lib.call('cg_swconv', ctypes.byref(d), stream())
  return d


def conv_desc(*a, **kw):
  return nets._conv_desc(*a, **kw)


def run_conv(d):
  _lib.call('cg_swconv', ctypes.byref(d), stream())


def run_wgrad(d):
  _lib.call('cg_wgrad', ctypes.byref(d), stream())


def sync():
  torch.cuda.synchronize()


_WS = {}


def reduce_ws():
  """Workspace of the ordered reductions (cg_reduce_ws_elems floats), filled
  with NaN: a kernel that read a slot nobody wrote would show it."""
  if 'ws' not in _WS:
    _WS['ws'] = torch.empty(_lib.load().cg_reduce_ws_elems(), dtype=torch.float32,
                            device=DEV)
  _WS['ws'].fill_(float('nan'))
  return _WS['ws']


def out_buffers(ws, *shapes):
  """Output tensors of a reducing kernel: zeros for the atomics form (it adds),
  a poison value for the ordered form (it must store)."""
  fill = 0.0 if ws is None else 12345.0
  return [torch.full(s if isinstance(s, tuple) else (s,), fill, device=DEV)
          for s in shapes]
