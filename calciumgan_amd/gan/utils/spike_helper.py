"""gan/utils/spike_helper.py counterpart: OASIS AR(1) deconvolution
(g = 0.95, s_min = 0.55, threshold 0.5; spike_helper.py:23-54) on the host.

The arithmetic lives in csrc/oasis_ar1.c (gcc, built by
calciumgan_amd.build.build_host); `oasis_ar1_python` is the same algorithm in
pure python, kept as the cross-check the tests use.  OASIS upstream is an
un-pinned git clone (setup.sh:43) and absent here: PARITY UNPINNED."""
import ctypes
import os

import numpy as np

FRAME_RATE = 24.0  # spike_helper.py:8
_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, '..', '..', 'csrc', 'libcalciumgan_host.so')
_lib = None


def _load():
  global _lib
  if _lib is None:
    if not os.path.exists(_LIB_PATH):
      from ... import build
      build.build_host(verbose=False)
    lib = ctypes.CDLL(os.path.abspath(_LIB_PATH))
    dp = ctypes.POINTER(ctypes.c_double)
    lib.cg_oasis_ar1.argtypes = [dp, ctypes.c_int, ctypes.c_double,
                                 ctypes.c_double, ctypes.c_double, dp, dp]
    lib.cg_deconvolve.argtypes = [dp, ctypes.c_int, ctypes.c_int,
                                  ctypes.c_double, ctypes.c_double,
                                  ctypes.c_double,
                                  ctypes.POINTER(ctypes.c_float)]
    _lib = lib
  return _lib


def oasis_ar1(y, g, lam=0.0, s_min=0.0):
  """(c, s) of the AR(1) active-set deconvolution of one trace."""
  y = np.ascontiguousarray(y, dtype=np.float64)
  c = np.empty_like(y)
  s = np.empty_like(y)
  dp = ctypes.POINTER(ctypes.c_double)
  rc = _load().cg_oasis_ar1(y.ctypes.data_as(dp), len(y), g, lam, s_min,
                            c.ctypes.data_as(dp), s.ctypes.data_as(dp))
  if rc:
    raise RuntimeError('cg_oasis_ar1 failed: {}'.format(rc))
  return c, s


def oasis_ar1_python(y, g, lam=0.0, s_min=0.0):
  """Pure-python restatement (pools as lists) of the same algorithm."""
  y = np.asarray(y, dtype=np.float64)
  T = len(y)
  P = [[y[0] - lam * (1 - g), 1.0, 0, 1]]
  for t in range(1, T):
    P.append([y[t] - lam * (1 if t == T - 1 else (1 - g)), 1.0, t, 1])
    while len(P) > 1 and (P[-2][0] / P[-2][1] * g**P[-2][3] + s_min >
                          P[-1][0] / P[-1][1]):
      v, w, _, l = P.pop()
      gl = g**P[-1][3]
      P[-1][0] += v * gl
      P[-1][1] += w * gl * gl
      P[-1][3] += l
  c = np.empty(T)
  for v, w, t, l in P:
    c[t:t + l] = max(v / w, 0.0) * g**np.arange(l)
  s = np.zeros(T)
  s[1:] = c[1:] - g * c[:-1]
  return c, s


def oasis_function(signal, threshold=0.5):
  """spike_helper.py:23-29."""
  _, train = oasis_ar1(signal, g=0.95, s_min=.55)
  return np.where(train > threshold, 1.0, 0.0)


def deconvolve_signals(signals, threshold=0.5):
  """spike_helper.py:32-54: (rows, T) calcium traces -> float32 {0,1} trains."""
  if hasattr(signals, 'detach'):
    signals = signals.detach().cpu().numpy()
  signals = np.ascontiguousarray(signals, dtype=np.float64)
  assert signals.ndim == 2
  out = np.empty(signals.shape, dtype=np.float32)
  rc = _load().cg_deconvolve(
      signals.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), signals.shape[0],
      signals.shape[1], 0.95, 0.55, threshold,
      out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
  if rc:
    raise RuntimeError('cg_deconvolve failed: {}'.format(rc))
  return out
