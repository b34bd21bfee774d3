"""Layer 1 of the penalty's x^ segment through its Gram operator (DESIGN section 9,
a lead for the next round): float64 check of the three identities on random data,
sample edges included.  CPU only; the oracle's Conv1D is the ground truth.

  conv (calciumgan.py:159-166, k = 24, s = 2, 'same'):  y[t] = sum_k x[2t + k - 11] W[k]
  g   = d<delta, y>/dx            (the input gradient the penalty takes the norm of)
  G_t'[l] = sum_{k' valid at t'} W[k' - 2l]^T W[k']   (23 lags, Co x Co; 'valid': the
            row 2t' + k' - 11 of g exists, i.e. lies in [0, L) -- all k' in the
            interior, fewer in the first / last six rows of a sample)
  (1) ||g||^2              = <delta, G * delta>
  (2) conv(c g)            = c (G * delta)           (the tangent chain's first layer)
  (3) d<conv(c g), d2>/dW  = c sum_l W[k - 2l] R_k[l],  R_k[l] = sum_t delta[t + l] (x) d2[t]
                             over the rows t whose window row 2t + k - 11 exists
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle as O  # noqa: E402

K, S, PL = 24, 2, 11


def gram(W, T):
  """G[t'][l + 11] (Co x Co) for every output row t' of a sample of T rows."""
  k, ci, co = W.shape
  A = np.einsum('kic,lid->klcd', W, W)  # A[k, k'] = W[k]^T W[k']
  G = np.zeros((T, 2 * PL + 1, co, co))
  for tp in range(T):
    for l in range(-PL, PL + 1):
      for kp in range(k):
        kk = kp - 2 * l
        r = 2 * tp + kp - PL
        if 0 <= kk < k and 0 <= r < S * T:
          G[tp, l + PL] += A[kk, kp]
  return G


def apply_gram(G, delta):
  """(G * delta)[t', co'] = sum_l delta[t' + l] G[t'][l]."""
  T = delta.shape[0]
  y = np.zeros_like(delta)
  for tp in range(T):
    for l in range(-PL, PL + 1):
      if 0 <= tp + l < T:
        y[tp] += delta[tp + l] @ G[tp, l + PL]
  return y


def check(T=20, ci=5, co=4, seed=0):
  rng = np.random.RandomState(seed)
  W = rng.randn(K, ci, co)
  delta = rng.randn(T, co)
  d2 = rng.randn(T, co)
  c = 0.37
  Wt = torch.tensor(W, requires_grad=True)
  x = torch.zeros(1, S * T, ci, dtype=torch.float64, requires_grad=True)
  y = O.conv1d_same(x, Wt, None, S)
  g, = torch.autograd.grad((y[0] * torch.tensor(delta)).sum(), x, create_graph=True)
  G = gram(W, T)
  Gd = apply_gram(G, delta)
  # (1) the penalty norm
  np.testing.assert_allclose(float((g.detach()**2).sum()), float((delta * Gd).sum()), rtol=1e-12)
  # (2) the tangent's first layer (before the mask)
  t1 = O.conv1d_same(c * g, Wt, None, S)
  np.testing.assert_allclose(t1[0].detach().numpy(), c * Gd, rtol=1e-11, atol=1e-12)
  # (3) layer 1's weight gradient from the penalty's second backward: v = c g is a
  # constant input there (its own dependence on W is the OTHER term of the product
  # rule, which the step takes through delta's chain)
  v = (c * g).detach()
  dW, = torch.autograd.grad((O.conv1d_same(v, Wt, None, S)[0] * torch.tensor(d2)).sum(), Wt)
  want = np.zeros_like(W)
  for k in range(K):
    for l in range(-PL, PL + 1):
      kk = k - 2 * l
      if not 0 <= kk < K:
        continue
      R = np.zeros((co, co))
      for t in range(T):
        if 0 <= t + l < T and 0 <= 2 * t + k - PL < S * T:
          R += np.outer(delta[t + l], d2[t])
      want[k] += c * W[kk] @ R
  np.testing.assert_allclose(dW.numpy(), want, rtol=1e-11, atol=1e-12)
  # the interior kernel is shift-invariant; only the first / last six rows differ
  interior = G[6:T - 6]
  assert np.abs(interior - interior[0]).max() < 1e-12
  assert np.abs(G[5] - G[6]).max() > 1e-6 and np.abs(G[T - 6] - G[T - 7]).max() > 1e-6
  macs_now = 3 * (T * K * ci * co)              # input gradient, tangent layer 1, dW's x^ third
  macs_gram = 2 * (T * (2 * PL + 1) * co * co)  # G * delta and the correlation R
  return macs_gram / macs_now


if __name__ == '__main__':
  r = check()
  print('identities hold (float64); MACs per sample at these toy widths: %.2f of today\'s' % r)
  print('cfg2 (Ci 102 -> pitch 128, Co 64): %.2f of today\'s three launches' % (
      2 * 23 * 64 * 64 / (3 * 24 * 128 * 64.0)))
