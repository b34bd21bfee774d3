"""Synthetic dichotomised-Gaussian (DG) calcium dataset -- the benchmark /
parity input of BASELINE.json.

Host-side numpy restatement of the reference's DG recipe:
  * binary spikes = [z + gamma > 0], z ~ N(0, Lambda) per time bin
    (dataset/dg/dichot_gauss.py:145-179), gamma = Phi^-1(p)
    (dataset/dg/optim_dichot_gauss.py:109-126);
  * calcium c_t = s_t + g*c_{t-1} for t >= 2, signal = c + sn*N(0,1)
    (dataset/generate_dg_data.py:54-70, g = .95, sn = .3);
  * stride-2 windows of length L, global min/max normalisation to [0, 1]
    (dataset/generate_tfrecords.py:82-86, :113-120).
The recorded-data statistics the reference fits (generate_dg_data.py:16-39) are
not available offline, so rates follow SURVEY 8(d): r_hz =
clip(exp(N(-2.5, 1.2^2)), 0.005, 2.0) at 24 Hz, equicorrelation rho = 0.05.
"""
import numpy as np
from scipy.stats import norm

FRAME_RATE = 24.0


def dg_parameters(num_neurons, rng, rho=0.05):
  rates_hz = np.clip(np.exp(rng.normal(-2.5, 1.2, size=num_neurons)), 0.005, 2.0)
  p = rates_hz / FRAME_RATE
  gamma = norm.ppf(p)  # gauss mean, optim_dichot_gauss.py:125
  corr = (1.0 - rho) * np.eye(num_neurons) + rho * np.ones(
      (num_neurons, num_neurons))
  return rates_hz, gamma, corr


def sample_spikes(gamma, rho, duration, rng):
  """(num_neurons, duration) float32 in {0,1}.  z ~ N(0, (1-rho)I + rho 11^T)
  drawn through its one-factor form (same law as scipy's mnorm.rvs used at
  dichot_gauss.py:173-176)."""
  n = gamma.shape[0]
  eps = rng.standard_normal((duration, n))
  eta = rng.standard_normal((duration, 1))
  z = np.sqrt(1.0 - rho) * eps + np.sqrt(rho) * eta
  return (z + gamma[None, :] > 0).astype(np.float32).T


def spikes_to_signals(spikes, rng, g=0.95, sn=0.3, b=0.0):
  """generate_dg_data.py:54-70 (the AR recursion starts at t = 2)."""
  c = spikes.astype(np.float32).copy()
  for i in range(2, c.shape[1]):
    c[:, i] += g * c[:, i - 1]
  noise = rng.standard_normal(c.shape)
  return (b + c + sn * noise).astype(np.float32)


def segment(raw, sequence_length, stride=2, max_segments=None):
  """generate_tfrecords.py:82-86: windows i, i+stride, ... while
  i + L < T; raw is (T, C); returns (N, L, C)."""
  starts = np.arange(0, raw.shape[0] - sequence_length, stride)
  if max_segments is not None:
    starts = starts[:max_segments]
  idx = starts[:, None] + np.arange(sequence_length)[None, :]
  return raw[idx]


def make_dataset(num_neurons=102, sequence_length=2048, num_segments=9192,
                 seed=1234, stride=2, rho=0.05):
  """Returns dict(signals (N,L,C) float32 in [0,1], spikes (N,L,C), info)."""
  rng = np.random.RandomState(seed)  # generate_dg_data.py:9
  rates, gamma, _ = dg_parameters(num_neurons, rng, rho)
  duration = sequence_length + stride * num_segments
  spikes = sample_spikes(gamma, rho, duration, rng)
  signals = spikes_to_signals(spikes, rng)
  sig = segment(signals.T, sequence_length, stride, num_segments)
  spk = segment(spikes.T, sequence_length, stride, num_segments)
  smin, smax = float(sig.min()), float(sig.max())
  sig = (sig - smin) / (smax - smin)
  info = dict(
      signal_shape=(sequence_length, num_neurons),
      spike_shape=(sequence_length, num_neurons),
      sequence_length=sequence_length,
      num_neurons=num_neurons,
      num_channels=num_neurons,
      normalize=True,
      stride=stride,
      fft=False,
      conv2d=False,
      signals_min=smin,
      signals_max=smax,
      rates_hz=rates)
  return dict(signals=sig.astype(np.float32), spikes=spk.astype(np.float32),
              info=info)
