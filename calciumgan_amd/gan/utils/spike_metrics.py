"""gan/utils/spike_metrics.py counterparts in numpy (Elephant / Neo are not
installed; they are un-pinned upstream -- PARITY UNPINNED, definitions below
follow Elephant's documented statistics on binary trains at 24 Hz).

spikes: (neurons, T) arrays of {0,1} at FRAME_RATE frames per second."""
import numpy as np

from .spike_helper import FRAME_RATE


def mean_firing_rate(spikes):
  """spike_metrics.py:6-12: spikes per second over [0, T / 24 s) per neuron
  (elephant.statistics.mean_firing_rate of the Neo train built at
  spike_helper.py:8-14)."""
  spikes = np.asarray(spikes)
  duration = spikes.shape[-1] / FRAME_RATE
  return (spikes.sum(axis=-1) / duration).astype(np.float32)


def bin_counts(spikes, binsize_ms=500.0):
  """BinnedSpikeTrain(binsize=500 ms) counts: 12 frames per bin at 24 Hz; the
  trailing partial bin is dropped as Elephant does."""
  spikes = np.asarray(spikes)
  per_bin = int(round(binsize_ms / 1000.0 * FRAME_RATE))
  nb = spikes.shape[-1] // per_bin
  return spikes[..., :nb * per_bin].reshape(spikes.shape[:-1] +
                                            (nb, per_bin)).sum(-1)


def covariance(spikes1, spikes2=None, binsize_ms=500.0):
  """spike_metrics.py:28-38: covariance matrix of the binned counts
  (unbiased, N-1), cross block when spikes2 is given."""
  spikes = spikes1 if spikes2 is None else np.concatenate([spikes1, spikes2], 0)
  cov = np.atleast_2d(np.cov(bin_counts(spikes, binsize_ms)))
  if spikes2 is not None:
    cov = cov[len(spikes1):, :len(spikes2)]
  return cov


def correlation_coefficients(spikes1, spikes2=None, binsize_ms=500.0):
  """spike_metrics.py:15-25 (Pearson r of binned counts; NaN for silent
  trains, like Elephant)."""
  spikes = spikes1 if spikes2 is None else np.concatenate([spikes1, spikes2], 0)
  with np.errstate(invalid='ignore', divide='ignore'):
    r = np.atleast_2d(np.corrcoef(bin_counts(spikes, binsize_ms)))
  if spikes2 is not None:
    r = r[len(spikes1):, :len(spikes2)]
  return r


def spike_times(spikes):
  """spike_helper.py:8-20 (train_to_neo / trains_to_neo without Neo): per train
  the spike times in seconds (frame / 24 Hz); t_stop = T / 24 s."""
  spikes = np.asarray(spikes)
  assert spikes.ndim == 2
  return [np.nonzero(tr)[0] / float(FRAME_RATE) for tr in spikes]


def _exp_sum(a, b, tau):
  """sum_k sum_l exp(-|a_k - b_l| / tau)."""
  if len(a) == 0 or len(b) == 0:
    return 0.0
  return float(np.exp(-np.abs(a[:, None] - b[None, :]) / tau).sum())


def van_rossum_distance(spikes1, spikes2=None, tau=1.0):
  """spike_metrics.py:41-51 -> elephant.spike_train_dissimilarity.van_rossum_dist
  (tau = 1 s default [ext]).  Elephant evaluates the closed form of Houghton &
  Kreuz 2012 on the summed kernel matrix S_ab = sum_k sum_l exp(-|t_k - t_l| /
  tau) and returns D[i, j] = sqrt(S_ii + S_jj - S_ij - S_ji) -- the
  normalisation in which ONE spike against an empty train is at distance 1
  (trains convolved with sqrt(2 / tau) exp(-t / tau) H(t)), without the factor
  1/2 of the plain exp(-t / tau) kernel (round 3 carried that factor:
  van_rossum_heatmap_min in spike_metrics.json was off by sqrt(2); the KL
  statistics do not depend on the normalisation).  Returns the full matrix, or
  the (spikes2 x spikes1) cross block exactly as the reference slices it
  (result[len(spikes1):, :len(spikes2)]).  PARITY UNPINNED (Elephant absent:
  the formula is restated from its published source, not run)."""
  spikes = np.asarray(spikes1) if spikes2 is None else np.concatenate(
      [np.asarray(spikes1), np.asarray(spikes2)], 0)
  # S = A E A^T: E the kernel between ALL spikes of the batch, A the train
  # membership (one matrix product instead of n^2 python-level pair sums: a
  # 102-neuron trial is 5 000 pairs)
  owner, frame = np.nonzero(spikes)
  n = len(spikes)
  t = frame / float(FRAME_RATE)
  E = np.exp(-np.abs(t[:, None] - t[None, :]) / tau)
  A = np.zeros((n, len(t)), np.float64)
  A[owner, np.arange(len(t))] = 1.0
  S = A @ E @ A.T
  d2 = np.diag(S)[:, None] + np.diag(S)[None, :] - 2.0 * S
  result = np.sqrt(np.maximum(d2, 0.0))
  if spikes2 is not None:
    result = result[len(spikes1):, :len(spikes2)]
  return result


def victor_purpura_distance(spikes1, spikes2=None, q=1.0):
  """spike_metrics.py:54-63 -> elephant victor_purpura_dist (q = 1 Hz default
  [ext]): minimal cost of turning one train into the other with insert / delete
  (cost 1) and shifts (cost q |dt|); dynamic programme of Victor & Purpura 1996.
  PARITY UNPINNED."""
  spikes = np.asarray(spikes1) if spikes2 is None else np.concatenate(
      [np.asarray(spikes1), np.asarray(spikes2)], 0)
  times = spike_times(spikes)
  n = len(times)
  result = np.zeros((n, n), np.float64)
  for i in range(n):
    for j in range(i + 1, n):
      a, b = times[i], times[j]
      G = np.zeros((len(a) + 1, len(b) + 1))
      G[:, 0] = np.arange(len(a) + 1)
      G[0, :] = np.arange(len(b) + 1)
      for k in range(1, len(a) + 1):
        for l in range(1, len(b) + 1):
          G[k, l] = min(G[k - 1, l] + 1, G[k, l - 1] + 1,
                        G[k - 1, l - 1] + q * abs(a[k - 1] - b[l - 1]))
      result[i, j] = result[j, i] = G[-1, -1]
  if spikes2 is not None:
    result = result[len(spikes1):, :len(spikes2)]
  return result
