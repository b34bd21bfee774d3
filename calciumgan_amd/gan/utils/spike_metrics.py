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
