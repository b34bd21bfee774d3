#!/usr/bin/env python
"""Post-hoc spike statistics of generated vs recorded signals -- the numbers of
the reference's compute_metrics.py (BASELINE configs[3]: recorded-data pipeline
+ OASIS deconvolution metrics), without its plots:

  deconvolve the generated calcium signals (OASIS AR(1), stored back into the
  generated file as `spikes`)                          compute_metrics.py:35-57
  firing rate per neuron over the trials, KL(recorded || synthetic)    :210-262
  pairwise correlation coefficients per trial, KL                      :306-358
  van Rossum distances between the neurons of a trial, KL; recorded x
  synthetic distance heatmaps for the chosen neurons                   :361-480
  (covariance KL: defined :265-303 but not called by the reference's main; same
  here)

  python compute_metrics.py --output_dir runs/001 [--all_epochs]

The KL is the reference's: both samples cut into 30 equal-width bins over their
pooled range by pandas.cut, empty bins replaced by 1e-10 (:80-111) -- pandas is
the same library here, so that step is shared, not restated.  Elephant / Neo /
OASIS are absent and un-pinned upstream (PARITY UNPINNED); the statistics come
from gan/utils/spike_metrics.py and csrc/oasis_ar1.c.  Results go to
<output_dir>/spike_metrics.json and, as the reference does, 'elapse/spike_metrics'
to the scalar log.
"""
import argparse
import json
import multiprocessing
import os
import pickle
import random
from time import time

import numpy as np
import pandas as pd

from calciumgan_amd.gan.utils import h5_helper, spike_helper, spike_metrics, utils

NUM_BINS = 30  # compute_metrics.py:96


def kl_divergence(p, q):
  """compute_metrics.py:80-84."""
  p = np.where(p == 0, 1e-10, p)
  q = np.where(q == 0, 1e-10, q)
  return np.sum(p * np.log(p / q))


def pairs_kl_divergence(pairs):
  """compute_metrics.py:87-111: per (recorded, synthetic) pair the KL of their
  histograms over NUM_BINS equal-width bins of the pooled values."""
  kl = np.zeros((len(pairs),), dtype=np.float32)
  for i, (real, fake) in enumerate(pairs):
    real, fake = np.asarray(real), np.asarray(fake)
    pooled = np.concatenate([real, fake])
    bins = np.asarray(pd.cut(pooled, bins=NUM_BINS, labels=np.arange(NUM_BINS)))
    is_real = np.arange(len(pooled)) < len(real)
    real_pdf = np.array([np.sum(bins[is_real] == b) for b in range(NUM_BINS)],
                        dtype=np.float32) / len(real)
    fake_pdf = np.array([np.sum(bins[~is_real] == b) for b in range(NUM_BINS)],
                        dtype=np.float32) / len(fake)
    kl[i] = kl_divergence(real_pdf, fake_pdf)
  return kl


def _spikes(hparams, filename, data_format, neuron=None, trial=None,
            num_trials=None):
  """get_neo_trains (:60-74) up to the Neo conversion: a 2-D {0,1} array in
  `data_format` ('NW' for one neuron over trials, 'CW' for one trial)."""
  assert data_format and (neuron is not None or trial is not None)
  spikes = h5_helper.get(filename, name='spikes', neuron=neuron, trial=trial)
  spikes = utils.set_array_format(np.asarray(spikes), data_format, hparams)
  if num_trials is not None:
    assert data_format[0] == 'N'
    spikes = spikes[:num_trials]
  return spikes.astype(np.float32)


def _deconvolve_neuron(hparams, filename, neuron):
  signals = h5_helper.get(filename, name='signals', neuron=neuron)
  signals = utils.set_array_format(np.asarray(signals), 'NW', hparams)
  return spike_helper.deconvolve_signals(signals, threshold=0.5)


def deconvolve_from_file(hparams, filename):
  """compute_metrics.py:42-57: spikes of every neuron of the generated set,
  stored as int8 NWC beside the signals."""
  if hparams.verbose:
    print('\tDeconvolve {}'.format(filename))
  per_neuron = _map(hparams, _deconvolve_neuron,
                    [(hparams, filename, n) for n in range(hparams.num_neurons)])
  spikes = utils.set_array_format(np.array(per_neuron, dtype=np.int8), 'NWC',
                                  hparams)
  h5_helper.write(filename, {'spikes': spikes})
  return spikes


def firing_rate(hparams, filename, neuron, num_trials=200):
  """:210-232: (recorded, synthetic) firing rates of one neuron, one value per
  trial."""
  real = _spikes(hparams, hparams.validation_cache, 'NW', neuron=neuron,
                 num_trials=num_trials)
  fake = _spikes(hparams, filename, 'NW', neuron=neuron, num_trials=num_trials)
  return (spike_metrics.mean_firing_rate(real),
          spike_metrics.mean_firing_rate(fake))


def _upper(matrix, n):
  return utils.remove_nan(np.asarray(matrix)[np.triu_indices(n, k=1)])


def covariance(hparams, filename, trial):
  """:265-281."""
  n = hparams.num_neurons
  return tuple(_upper(spike_metrics.covariance(_spikes(hparams, f, 'CW',
                                                       trial=trial)), n)
               for f in (hparams.validation_cache, filename))


def correlation_coefficient(hparams, filename, trial):
  """:306-324."""
  n = hparams.num_neurons
  return tuple(
      _upper(spike_metrics.correlation_coefficients(
          _spikes(hparams, f, 'CW', trial=trial)), n)
      for f in (hparams.validation_cache, filename))


def trial_van_rossum(hparams, filename, trial):
  """:415-443: distances between the neurons of one trial, upper triangle."""
  out = []
  for f in (hparams.validation_cache, filename):
    d = spike_metrics.van_rossum_distance(_spikes(hparams, f, 'CW', trial=trial))
    out.append(d[np.triu_indices(len(d), k=1)])
  assert out[0].shape == out[1].shape
  return tuple(out)


def sort_heatmap(matrix):
  """:361-386: rows / columns reordered so that the minimum sits top left --
  columns by the row holding the global minimum, then for every column in turn
  the not-yet-used row that is smallest there."""
  matrix = np.asarray(matrix, dtype=np.float32)
  n = len(matrix)
  work = matrix.copy()
  first_row = np.unravel_index(np.argmin(matrix), matrix.shape)[0]
  column_order = np.argsort(matrix[first_row])
  row_order = np.full((n,), -1, dtype=np.int64)
  heatmap = np.full(matrix.shape, np.nan, dtype=np.float32)
  for i in range(n):
    row_order[i] = first_row if i == 0 else np.argsort(work[:, column_order[i]])[0]
    heatmap[i] = matrix[row_order[i]][column_order]
    work[row_order[i], :] = np.inf
  return heatmap, row_order, column_order


def neuron_van_rossum(hparams, filename, neuron, num_trials=50):
  """:389-412: recorded x synthetic distances of one neuron's first trials."""
  real = _spikes(hparams, hparams.validation_cache, 'NW', neuron=neuron,
                 num_trials=num_trials)
  fake = _spikes(hparams, filename, 'NW', neuron=neuron, num_trials=num_trials)
  heatmap, rows, cols = sort_heatmap(
      spike_metrics.van_rossum_distance(real, fake))
  return dict(heatmap=heatmap, xticklabels=rows, yticklabels=cols)


def _map(hparams, fn, args):
  """pool.starmap of the reference (one pool per metric, :45,:240,...)."""
  if getattr(hparams, 'num_processors', 1) > 1 and len(args) > 1:
    with multiprocessing.Pool(hparams.num_processors) as pool:
      return pool.starmap(fn, args)
  return [fn(*a) for a in args]


def compute_epoch_spike_metrics(hparams, filename, epoch):
  """:483-497 without plot_signals / raster_plots: dict of the statistics."""
  if not h5_helper.contains(filename, 'spikes'):
    deconvolve_from_file(hparams, filename)
  out = {}
  pairs = _map(hparams, firing_rate, [(hparams, filename, n, hparams.num_samples)
                                      for n in range(hparams.num_neurons)])
  kl = pairs_kl_divergence(pairs)
  out['firing_rate_kl'] = dict(
      mean=float(np.mean(kl)),
      neurons={int(n): float(kl[n]) for n in hparams.neurons})
  if hparams.verbose:
    print('\tfiring rate        KL mean: {:.04f}'.format(np.mean(kl)))
  trials = [(hparams, filename, i) for i in range(hparams.num_samples)]
  kl = pairs_kl_divergence(_map(hparams, correlation_coefficient, trials))
  out['correlation_kl'] = dict(mean=float(np.mean(kl)))
  if hparams.verbose:
    print('\tcorrelation        KL mean: {:.04f}'.format(np.mean(kl)))
  heat = _map(hparams, neuron_van_rossum,
              [(hparams, filename, n, 45) for n in hparams.neurons])
  out['van_rossum_heatmap_min'] = {
      int(n): float(np.nanmin(h['heatmap'])) for n, h in zip(hparams.neurons, heat)}
  kl = pairs_kl_divergence(_map(hparams, trial_van_rossum, trials))
  out['van_rossum_kl'] = dict(mean=float(np.mean(kl)))
  if hparams.verbose:
    print('\tvan Rossum         KL mean: {:.04f}'.format(np.mean(kl)))
  return out


def main(hparams):
  """compute_metrics.py:500-542."""
  if not os.path.exists(hparams.output_dir):
    print('{} not found'.format(hparams.output_dir))
    exit()
  random.seed(hparams.seed)
  np.random.seed(hparams.seed)
  utils.load_hparams(hparams)
  with open(os.path.join(hparams.generated_dir, 'info.pkl'), 'rb') as f:
    info = pickle.load(f)
  hparams.num_samples = min(
      h5_helper.get_dataset_length(hparams.validation_cache, 'signals'), 1000)
  # neurons / trials the reference picks for its plots; the heatmaps and the
  # per-neuron KL lines follow the same choice
  hparams.neurons = (list(range(hparams.num_neurons))
                     if hparams.num_neuron_plots >= hparams.num_neurons else
                     list(np.random.choice(hparams.num_neurons,
                                           hparams.num_neuron_plots)))
  hparams.trials = list(np.random.choice(hparams.num_samples,
                                         hparams.num_trial_plots))
  epochs = sorted(info.keys())
  if not hparams.all_epochs:  # only the last generated file
    epochs = [epochs[-1]]
  report = {}
  for epoch in epochs:
    start = time()
    if hparams.verbose:
      print('\nCompute metrics for {}'.format(info[epoch]['filename']))
    report[int(epoch)] = compute_epoch_spike_metrics(
        hparams, filename=info[epoch]['filename'], epoch=epoch)
    report[int(epoch)]['elapse'] = time() - start
    with open(os.path.join(hparams.output_dir, 'scalars.jsonl'), 'a') as f:
      f.write(json.dumps({'tag': 'elapse/spike_metrics',
                          'value': report[int(epoch)]['elapse'],
                          'step': int(epoch)}) + '\n')
  with open(os.path.join(hparams.output_dir, 'spike_metrics.json'), 'w') as f:
    json.dump(report, f, indent=1)
  return report


def build_parser():
  parser = argparse.ArgumentParser()
  parser.add_argument('--output_dir', default='runs')
  parser.add_argument('--num_processors', default=6, type=int)
  parser.add_argument('--all_epochs', action='store_true')
  parser.add_argument('--num_neuron_plots', default=6, type=int)
  parser.add_argument('--num_trial_plots', default=6, type=int)
  parser.add_argument('--plots_per_row', default=3, type=int)
  parser.add_argument('--dpi', default=120, type=int)
  parser.add_argument('--format', default='pdf', choices=['pdf', 'png'])
  parser.add_argument('--verbose', default=1, type=int)
  parser.add_argument('--seed', default=12, type=int)
  return parser


if __name__ == '__main__':
  main(build_parser().parse_args())
