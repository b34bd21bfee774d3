#!/usr/bin/env python
"""Mean-firing-rate / pairwise-covariance report of generated vs validation
data (compute_dg_metrics.py:40-58, :146-201 of the reference): deconvolve the
calcium signals of `num_trials` samples with OASIS AR(1), compute per-neuron
firing rates and the upper-triangular covariance of 500-ms-binned counts, and
print MAE / RMSE / MAPE.

  python compute_dg_metrics.py --output_dir runs/001 [--num_trials 5]
"""
import argparse
import os
import pickle

import numpy as np

from calciumgan_amd.gan.utils import h5_helper, spike_helper, spike_metrics, utils


def get_data_statistics(hparams, filename):
  """compute_dg_metrics.py:40-58."""
  n = hparams.num_neurons
  firing_rates = np.zeros((n, hparams.num_trials), np.float32)
  covariances = np.zeros((n * (n + 1) // 2, hparams.num_trials), np.float32)
  have_spikes = h5_helper.contains(filename, 'spikes')
  for i in range(hparams.num_trials):
    if have_spikes:  # validation cache stores the ground-truth trains
      spikes = h5_helper.get(filename, 'spikes', trial=i).T.astype(np.float32)
    else:
      signals = h5_helper.get(filename, 'signals', trial=i).T  # (C, W)
      spikes = spike_helper.deconvolve_signals(signals)
    firing_rates[:, i] = spike_metrics.mean_firing_rate(spikes)
    cov = spike_metrics.covariance(spikes)
    covariances[:, i] = np.nan_to_num(cov[np.triu_indices(len(cov))])
  return firing_rates, covariances


def percentage_error(y_true, y_pred):
  """compute_dg_metrics.py:146-153."""
  error = np.empty(y_true.shape)
  for j in range(y_true.shape[0]):
    if y_true[j] != 0.0:
      error[j] = (y_true[j] - y_pred[j]) / y_true[j]
    else:
      error[j] = y_pred[j] / np.mean(y_true)
  return error


def mean_absolute_percentage_error(y_true, y_pred):
  """compute_dg_metrics.py:156-162."""
  errors = np.zeros(y_true.shape, np.float32)
  for i in range(errors.shape[1]):
    errors[..., i] = percentage_error(y_true[..., i], y_pred[..., i])
  return float(np.mean(np.mean(np.abs(errors), axis=0), axis=0) * 100)


def report(real_fr, fake_fr, real_cov, fake_cov):
  return {
      'firing_rate': dict(
          mae=float(np.mean(np.abs(real_fr - fake_fr))),
          rmse=float(np.sqrt(np.mean(np.square(real_fr - fake_fr)))),
          mape=mean_absolute_percentage_error(real_fr, fake_fr)),
      'covariance': dict(
          mae=float(np.mean(np.abs(real_cov - fake_cov))),
          mse=float(np.mean(np.square(real_cov - fake_cov))),
          mape=mean_absolute_percentage_error(real_cov, fake_cov)),
  }


def main(hparams):
  if not os.path.exists(hparams.output_dir):
    print('{} not found'.format(hparams.output_dir))
    exit()
  utils.load_hparams(hparams)
  with open(os.path.join(hparams.generated_dir, 'info.pkl'), 'rb') as f:
    info = pickle.load(f)
  epochs = sorted(info.keys())
  real_fr, real_cov = get_data_statistics(hparams, hparams.validation_cache)
  fake_fr, fake_cov = get_data_statistics(hparams, info[epochs[-1]]['filename'])
  r = report(real_fr, fake_fr, real_cov, fake_cov)
  print('\nmean firing rate\n\tMAE\t{mae:.02f}\n\tRMSE\t{rmse:.02f}\n\tMAPE\t'
        '{mape:.02f}%'.format(**r['firing_rate']))
  print('\ncovariance\n\tMAE\t{mae:.02f}\n\tMSE\t{mse:.02f}\n\tMAPE\t'
        '{mape:.02f}%'.format(**r['covariance']))
  path = os.environ.get('CALCIUMGAN_METRICS_JSON')
  if path:  # (development: the same numbers at full precision, plus the means)
    import json
    r['population'] = dict(real_rate=float(np.mean(real_fr)),
                           fake_rate=float(np.mean(fake_fr)),
                           real_cov=float(np.mean(real_cov)),
                           fake_cov=float(np.mean(fake_cov)))
    with open(path, 'w') as f:
      json.dump(r, f)
  return r


if __name__ == '__main__':
  parser = argparse.ArgumentParser()
  parser.add_argument('--output_dir', default='runs')
  parser.add_argument('--num_trials', default=5, type=int)
  main(parser.parse_args())
