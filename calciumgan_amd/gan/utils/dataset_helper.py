"""Input pipeline for the hot path (gan/utils/dataset_helper.py:113-206).

The reference reads TFRecords of raw-float32 `signal` / `spike` features.  A
dataset directory here holds an ``info.pkl`` with the reference's keys
(dataset/generate_tfrecords.py:229-248) next to EITHER the reference's
``train-*.record`` / ``validation-*.record`` files (decoded by
``tfrecord.py``, no TensorFlow) OR the same segments as ``train.npy`` /
``validation.npy`` dicts (what dataset/generate_dg_dataset.py writes).  Batches
are float32 (B, L, C) in [0, 1]; the last batch of an epoch may be short
(no drop_remainder, dataset_helper.py:173)."""
import json
import os
import pickle
from math import ceil

import numpy as np

from . import h5_helper


class LazyRows(object):
  """Rows `index` of a host array, gathered when first looked at.  The train
  loop never reads the spike half of a batch (reference main.py:39 `for signal,
  _ in train_ds`); gathered eagerly, 128 x 2048 x 102 spikes per step were
  ~10 ms of host time in front of every train() -- as long as the step itself
  (main.py at cfg2: 24 -> 13 ms per step)."""

  def __init__(self, array, index):
    self._array, self._index, self._rows = array, index, None

  def _get(self):
    if self._rows is None:
      self._rows = self._array[self._index]
    return self._rows

  def __array__(self, dtype=None, copy=None):
    r = self._get()
    return r if dtype is None else r.astype(dtype)

  def __getitem__(self, k):
    return self._get()[k]

  def __len__(self):
    return len(self._index)

  @property
  def shape(self):
    return (len(self._index),) + tuple(self._array.shape[1:])

  @property
  def dtype(self):
    return self._array.dtype


class ArrayDataset(object):
  """Iterable of (signal, spike) batches; reshuffled every epoch when asked
  (tf.data shuffle(buffer) + batch, dataset_helper.py:170-174).

  With `gather_into` bound (main.py: WGAN_GP.batch_buffer) every batch of that
  size is gathered into ONE device buffer, so the `signal` tensors this iterator
  yields ALIAS each other: a consumer that keeps a reference across steps sees
  the later batch (the training loop reads it within the step; clone to keep)."""

  def __init__(self, signals, spikes, batch_size, shuffle, seed=1234):
    self.signals, self.spikes = signals, spikes
    self.batch_size, self.shuffle = batch_size, shuffle
    self._rng = np.random.RandomState(seed)
    self._dev_signals = None
    # callable(batch_len) -> device tensor to gather the batch into, or None
    # (main.py binds the training set to WGAN_GP.batch_buffer)
    self.gather_into = None

  def to_device(self, device):
    """Keep the whole signal set resident in HBM (8192 x 2048 x 102 f32 is
    6.8 GB of 288 GB): batches are then gathered on the device and the train
    loop never waits for a host copy.  Spikes stay on the host (unused by
    training, main.py:39)."""
    import torch
    self._dev_signals = torch.from_numpy(np.ascontiguousarray(
        self.signals)).to(device)
    return self

  def __len__(self):
    return ceil(len(self.signals) / self.batch_size)

  def __iter__(self):
    idx = np.arange(len(self.signals))
    if self.shuffle:
      self._rng.shuffle(idx)
    for i in range(0, len(idx), self.batch_size):
      j = np.sort(idx[i:i + self.batch_size])
      if self._dev_signals is not None:
        import torch
        jj = torch.from_numpy(j).to(self._dev_signals.device)
        out = self.gather_into(len(j)) if self.gather_into is not None else None
        if out is not None:
          # (the SAME tensor every step: the consumer is done with batch k
          # before batch k + 1 is gathered -- one stream, a sequential loop)
          torch.index_select(self._dev_signals, 0, jj, out=out)
          yield out, LazyRows(self.spikes, j)
        else:
          yield self._dev_signals.index_select(0, jj), LazyRows(self.spikes, j)
      else:
        yield self.signals[j], LazyRows(self.spikes, j)


def get_dataset_info(hparams):
  """dataset_helper.py:113-144."""
  with open(os.path.join(hparams.input_dir, 'info.pkl'), 'rb') as file:
    info = pickle.load(file)
  hparams.train_size = info['train_size']
  hparams.validation_size = info['validation_size']
  hparams.signal_shape = tuple(info['signal_shape'])
  hparams.spike_shape = tuple(info['spike_shape'])
  hparams.sequence_length = info['sequence_length']
  hparams.num_neurons = info['num_neurons']
  hparams.num_channels = info['num_channels']
  hparams.normalize = info['normalize']
  hparams.fft = info['fft']
  hparams.conv2d = info['conv2d']
  if hparams.normalize:
    hparams.signals_min = float(info['signals_min'])
    hparams.signals_max = float(info['signals_max'])
  if hparams.save_generated:
    hparams.generated_dir = os.path.join(hparams.output_dir, 'generated')
    if getattr(hparams, 'rank', 0) == 0:
      os.makedirs(hparams.generated_dir, exist_ok=True)
    hparams.validation_cache = os.path.join(hparams.generated_dir,
                                            'validation.h5')


def cache_validation_set(hparams, validation):
  """dataset_helper.py:12-30: validation signals (denormalised) + spikes in
  generated/validation.h5, written once -- per (batch size, world size): the
  cache holds exactly the samples a run validates (ragged tails dropped under
  data parallelism), so one written by an earlier run of the same output_dir
  with another batch / world size would pair compute_metrics' recorded and
  generated trials wrongly.  Its geometry sits beside it and a mismatch rewrites
  it."""
  sig = validation['signals']
  spikes = validation['spikes']
  world = getattr(hparams, 'world_size', 1)
  keep = validated_samples(len(sig), hparams.batch_size, world)
  meta = dict(batch_size=int(hparams.batch_size), world_size=int(world),
              kept=int(len(keep)), total=int(len(sig)))
  meta_file = hparams.validation_cache + '.json'
  if os.path.exists(hparams.validation_cache):
    try:
      with open(meta_file) as f:
        if json.load(f) == meta:
          return
    except (OSError, ValueError):
      pass
    os.remove(hparams.validation_cache)
  if len(keep) != len(sig):
    sig, spikes = sig[keep], spikes[keep]
  if hparams.normalize:
    sig = sig * (hparams.signals_max - hparams.signals_min) + hparams.signals_min
  h5_helper.write(hparams.validation_cache, {
      'signals': sig.astype(np.float32),
      'spikes': spikes.astype(np.int8)
  })
  with open(meta_file, 'w') as f:
    json.dump(meta, f)


def validated_samples(num_samples, batch_size, world_size):
  """Indices of the validation samples a run really validates, in order.  Data
  parallel: every rank takes r::world of the largest prefix of a batch that
  divides evenly (parallel.shard_batch), so a ragged last batch loses up to
  world - 1 samples; generated/validation.h5 holds exactly the samples the
  generated files have counterparts for (compute_metrics pairs them by index)."""
  if world_size <= 1:
    return np.arange(num_samples)
  keep = [np.arange(s, s + (min(batch_size, num_samples - s) // world_size) *
                    world_size) for s in range(0, num_samples, batch_size)]
  return np.concatenate(keep) if keep else np.arange(0)


SURROGATE_TRAIN_SIZE = 8192  # dataset_helper.py:76


def get_surrogate_dataset(hparams):
  """dataset_helper.py:54-110: <input_dir>/training.pkl holds
  {'signals': (trials, neurons, time), 'spikes': ...}; signals become
  (trials, time, neurons), are min-max normalised over the WHOLE array (the
  bounds land in hparams.signals_min / signals_max), and the first 8192
  trials train, the rest validate."""
  filename = os.path.join(hparams.input_dir, 'training.pkl')
  if not os.path.exists(filename):
    print('training dataset {} not found'.format(filename))
    exit()
  with open(filename, 'rb') as file:
    data = pickle.load(file)
  signals = np.ascontiguousarray(
      np.transpose(np.asarray(data['signals'], dtype=np.float32), (0, 2, 1)))
  spikes = np.asarray(data['spikes'])
  hparams.signals_min = float(signals.min())
  hparams.signals_max = float(signals.max())
  signals = ((signals - hparams.signals_min) /
             (hparams.signals_max - hparams.signals_min)).astype(np.float32)
  n = SURROGATE_TRAIN_SIZE
  hparams.train_size = len(signals[:n])
  hparams.validation_size = len(signals[n:])
  hparams.signal_shape = tuple(signals.shape[1:])
  hparams.spike_shape = tuple(spikes.shape[1:])
  hparams.sequence_length = signals.shape[1]
  hparams.num_neurons = hparams.num_channels = signals.shape[-1]
  hparams.normalize, hparams.fft, hparams.conv2d = True, False, False
  if hparams.save_generated:
    hparams.generated_dir = os.path.join(hparams.output_dir, 'generated')
    if getattr(hparams, 'rank', 0) == 0:
      os.makedirs(hparams.generated_dir, exist_ok=True)
    hparams.validation_cache = os.path.join(hparams.generated_dir,
                                            'validation.h5')
  return (ArrayDataset(signals[:n], spikes[:n], hparams.batch_size,
                       shuffle=True),
          ArrayDataset(signals[n:], spikes[n:], hparams.batch_size,
                       shuffle=False))


def get_dataset(hparams, summary=None):
  """dataset_helper.py:185-206."""
  hparams.noise_shape = (hparams.noise_dim,)
  if not os.path.exists(hparams.input_dir):
    print('input directory {} cannot be found'.format(hparams.input_dir))
    exit()
  if getattr(hparams, 'surrogate_ds', False):
    train_ds, validation_ds = get_surrogate_dataset(hparams)
    hparams.train_steps = ceil(hparams.train_size / hparams.batch_size)
    hparams.validation_steps = ceil(hparams.validation_size /
                                    hparams.batch_size)
    return train_ds, validation_ds
  get_dataset_info(hparams)
  if os.path.exists(os.path.join(hparams.input_dir, 'train.npy')):
    with open(os.path.join(hparams.input_dir, 'train.npy'), 'rb') as f:
      train = pickle.load(f)
    with open(os.path.join(hparams.input_dir, 'validation.npy'), 'rb') as f:
      validation = pickle.load(f)
  else:
    # the reference's own layout: train-*.record / validation-*.record
    # (dataset_helper.py:116-118,147-182), decoded and cached in memory
    from . import tfrecord
    train, validation = ({
        'signals': sig, 'spikes': spk
    } for sig, spk in (tfrecord.read_segments(
        os.path.join(hparams.input_dir, mode + '-*.record'),
        hparams.signal_shape, hparams.spike_shape)
                       for mode in ('train', 'validation')))
  if hparams.save_generated and getattr(hparams, 'rank', 0) == 0:
    cache_validation_set(hparams, validation)  # one writer under torchrun
  train_ds = ArrayDataset(train['signals'], train['spikes'], hparams.batch_size,
                          shuffle=True)
  validation_ds = ArrayDataset(validation['signals'], validation['spikes'],
                               hparams.batch_size, shuffle=False)
  hparams.train_steps = ceil(hparams.train_size / hparams.batch_size)
  hparams.validation_steps = ceil(hparams.validation_size / hparams.batch_size)
  return train_ds, validation_ds


def write_dataset(output_dir, signals, spikes, info, validation_size,
                  tfrecords=False, num_per_shard=0):
  """Counterpart of dataset/generate_tfrecords.py:186-252: shuffle, split
  off `validation_size` segments, write train/validation + info.pkl.  With
  `tfrecords` the segments go into the reference's sharded
  ``<mode>-NNN-of-MMM.record`` files (generate_tfrecords.py:141-183; shard
  split as its `split()`), else into one pickled array dict per mode."""
  os.makedirs(output_dir, exist_ok=True)
  rng = np.random.RandomState(1234)  # generate_tfrecords.py:14
  idx = rng.permutation(len(signals))
  train_size = len(signals) - validation_size
  parts = {'train': idx[:train_size], 'validation': idx[train_size:]}
  shards = {'train': 1, 'validation': 1}
  for name, ii in parts.items():
    if tfrecords:
      from . import tfrecord
      n = 1 if num_per_shard <= 0 else ceil(len(ii) / num_per_shard)
      shards[name] = n
      k, m = divmod(len(ii), n)
      for s in range(n):
        part = ii[s * k + min(s, m):(s + 1) * k + min(s + 1, m)]
        tfrecord.write_segments(
            tfrecord.record_filename(output_dir, name, s, n), signals[part],
            spikes[part])
    else:
      with open(os.path.join(output_dir, name + '.npy'), 'wb') as f:
        pickle.dump({'signals': signals[ii], 'spikes': spikes[ii]}, f,
                    protocol=4)
  full = dict(info)
  full.update(train_size=train_size, validation_size=validation_size,
              num_train_shards=shards['train'],
              num_validation_shards=shards['validation'],
              buffer_size=train_size)
  with open(os.path.join(output_dir, 'info.pkl'), 'wb') as f:
    pickle.dump(full, f)
  return full
