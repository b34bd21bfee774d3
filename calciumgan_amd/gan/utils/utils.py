"""gan/utils/utils.py counterparts: run utilities around the hot path
(hparams persistence, checkpoints, generated-sample files).  On-disk layouts
follow SURVEY Appendix C."""
import json
import os
import pickle
import subprocess
from glob import glob

import numpy as np

from . import h5_helper


def normalize(x, x_min, x_max):
  """utils.py:25-27."""
  return (x - x_min) / (x_max - x_min)


def denormalize(x, x_min, x_max):
  """utils.py:30-32."""
  return x * (x_max - x_min) + x_min


def _to_numpy(x):
  if hasattr(x, 'detach'):
    x = x.detach().cpu().numpy()
  return np.asarray(x)


def reverse_preprocessing(hparams, x):
  """utils.py:49-63 (1-D, no fft: denormalise only)."""
  x = _to_numpy(x)
  if hparams.normalize:
    x = denormalize(x, x_min=hparams.signals_min, x_max=hparams.signals_max)
  return x


def get_current_git_hash():
  """utils.py:66-69, tolerant of running outside a git checkout (the
  reference crashes there -- SURVEY Appendix D.7)."""
  try:
    return subprocess.check_output(['git', 'describe', '--always'],
                                   stderr=subprocess.DEVNULL).strip().decode()
  except Exception:  # noqa
    return 'unknown'


def _jsonable(v):
  if isinstance(v, (np.integer,)):
    return int(v)
  if isinstance(v, (np.floating,)):
    return float(v)
  if isinstance(v, np.ndarray):
    return v.tolist()
  if isinstance(v, tuple):
    return list(v)
  return v


def save_hparams(hparams):
  """utils.py:72-75."""
  hparams.git_hash = get_current_git_hash()
  os.makedirs(hparams.output_dir, exist_ok=True)
  with open(os.path.join(hparams.output_dir, 'hparams.json'), 'w') as file:
    json.dump({k: _jsonable(v) for k, v in hparams.__dict__.items()}, file)


def load_hparams(hparams):
  """utils.py:78-84."""
  filename = os.path.join(hparams.output_dir, 'hparams.json')
  with open(filename, 'r') as file:
    content = json.load(file)
  for key, value in content.items():
    if not hasattr(hparams, key):
      setattr(hparams, key, value)


def save_fake_signals(hparams, epoch, signals):
  """utils.py:93-113: append denormalised (N, L, C) float32 signals to
  generated/epoch{:03d}_signals.h5 and record it in generated/info.pkl."""
  signals = reverse_preprocessing(hparams, signals)
  filename = os.path.join(hparams.generated_dir,
                          'epoch{:03d}_signals.h5'.format(epoch))
  h5_helper.write(filename, {'signals': signals.astype(np.float32)})
  info_filename = os.path.join(hparams.generated_dir, 'info.pkl')
  info = {}
  if os.path.exists(info_filename):
    with open(info_filename, 'rb') as file:
      info = pickle.load(file)
  if epoch not in info:
    info[epoch] = {'global_step': hparams.global_step, 'filename': filename}
    with open(info_filename, 'wb') as file:
      pickle.dump(info, file)


def save_models(hparams, gan, epoch):
  """utils.py:116-132: checkpoints/epoch-{:03d}.pkl with Keras-order weight
  lists; the step counters are plain ints (the reference pickles tf.Variables,
  SURVEY 5.4)."""
  if not os.path.exists(hparams.ckpt_dir):
    os.makedirs(hparams.ckpt_dir)
  filename = os.path.join(hparams.ckpt_dir, 'epoch-{:03d}.pkl'.format(epoch))
  with open(filename, 'wb') as file:
    pickle.dump({
        'epoch': epoch,
        'gen_weights': gan.generator.get_weights(),
        'dis_weights': gan.discriminator.get_weights(),
        'gen_steps': int(gan.gen_optimizer.iterations),
        'dis_steps': int(gan.dis_optimizer.iterations)
    }, file)
  if hparams.verbose:
    print('Saved checkpoint to {}'.format(filename))


def load_models(hparams, gan):
  """utils.py:135-152: resume from the lexicographically last epoch-*."""
  if not hasattr(hparams, 'ckpt_dir'):
    hparams.ckpt_dir = os.path.join(hparams.output_dir, 'checkpoints')
  hparams.start_epoch = 0
  filenames = glob(os.path.join(hparams.ckpt_dir, 'epoch-*'))
  if filenames:
    filename = sorted(filenames)[-1]
    with open(filename, 'rb') as file:
      ckpt = pickle.load(file)
    hparams.start_epoch = ckpt['epoch'] + 1
    gan.generator.set_weights(ckpt['gen_weights'])
    gan.discriminator.set_weights(ckpt['dis_weights'])
    gan.gen_optimizer.iterations = int(ckpt['gen_steps'])
    gan.dis_optimizer.iterations = int(ckpt['dis_steps'])
    if hparams.verbose:
      print('\n\nRestored checkpoint at {}\n\n'.format(filename))


def generate_dataset(hparams, gan, num_samples=1000):
  """utils.py:191-207: sample the generator in batches of 100 into
  <output_dir>/generated.pkl."""
  generated = np.zeros((num_samples,) + tuple(hparams.signal_shape),
                       dtype=np.float32)
  batch_size = 100
  for i in range(0, num_samples, batch_size):
    noise = gan.get_noise(batch_size)
    signals = _to_numpy(gan.generate(noise, denorm=True))
    generated[i:i + batch_size] = signals[:num_samples - i]
  filename = os.path.join(hparams.output_dir, 'generated.pkl')
  with open(filename, 'wb') as file:
    pickle.dump({'signals': generated}, file)
  if hparams.verbose:
    print('save {} samples to {}'.format(num_samples, filename))


def get_array_format(shape, hparams):
  """utils.py:155-165: 'N' samples / 'W' sequence length / 'C' channels."""
  assert len(shape) <= 3
  return ''.join('W' if s == hparams.sequence_length else
                 'C' if s == hparams.num_neurons else 'N' for s in shape)


def set_array_format(array, data_format, hparams):
  """utils.py:168-184."""
  assert len(array.shape) == len(data_format)
  current_format = get_array_format(array.shape, hparams)
  assert set(current_format) == set(data_format)
  if data_format == current_format:
    return array
  return np.transpose(array, axes=[current_format.index(s) for s in data_format])


def remove_nan(array):
  """utils.py:187-188."""
  return array[np.logical_not(np.isnan(array))]
