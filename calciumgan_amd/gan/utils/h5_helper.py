"""gan/utils/h5_helper.py counterpart: append-able NWC datasets.

Uses h5py when it is installed (same file layout as the reference: dataset
`name`, chunked, maxshape (None, ...), h5_helper.py:11-27).  libhdf5 / h5py are
absent from the build image, so the fallback stores every dataset as numbered
``.npy`` chunks inside a directory named like the .h5 file -- same logical NWC
content and the same function surface (write / append / overwrite / get /
get_dataset_length / contains)."""
import os
from glob import glob

import numpy as np

try:  # pragma: no cover - not installed in the build image
  import h5py
except ImportError:  # noqa
  h5py = None


def _chunks(filename, name):
  return sorted(glob(os.path.join(filename, '{}.*.npy'.format(name))))


def write(filename, content):
  """write or append content (dict name -> NWC array), h5_helper.py:11-27."""
  assert type(content) == dict
  if h5py is not None:
    with h5py.File(filename, mode='a') as file:
      for k, v in content.items():
        if k in file:
          ds = file[k]
          ds.resize((ds.shape[0] + v.shape[0]), axis=0)
          ds[-v.shape[0]:] = v
        else:
          file.create_dataset(k, shape=v.shape, dtype=v.dtype, data=v,
                              chunks=True, maxshape=(None,) + v.shape[1:])
    return
  os.makedirs(filename, exist_ok=True)
  for k, v in content.items():
    n = len(_chunks(filename, k))
    np.save(os.path.join(filename, '{}.{:06d}.npy'.format(k, n)), np.asarray(v))


def overwrite(filename, name, value):
  """h5_helper.py:30-36."""
  if h5py is not None:
    with h5py.File(filename, mode='r+') as file:
      if name not in file.keys():
        raise KeyError('{} cannot be found'.format(name))
      del file[name]
      file.create_dataset(name, shape=value.shape, dtype=value.dtype, data=value)
    return
  if not contains(filename, name):
    raise KeyError('{} cannot be found'.format(name))
  for f in _chunks(filename, name):
    os.remove(f)
  write(filename, {name: value})


def get(filename, name, neuron=None, trial=None):
  """h5_helper.py:39-56 (datasets are NWC)."""
  assert not (neuron is not None and trial is not None)
  if h5py is not None:
    with h5py.File(filename, mode='r') as file:
      if name not in file.keys():
        raise KeyError('{} cannot be found'.format(name))
      ds = file[name]
      if neuron is not None:
        return ds[:, :, neuron]
      if trial is not None:
        return ds[trial, :, :]
      return ds[:]
  files = _chunks(filename, name)
  if not files:
    raise KeyError('{} cannot be found'.format(name))
  ds = np.concatenate([np.load(f) for f in files], axis=0)
  if neuron is not None:
    return ds[:, :, neuron]
  if trial is not None:
    return ds[trial, :, :]
  return ds


def get_dataset_length(filename, name):
  """h5_helper.py:59-63."""
  if h5py is not None:
    with h5py.File(filename, mode='r') as file:
      return file[name].len()
  return int(sum(np.load(f, mmap_mode='r').shape[0]
                 for f in _chunks(filename, name)))


def contains(filename, name):
  """h5_helper.py:66-69."""
  if h5py is not None:
    with h5py.File(filename, mode='r') as file:
      return name in list(file.keys())
  return len(_chunks(filename, name)) > 0
