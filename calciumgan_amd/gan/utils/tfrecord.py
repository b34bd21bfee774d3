"""TFRecord files of `signal` / `spike` segments without TensorFlow.

The reference stores its datasets as TFRecords (dataset/generate_tfrecords.py:
128-153: one tf.train.Example per segment with two bytes features, `signal` and
`spike`, each the raw float32 bytes of a (sequence_length, channels) array) and
reads them back with tf.data (gan/utils/dataset_helper.py:147-182).  This
module restates the two public formats involved so that such directories can
be read (and, for interchange and tests, written) here:

* TFRecord framing: uint64 length | masked crc32c(length) | data | masked
  crc32c(data), little endian, mask(c) = ((c >> 15 | c << 17) + 0xa282ead8).
* protobuf wire format of tf.train.Example:
    Example  { Features features = 1; }
    Features { map<string, Feature> feature = 1; }   (entry: key = 1, value = 2)
    Feature  { oneof { BytesList bytes_list = 1; FloatList float_list = 2;
                       Int64List int64_list = 3; } }
    BytesList { repeated bytes value = 1; }

TensorFlow is absent from this image, so no file written by the reference is
available: parity with TF-written files is UNPINNED; the format constants are
checked against the published known answers (RFC 3720 CRC-32C vectors) and a
hand-assembled record in tests/test_compat_io.py.  Host code, off the hot path.
"""
import ctypes
import glob
import os
import struct

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, '..', '..', 'csrc', 'libcalciumgan_host.so')
_lib = None
_MASK_DELTA = 0xa282ead8


def _host_lib():
  global _lib
  if _lib is None:
    if not os.path.exists(_LIB_PATH):
      from ... import build
      build.build_host(verbose=False)
    lib = ctypes.CDLL(_LIB_PATH)
    lib.cg_crc32c.argtypes = [ctypes.c_uint32, ctypes.c_char_p, ctypes.c_size_t]
    lib.cg_crc32c.restype = ctypes.c_uint32
    _lib = lib
  return _lib


def crc32c(data):
  return _host_lib().cg_crc32c(0, bytes(data), len(data))


def masked_crc32c(data):
  c = crc32c(data)
  return (((c >> 15) | (c << 17)) + _MASK_DELTA) & 0xffffffff


# ---------------------------------------------------------------------------
# protobuf wire format (only what tf.train.Example needs)
# ---------------------------------------------------------------------------
def _varint(n):
  out = bytearray()
  while True:
    b = n & 0x7f
    n >>= 7
    if n:
      out.append(b | 0x80)
    else:
      out.append(b)
      return bytes(out)


def _read_varint(buf, pos):
  shift = value = 0
  while True:
    b = buf[pos]
    pos += 1
    value |= (b & 0x7f) << shift
    if not b & 0x80:
      return value, pos
    shift += 7
    if shift > 63:
      raise ValueError('malformed varint')


def _len_field(number, payload):
  return _varint((number << 3) | 2) + _varint(len(payload)) + payload


def _fields(buf):
  """Yield (field_number, wire_type, value) of one message; value is a
  memoryview for length-delimited fields, an int otherwise."""
  pos, n = 0, len(buf)
  while pos < n:
    key, pos = _read_varint(buf, pos)
    number, wt = key >> 3, key & 7
    if wt == 2:
      ln, pos = _read_varint(buf, pos)
      if pos + ln > n:
        raise ValueError('truncated length-delimited field')
      yield number, wt, buf[pos:pos + ln]
      pos += ln
    elif wt == 0:
      v, pos = _read_varint(buf, pos)
      yield number, wt, v
    elif wt == 5:
      yield number, wt, struct.unpack_from('<I', buf, pos)[0]
      pos += 4
    elif wt == 1:
      yield number, wt, struct.unpack_from('<Q', buf, pos)[0]
      pos += 8
    else:
      raise ValueError('unsupported wire type {}'.format(wt))


def serialize_example(features):
  """{name: bytes} -> serialized tf.train.Example with one bytes_list value per
  feature (generate_tfrecords.py:128-138).  Map entries in sorted key order."""
  entries = b''
  for name in sorted(features):
    bytes_list = _len_field(1, bytes(features[name]))     # BytesList.value
    feature = _len_field(1, bytes_list)                   # Feature.bytes_list
    entry = _len_field(1, name.encode()) + _len_field(2, feature)
    entries += _len_field(1, entry)                       # Features.feature
  return _len_field(1, entries)                           # Example.features


def parse_example(data):
  """Serialized tf.train.Example -> {name: bytes} for its bytes features (the
  first value of each bytes_list, like FixedLenFeature([], tf.string))."""
  buf = memoryview(data)
  out = {}
  for number, wt, features in _fields(buf):
    if number != 1 or wt != 2:
      continue
    for n2, wt2, entry in _fields(features):
      if n2 != 1 or wt2 != 2:
        continue
      key, feature = None, None
      for n3, wt3, v in _fields(entry):
        if n3 == 1 and wt3 == 2:
          key = bytes(v).decode()
        elif n3 == 2 and wt3 == 2:
          feature = v
      if key is None or feature is None:
        continue
      for n4, wt4, blist in _fields(feature):
        if n4 == 1 and wt4 == 2:  # bytes_list
          for n5, wt5, value in _fields(blist):
            if n5 == 1 and wt5 == 2:
              out.setdefault(key, bytes(value))
  return out


# ---------------------------------------------------------------------------
# TFRecord framing
# ---------------------------------------------------------------------------
def frame(record):
  """One framed TFRecord: length | masked crc(length) | data | masked crc(data)."""
  header = struct.pack('<Q', len(record))
  return (header + struct.pack('<I', masked_crc32c(header)) + bytes(record) +
          struct.pack('<I', masked_crc32c(record)))


class TFRecordWriter(object):
  """tf.io.TFRecordWriter counterpart (generate_tfrecords.py:150-153)."""

  def __init__(self, path):
    self._f = open(path, 'wb')

  def write(self, record):
    self._f.write(frame(record))

  def close(self):
    self._f.close()

  def __enter__(self):
    return self

  def __exit__(self, *exc):
    self.close()


def read_records(path, verify=True):
  """Yield the records of one TFRecord file; checksums verified by default
  (a mismatch raises IOError, like TF's DataLossError)."""
  with open(path, 'rb') as f:
    while True:
      header = f.read(8)
      if not header:
        return
      if len(header) < 8:
        raise IOError('{}: truncated record header'.format(path))
      length = struct.unpack('<Q', header)[0]
      crc_h = f.read(4)
      data = f.read(length)
      crc_d = f.read(4)
      if len(crc_h) < 4 or len(data) < length or len(crc_d) < 4:
        raise IOError('{}: truncated record'.format(path))
      if verify:
        if struct.unpack('<I', crc_h)[0] != masked_crc32c(header):
          raise IOError('{}: corrupted record length'.format(path))
        if struct.unpack('<I', crc_d)[0] != masked_crc32c(data):
          raise IOError('{}: corrupted record data'.format(path))
      yield data


def record_filename(output_dir, mode, shard_id, num_shards):
  """generate_tfrecords.py:141-143."""
  return os.path.join(output_dir, '{}-{:03d}-of-{:03d}.record'.format(
      mode, shard_id + 1, num_shards))


def write_segments(path, signals, spikes):
  """One Example per segment: raw float32 bytes of signal and spike
  (generate_tfrecords.py:132-153; `spike` is float32 there too,
  dataset_helper.py:161)."""
  with TFRecordWriter(path) as w:
    for sig, spk in zip(signals, spikes):
      w.write(serialize_example({
          'signal': np.ascontiguousarray(sig, np.float32).tobytes(),
          'spike': np.ascontiguousarray(spk, np.float32).tobytes(),
      }))


def read_segments(pattern, signal_shape, spike_shape, verify=True):
  """All segments of the files matching `pattern` (sorted), decoded like
  dataset_helper.py:154-165 -> (signals f32 (N,)+signal_shape, spikes f32)."""
  files = sorted(glob.glob(pattern))
  if not files:
    raise FileNotFoundError('no TFRecord file matches {}'.format(pattern))
  sig, spk = [], []
  for path in files:
    for rec in read_records(path, verify):
      ex = parse_example(rec)
      if 'signal' not in ex or 'spike' not in ex:
        raise ValueError('{}: record without signal / spike feature'.format(path))
      sig.append(np.frombuffer(ex['signal'], np.float32).reshape(signal_shape))
      spk.append(np.frombuffer(ex['spike'], np.float32).reshape(spike_shape))
  return np.stack(sig), np.stack(spk)
