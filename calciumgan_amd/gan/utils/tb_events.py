"""TensorBoard event files without TensorFlow (scalars only).

The reference logs through tf.summary (gan/utils/summary_helper.py:32-40,
98-101: one FileWriter per directory, `tf.summary.scalar(tag, value, step)`).
An event file is a TFRecord stream (framing + masked CRC-32C: tfrecord.py) of
`Event` protobufs; what a scalar needs of the schema (tensorflow/core/util/
event.proto, summary.proto -- public formats, restated):

  Event   { double wall_time = 1; int64 step = 2;
            oneof { string file_version = 3; Summary summary = 5; } }
  Summary { repeated Value value = 1; }
  Value   { string tag = 1; float simple_value = 2; }

The first record of a file is Event{file_version: "brain.Event:2"}.  TF2's
tf.summary.scalar stores a tensor-valued Value with plugin metadata instead of
simple_value; TensorBoard's scalar dashboard reads both forms.  TensorFlow and
TensorBoard are absent from this image: reading these files with TensorBoard
is UNPINNED; tests check the framing, CRCs and a decode of the protobufs.
Host code, off the hot path.
"""
import os
import socket
import struct
import time

from . import tfrecord

_VERSION = b'brain.Event:2'


def _key(field, wire):
  return tfrecord._varint((field << 3) | wire)


def _len_delim(field, payload):
  return _key(field, 2) + tfrecord._varint(len(payload)) + payload


def encode_event(wall_time, step=0, tag=None, value=None, file_version=None):
  ev = _key(1, 1) + struct.pack('<d', wall_time)
  if step:
    ev += _key(2, 0) + tfrecord._varint(int(step) & 0xffffffffffffffff)
  if file_version is not None:
    ev += _len_delim(3, file_version)
  if tag is not None:
    val = _len_delim(1, tag.encode('utf-8')) + _key(2, 5) + struct.pack(
        '<f', float(value))
    ev += _len_delim(5, _len_delim(1, val))
  return ev


def decode_event(buf):
  """-> dict(wall_time, step, file_version | (tag, value)); test helper."""
  out = dict(step=0)
  pos = 0
  while pos < len(buf):
    key, pos = tfrecord._read_varint(buf, pos)
    field, wire = key >> 3, key & 7
    if wire == 1:
      out['wall_time'] = struct.unpack('<d', buf[pos:pos + 8])[0]
      pos += 8
    elif wire == 0:
      out['step'], pos = tfrecord._read_varint(buf, pos)
    elif wire == 2:
      n, pos = tfrecord._read_varint(buf, pos)
      payload = buf[pos:pos + n]
      pos += n
      if field == 3:
        out['file_version'] = bytes(payload)
      elif field == 5:
        # Summary { Value value = 1 } -> Value { tag = 1; simple_value = 2 }
        k, p = tfrecord._read_varint(payload, 0)
        assert k == (1 << 3 | 2)
        n2, p = tfrecord._read_varint(payload, p)
        val = payload[p:p + n2]
        q = 0
        while q < len(val):
          k, q = tfrecord._read_varint(val, q)
          if k == (1 << 3 | 2):
            n3, q = tfrecord._read_varint(val, q)
            out['tag'] = bytes(val[q:q + n3]).decode('utf-8')
            q += n3
          elif k == (2 << 3 | 5):
            out['value'] = struct.unpack('<f', val[q:q + 4])[0]
            q += 4
          else:
            raise ValueError('unexpected field in Summary.Value')
    else:
      raise ValueError('unexpected wire type {}'.format(wire))
  return out


class EventFileWriter(object):
  """Appends scalar events to <logdir>/events.out.tfevents.<time>.<host>."""

  def __init__(self, logdir):
    os.makedirs(logdir, exist_ok=True)
    self.path = os.path.join(
        logdir, 'events.out.tfevents.{:010d}.{}'.format(
            int(time.time()), socket.gethostname()))
    with open(self.path, 'ab') as f:
      f.write(tfrecord.frame(encode_event(time.time(), file_version=_VERSION)))

  def scalar(self, tag, value, step=0):
    with open(self.path, 'ab') as f:
      f.write(tfrecord.frame(encode_event(time.time(), step, tag, value)))


def read_events(path):
  """All events of a file (CRCs checked); test helper."""
  return [decode_event(rec) for rec in tfrecord.read_records(path)]
