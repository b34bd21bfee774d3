"""CPU tests of the compatibility layer around the hot path (SURVEY 8(f)
rows 3-4, Appendix C): CLI flag surface, hparams.json, checkpoint pickle
layout + resume, generated-sample store, scalar tags, dataset directory."""
import glob
import json
import os
import pickle
import struct
from types import SimpleNamespace

import numpy as np
import pytest

import main as cli
from calciumgan_amd.data import dg
from calciumgan_amd.gan.utils import dataset_helper, h5_helper, utils
from calciumgan_amd.gan.utils.summary_helper import Summary


class _FakeModel(object):

  def __init__(self, weights):
    self._w = [w.copy() for w in weights]

  def get_weights(self):
    return [w.copy() for w in self._w]

  def set_weights(self, w):
    self._w = [np.asarray(a).copy() for a in w]


def _fake_gan():
  rng = np.random.RandomState(0)
  g = _FakeModel([rng.randn(3, 4).astype(np.float32), rng.randn(4)])
  d = _FakeModel([rng.randn(24, 2, 5).astype(np.float32)])
  return SimpleNamespace(generator=g, discriminator=d,
                         gen_optimizer=SimpleNamespace(iterations=7),
                         dis_optimizer=SimpleNamespace(iterations=35))


def test_cli_flags_match_reference_defaults():
  p = cli.build_parser().parse_args([])
  ref = dict(input_dir='dataset/tfrecords', output_dir='runs', batch_size=64,
             num_units=32, kernel_size=24, strides=2, m=2, n=2, epochs=20,
             dropout=0.2, learning_rate=0.0001, noise_dim=32,
             gradient_penalty=10.0, model='wavegan', activation='leakyrelu',
             batch_norm=False, layer_norm=False, algorithm='wgan-gp',
             n_critic=5, clear_output_dir=False, save_generated='',
             plot_weights=False, skip_checkpoints=False, mixed_precision=False,
             profile=False, dpi=120, verbose=1)  # main.py:228-262
  assert vars(p) == ref
  p = cli.build_parser().parse_args(
      '--batch_size 128 --model calciumgan --algorithm wgan-gp --noise_dim 32 '
      '--num_units 64 --kernel_size 24 --strides 2 --m 10 --layer_norm '
      '--mixed_precision --save_generated all'.split())  # README.md:92
  assert p.num_units == 64 and p.layer_norm and p.m == 10


def test_checkpoint_layout_and_resume(tmp_path):
  hp = SimpleNamespace(output_dir=str(tmp_path), verbose=0,
                       ckpt_dir=str(tmp_path / 'checkpoints'))
  gan = _fake_gan()
  utils.save_models(hp, gan, 3)
  utils.save_models(hp, gan, 10)
  path = tmp_path / 'checkpoints' / 'epoch-010.pkl'
  ck = pickle.load(open(path, 'rb'))
  assert sorted(ck) == ['dis_steps', 'dis_weights', 'epoch', 'gen_steps',
                        'gen_weights']  # utils.py:121-128
  assert ck['epoch'] == 10 and ck['gen_steps'] == 7 and ck['dis_steps'] == 35
  assert ck['dis_weights'][0].shape == (24, 2, 5)
  other = _fake_gan()
  other.generator.set_weights([w * 0 for w in other.generator.get_weights()])
  other.gen_optimizer.iterations = 0
  hp2 = SimpleNamespace(output_dir=str(tmp_path), verbose=0)
  utils.load_models(hp2, other)
  assert hp2.start_epoch == 11  # lexicographically last epoch-*, +1
  np.testing.assert_array_equal(other.generator.get_weights()[0],
                                gan.generator.get_weights()[0])
  assert other.gen_optimizer.iterations == 7
  empty = SimpleNamespace(output_dir=str(tmp_path / 'none'), verbose=0)
  utils.load_models(empty, other)
  assert empty.start_epoch == 0


def test_hparams_json_roundtrip(tmp_path):
  hp = SimpleNamespace(output_dir=str(tmp_path), signal_shape=(2048, 102),
                       signals_min=np.float32(-1.5), global_step=3,
                       focus_neurons=[1, 2])
  utils.save_hparams(hp)  # must not crash outside a git checkout
  d = json.load(open(tmp_path / 'hparams.json'))
  assert d['signal_shape'] == [2048, 102] and 'git_hash' in d
  hp2 = SimpleNamespace(output_dir=str(tmp_path))
  utils.load_hparams(hp2)
  assert hp2.global_step == 3 and hp2.output_dir == str(tmp_path)


def test_generated_samples_store(tmp_path):
  gen_dir = tmp_path / 'generated'
  os.makedirs(gen_dir)
  hp = SimpleNamespace(generated_dir=str(gen_dir), normalize=True,
                       signals_min=-1.0, signals_max=3.0, global_step=12)
  a = np.full((2, 8, 3), 0.5, np.float32)
  utils.save_fake_signals(hp, 0, a)
  utils.save_fake_signals(hp, 0, a * 0)  # second validation batch appends
  fn = str(gen_dir / 'epoch000_signals.h5')
  assert h5_helper.contains(fn, 'signals')
  assert h5_helper.get_dataset_length(fn, 'signals') == 4
  out = h5_helper.get(fn, 'signals')
  assert out.shape == (4, 8, 3) and out.dtype == np.float32
  np.testing.assert_allclose(out[0], 1.0)  # denormalised: .5*4-1
  np.testing.assert_allclose(out[3], -1.0)
  assert h5_helper.get(fn, 'signals', neuron=1).shape == (4, 8)
  assert h5_helper.get(fn, 'signals', trial=2).shape == (8, 3)
  info = pickle.load(open(gen_dir / 'info.pkl', 'rb'))
  assert info == {0: {'global_step': 12, 'filename': fn}}  # utils.py:104-113
  h5_helper.overwrite(fn, 'signals', out[:1])
  assert h5_helper.get_dataset_length(fn, 'signals') == 1


def test_summary_tags(tmp_path):
  hp = SimpleNamespace(output_dir=str(tmp_path))
  s = Summary(hp)
  s.log(1.0, 2.0, 3.0, elapse=4.0, step=5, training=True)
  s.log(1.0, 2.0, None, metrics={'signals_metrics/min': 0.1}, step=5,
        training=False)
  tr = [json.loads(l) for l in open(tmp_path / 'scalars.jsonl')]
  va = [json.loads(l) for l in open(tmp_path / 'validation' / 'scalars.jsonl')]
  assert [r['tag'] for r in tr] == ['loss/generator', 'loss/discriminator',
                                    'loss/gradient_penalty', 'elapse']
  assert [r['tag'] for r in va] == ['loss/generator', 'loss/discriminator',
                                    'signals_metrics/min']
  assert all(r['step'] == 5 for r in tr + va)


def test_dataset_directory_roundtrip(tmp_path):
  d = dg.make_dataset(num_neurons=8, sequence_length=64, num_segments=20)
  info = {k: v for k, v in d['info'].items() if k != 'rates_hz'}
  dataset_helper.write_dataset(str(tmp_path / 'ds'), d['signals'], d['spikes'],
                               info, validation_size=6)
  hp = SimpleNamespace(input_dir=str(tmp_path / 'ds'),
                       output_dir=str(tmp_path / 'run'), batch_size=4,
                       noise_dim=32, save_generated='all')
  train_ds, val_ds = dataset_helper.get_dataset(hp)
  assert hp.signal_shape == (64, 8) and hp.train_size == 14
  assert hp.validation_size == 6 and hp.train_steps == 4
  assert hp.noise_shape == (32,) and hp.normalize and hp.num_channels == 8
  batches = list(train_ds)
  assert len(batches) == 4 and batches[-1][0].shape == (2, 64, 8)  # short tail
  assert batches[0][0].dtype == np.float32
  assert sum(len(b[0]) for b in val_ds) == 6
  # the spike half of a batch is gathered lazily (the train loop never reads
  # it) and equals the rows of the signal half
  order = np.concatenate([np.asarray(b[1]) for b in batches])
  sig_order = np.concatenate([b[0] for b in batches])
  assert batches[0][1].shape == (4,) + tuple(train_ds.spikes.shape[1:])
  rows = [int(np.argmax((train_ds.signals == s).all(axis=(1, 2)))) for s in sig_order]
  np.testing.assert_array_equal(order, train_ds.spikes[rows])
  # validation cache for the spike-metric scripts (dataset_helper.py:12-30)
  assert h5_helper.get(hp.validation_cache, 'spikes').dtype == np.int8
  assert len(h5_helper.get(hp.validation_cache, 'signals')) == 6


def test_loader_gathers_batches_into_the_bound_buffer():
  """ArrayDataset.gather_into (main.py binds it to WGAN_GP.batch_buffer): every
  batch of a device-resident set is gathered into the tensor the callable hands
  out for its length -- the ragged last batch gets its own -- and holds the
  same rows as the unbound loader's batches."""
  import torch
  rng = np.random.RandomState(3)
  sig = rng.rand(10, 6, 2).astype(np.float32)
  spk = rng.randint(0, 2, size=(10, 6, 2)).astype(np.int8)
  plain = dataset_helper.ArrayDataset(sig, spk, 4, shuffle=True, seed=9).to_device('cpu')
  bound = dataset_helper.ArrayDataset(sig, spk, 4, shuffle=True, seed=9).to_device('cpu')
  buffers = {}

  def buffer(n):
    if n not in buffers:
      buffers[n] = torch.empty((n, 6, 2), dtype=torch.float32)
    return buffers[n]

  bound.gather_into = buffer
  seen = 0
  for (a, sa), (b, sb) in zip(plain, bound):
    assert b is buffers[len(b)]
    assert torch.equal(a, b)
    np.testing.assert_array_equal(np.asarray(sa), np.asarray(sb))
    seen += len(b)
  assert seen == 10 and sorted(buffers) == [2, 4]


def test_validation_cache_holds_the_samples_a_sharded_run_validates():
  """Data parallel: shard_batch drops the ragged tail of every batch, and the
  cache must hold exactly the samples that get a generated counterpart."""
  from calciumgan_amd import parallel
  for n, bs, world in [(6, 4, 1), (6, 4, 2), (11, 4, 3), (5, 8, 4), (3, 8, 4)]:
    keep = dataset_helper.validated_samples(n, bs, world)
    want = []
    for s in range(0, n, bs):
      batch = np.arange(s, min(s + bs, n))
      shards = [parallel.shard_batch(batch, r, world) for r in range(world)]
      if shards[0] is None:
        assert all(sh is None for sh in shards)
        continue
      # what gather_batch puts back together: sample i * world + r
      want.extend(np.stack(shards, axis=1).reshape(-1))
    np.testing.assert_array_equal(keep, np.asarray(want, dtype=keep.dtype))


# ---------------------------------------------------------------------------
# TFRecord layout of the reference's datasets (generate_tfrecords.py:128-153,
# dataset_helper.py:147-182) without TensorFlow
# ---------------------------------------------------------------------------
def test_crc32c_known_answers_and_mask():
  import struct
  from calciumgan_amd.gan.utils import tfrecord as T
  # RFC 3720 B.4 test vectors
  assert T.crc32c(b'123456789') == 0xe3069283
  assert T.crc32c(bytes(32)) == 0x8a9136aa
  assert T.crc32c(bytes([0xff] * 32)) == 0x62a8ab43
  assert T.crc32c(bytes(range(32))) == 0x46dd794e
  assert T.crc32c(b'') == 0
  # unaligned starts / tails take the byte-wise path
  data = bytes(range(256)) * 5
  whole = T.crc32c(data)
  lib = T._host_lib()
  for cut in (1, 3, 7, 8, 13, 1000):
    c = lib.cg_crc32c(0, data[:cut], cut)
    assert lib.cg_crc32c(c, data[cut:], len(data) - cut) == whole
  # TFRecord mask: rotate right by 15, add the delta, mod 2^32
  c = T.crc32c(struct.pack('<Q', 15))
  assert T.masked_crc32c(struct.pack('<Q', 15)) == (
      (((c >> 15) | (c << 17)) + 0xa282ead8) & 0xffffffff)


def test_example_wire_format_hand_assembled():
  from calciumgan_amd.gan.utils import tfrecord as T
  # Example{features{feature{key:"a" value{bytes_list{value:"xy"}}}}}, byte by
  # byte from the protobuf encoding rules (tag = field << 3 | 2, then length)
  expect = bytes([
      0x0a, 0x0d,              # Example.features, 13 bytes
      0x0a, 0x0b,              # Features.feature map entry, 11 bytes
      0x0a, 0x01, ord('a'),    # entry.key
      0x12, 0x06,              # entry.value (Feature), 6 bytes
      0x0a, 0x04,              # Feature.bytes_list, 4 bytes
      0x0a, 0x02, ord('x'), ord('y'),  # BytesList.value
  ])
  assert T.serialize_example({'a': b'xy'}) == expect
  assert T.parse_example(expect) == {'a': b'xy'}
  # unknown fields / other feature kinds are skipped, order is free
  other = T._len_field(2, T._len_field(1, b'\x00\x00\x80\x3f'))  # float_list
  entry_f = T._len_field(1, b'f') + T._len_field(2, other)
  entry_s = T._len_field(1, b'signal') + T._len_field(
      2, T._len_field(1, T._len_field(1, b'\x01\x02')))
  ex = T._len_field(1, T._len_field(1, entry_f) + T._len_field(1, entry_s))
  assert T.parse_example(ex) == {'signal': b'\x01\x02'}
  # a payload longer than 127 bytes needs a two-byte varint length
  big = T.serialize_example({'signal': bytes(300)})
  assert T.parse_example(big)['signal'] == bytes(300)


def test_tfrecord_directory_reads_like_the_array_directory(tmp_path):
  import pytest
  from calciumgan_amd.gan.utils import tfrecord as T
  rng = np.random.RandomState(3)
  L, C, n_train, n_val = 64, 6, 11, 4
  sig = rng.rand(n_train + n_val, L, C).astype(np.float32)
  spk = (rng.rand(n_train + n_val, L, C) < 0.1).astype(np.float32)
  d = str(tmp_path / 'records')
  os.makedirs(d)
  # two train shards + one validation shard, named as the reference names them
  T.write_segments(T.record_filename(d, 'train', 0, 2), sig[:6], spk[:6])
  T.write_segments(T.record_filename(d, 'train', 1, 2), sig[6:n_train],
                   spk[6:n_train])
  T.write_segments(T.record_filename(d, 'validation', 0, 1), sig[n_train:],
                   spk[n_train:])
  assert sorted(os.listdir(d)) == [
      'train-001-of-002.record', 'train-002-of-002.record',
      'validation-001-of-001.record'
  ]
  info = dict(train_size=n_train, validation_size=n_val, signal_shape=(L, C),
              spike_shape=(L, C), sequence_length=L, num_neurons=C,
              num_channels=C, num_train_shards=2, num_validation_shards=1,
              buffer_size=n_train, normalize=True, stride=2, fft=False,
              conv2d=False, signals_min=0.0, signals_max=1.0)
  with open(os.path.join(d, 'info.pkl'), 'wb') as f:
    pickle.dump(info, f)
  hp = SimpleNamespace(input_dir=d, output_dir=str(tmp_path / 'out'),
                       batch_size=4, noise_dim=32, save_generated='')
  train_ds, val_ds = dataset_helper.get_dataset(hp)
  np.testing.assert_array_equal(train_ds.signals, sig[:n_train])
  np.testing.assert_array_equal(train_ds.spikes, spk[:n_train])
  np.testing.assert_array_equal(val_ds.signals, sig[n_train:])
  assert hp.train_steps == 3 and hp.validation_steps == 1
  assert hp.signal_shape == (L, C)
  batches = [b for b, _ in val_ds]
  assert batches[0].shape == (4, L, C) and batches[0].dtype == np.float32
  # record framing: one flipped payload byte is a data-loss error, a short
  # file a truncation error; verify=False reads past a bad checksum
  path = T.record_filename(d, 'validation', 0, 1)
  raw = bytearray(open(path, 'rb').read())
  raw[40] ^= 0x01
  bad = str(tmp_path / 'bad.record')
  open(bad, 'wb').write(raw)
  with pytest.raises(IOError):
    list(T.read_records(bad))
  assert len(list(T.read_records(bad, verify=False))) == n_val
  open(bad, 'wb').write(bytes(raw[:-3]))
  with pytest.raises(IOError):
    list(T.read_records(bad, verify=False))
  # an empty file holds no records
  open(bad, 'wb').close()
  assert list(T.read_records(bad)) == []


def test_write_dataset_as_tfrecord_shards(tmp_path):
  d = dg.make_dataset(num_neurons=8, sequence_length=64, num_segments=20)
  info = {k: v for k, v in d['info'].items() if k != 'rates_hz'}
  a = dataset_helper.write_dataset(str(tmp_path / 'npy'), d['signals'],
                                   d['spikes'], info, validation_size=6)
  b = dataset_helper.write_dataset(str(tmp_path / 'rec'), d['signals'],
                                   d['spikes'], info, validation_size=6,
                                   tfrecords=True, num_per_shard=5)
  assert b['num_train_shards'] == 3 and b['num_validation_shards'] == 2
  assert a['train_size'] == b['train_size'] == 14
  out = []
  for sub in ('npy', 'rec'):
    hp = SimpleNamespace(input_dir=str(tmp_path / sub),
                         output_dir=str(tmp_path / 'run'), batch_size=4,
                         noise_dim=32, save_generated='')
    out.append(dataset_helper.get_dataset(hp))
  for k in (0, 1):  # same segments in the same order from both layouts
    np.testing.assert_array_equal(out[0][k].signals, out[1][k].signals)
    np.testing.assert_array_equal(out[0][k].spikes.astype(np.float32),
                                  out[1][k].spikes)


def test_tensorboard_event_files(tmp_path):
  """Summary.scalar writes TensorBoard event files (TFRecord framing + Event /
  Summary protobufs, tb_events.py) next to the JSON lines: first record the
  file version, then one scalar event per call with the reference's tags and
  steps; every record's CRCs verify."""
  from calciumgan_amd.gan.utils import tb_events
  hp = SimpleNamespace(output_dir=str(tmp_path / 'run'), verbose=0)
  os.makedirs(hp.output_dir)
  s = Summary(hp)
  s.log(1.5, -2.25, 0.125, metrics={'signals_metrics/min': 3.0}, elapse=7.0,
        step=4, training=True)
  s.scalar('loss/generator', 9.0, step=2, training=False)
  tr = glob.glob(os.path.join(hp.output_dir, 'events.out.tfevents.*'))
  va = glob.glob(os.path.join(hp.output_dir, 'validation',
                              'events.out.tfevents.*'))
  assert len(tr) == 1 and len(va) == 1
  ev = tb_events.read_events(tr[0])
  assert ev[0]['file_version'] == b'brain.Event:2'
  got = [(e['tag'], e['value'], e['step']) for e in ev[1:]]
  assert got == [('loss/generator', 1.5, 4), ('loss/discriminator', -2.25, 4),
                 ('loss/gradient_penalty', 0.125, 4),
                 ('signals_metrics/min', 3.0, 4), ('elapse', 7.0, 4)]
  assert all(e['wall_time'] > 1.6e9 for e in ev)
  ev = tb_events.read_events(va[0])
  assert (ev[1]['tag'], ev[1]['value'], ev[1]['step']) == ('loss/generator',
                                                           9.0, 2)
  # hand-assembled known answer of the wire format: Event{wall_time=1.0,
  # step=3, summary{value{tag="a", simple_value=0.5}}}
  want = (b'\x09' + struct.pack('<d', 1.0) + b'\x10\x03' +
          b'\x2a\x0a' + b'\x0a\x08' + b'\x0a\x01a' + b'\x15' +
          struct.pack('<f', 0.5))
  assert tb_events.encode_event(1.0, 3, 'a', 0.5) == want
  # a flipped payload byte is caught by the record CRC
  raw = bytearray(open(tr[0], 'rb').read())
  raw[-6] ^= 0xff
  bad = tmp_path / 'bad.tfevents'
  bad.write_bytes(bytes(raw))
  with pytest.raises(IOError):
    tb_events.read_events(str(bad))
