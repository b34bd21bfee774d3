"""The host loop of main.py on the GPU (reference main.py:125-224): two tiny
epochs of train -> validate -> generated samples -> checkpoint, the --profile
window, a resume from the checkpoint, and the same under --mixed_precision."""
import glob
import json
import os
import pickle

import numpy as np
import pytest

import main as cli
from calciumgan_amd.data import dg
from calciumgan_amd.gan.utils import dataset_helper, h5_helper, tb_events

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _back_to_bf16():
  yield
  from calciumgan_amd import _lib
  _lib.use('bf16')


def _dataset(tmp_path, n=70, L=256, C=16):
  d = dg.make_dataset(num_neurons=C, sequence_length=L, num_segments=n)
  info = {k: v for k, v in d['info'].items() if k != 'rates_hz'}
  path = str(tmp_path / 'ds')
  dataset_helper.write_dataset(path, d['signals'], d['spikes'], info,
                               validation_size=6)
  return path


def _args(input_dir, output_dir, *extra):
  a = cli.build_parser().parse_args([
      '--input_dir', input_dir, '--output_dir', output_dir, '--model',
      'calciumgan', '--algorithm', 'wgan-gp', '--batch_size', '8', '--num_units',
      '8', '--m', '2', '--layer_norm', '--epochs', '2', '--save_generated',
      'last', '--verbose', '0'] + list(extra))
  a.global_step = 0
  a.surrogate_ds = False
  return a


def _scalars(path):
  return [json.loads(l) for l in open(path)]


def test_main_trains_validates_saves_and_resumes(tmp_path):
  ds = _dataset(tmp_path)
  out = str(tmp_path / 'run')
  hp = _args(ds, out, '--profile')
  metrics = cli.main(hp, return_metrics=True)
  assert set(metrics) == {'signals_metrics/min', 'signals_metrics/max',
                          'signals_metrics/mean', 'signals_metrics/std'}
  assert all(np.isfinite(v) for v in metrics.values()), metrics
  # 64 training segments / batch 8 = 8 train() per epoch (each kept as a device
  # scalar until the end of the epoch: the graph-replay aliasing regression)
  assert hp.global_step == 16
  tr = _scalars(os.path.join(out, 'scalars.jsonl'))
  tags = [r['tag'] for r in tr if r['step'] == 1]
  assert {'loss/generator', 'loss/discriminator', 'loss/gradient_penalty',
          'elapse', 'samples_per_sec'} <= set(tags)
  assert all(np.isfinite(r['value']) for r in tr)
  va = _scalars(os.path.join(out, 'validation', 'scalars.jsonl'))
  assert {'loss/generator', 'signals_metrics/std'} <= {r['tag'] for r in va}
  # TensorBoard event files beside them
  ev = glob.glob(os.path.join(out, 'events.out.tfevents.*'))
  assert len(ev) == 1
  assert len(tb_events.read_events(ev[0])) == len(tr) + 1
  # checkpoints of epochs 0 and 1, Keras weight order
  ck = sorted(glob.glob(os.path.join(out, 'checkpoints', 'epoch-*.pkl')))
  assert [os.path.basename(c) for c in ck] == ['epoch-000.pkl', 'epoch-001.pkl']
  c1 = pickle.load(open(ck[1], 'rb'))
  assert c1['epoch'] == 1 and len(c1['gen_weights']) == 24
  assert len(c1['dis_weights']) == 12
  assert int(c1['dis_steps']) == 80 and int(c1['gen_steps']) == 16
  # generated samples of the last epoch: the 6 validation segments, denormalised
  gen = h5_helper.get(os.path.join(out, 'generated', 'epoch001_signals.h5'),
                      'signals')
  assert gen.shape == (6, 256, 16) and np.isfinite(gen).all()
  val = h5_helper.get(os.path.join(out, 'generated', 'validation.h5'), 'signals')
  assert val.shape == (6, 256, 16)
  # --profile: batches 2..6 of the second epoch, every MFMA launch timed
  prof = json.load(open(os.path.join(out, 'profiler', 'mfma_kernels.json')))
  fam = prof['families']
  assert fam['cg_swconv']['launches'] > 100 and fam['cg_wgrad']['launches'] > 20
  assert fam['cg_swconv']['mean_us'] > 0
  # resume: two more epochs continue from epoch 2 with the optimizers' counts
  hp2 = _args(ds, out, '--epochs', '3')
  hp2.epochs = 3
  cli.main(hp2)
  ck = sorted(glob.glob(os.path.join(out, 'checkpoints', 'epoch-*.pkl')))
  assert os.path.basename(ck[-1]) == 'epoch-002.pkl'
  c2 = pickle.load(open(ck[-1], 'rb'))
  assert int(c2['dis_steps']) == 120 and int(c2['gen_steps']) == 24
  moved = sum(float(np.abs(a - b).sum())
              for a, b in zip(c1['gen_weights'], c2['gen_weights']))
  assert moved > 0


def test_main_mixed_precision_runs_fp16_with_loss_scaling(tmp_path):
  ds = _dataset(tmp_path)
  out = str(tmp_path / 'run16')
  hp = _args(ds, out, '--mixed_precision', '--epochs', '1')
  hp.epochs = 1
  cli.main(hp)
  tr = _scalars(os.path.join(out, 'scalars.jsonl'))
  scale = [r['value'] for r in tr if r['tag'] == 'model/loss_scale']
  assert scale and 1.0 <= scale[-1] <= 2.0**15
  assert all(np.isfinite(r['value']) for r in tr)
  ck = pickle.load(open(os.path.join(out, 'checkpoints', 'epoch-000.pkl'), 'rb'))
  # applied steps: at most 8 train() x (5 + 1) updates, fewer if the scaler
  # had to skip overflowing ones on its way down from 2**15
  assert 0 < int(ck['dis_steps']) <= 40 and 0 < int(ck['gen_steps']) <= 8


def test_recorded_data_pipeline_tfrecords_to_spike_metrics(tmp_path):
  """BASELINE configs[3]'s pipeline end to end (no recorded data exists here:
  DG calcium written in the reference's TFRecord layout stands in for it):
  train-*/validation-*.record shards of tf.train.Example{signal, spike}
  (dataset/generate_tfrecords.py, read by gan/utils/tfrecord.py without
  TensorFlow) -> main.py -> generated/epochNNN_signals.h5 + validation.h5 ->
  compute_metrics.py: OASIS deconvolution written back into the generated file,
  KL of firing rate / correlation / van Rossum distance, recorded vs synthetic."""
  import compute_metrics as cm
  d = dg.make_dataset(num_neurons=16, sequence_length=256, num_segments=70)
  info = {k: v for k, v in d['info'].items() if k != 'rates_hz'}
  ds = str(tmp_path / 'ds_tfr')
  dataset_helper.write_dataset(ds, d['signals'], d['spikes'], info,
                               validation_size=6, tfrecords=True, num_per_shard=40)
  assert len(glob.glob(os.path.join(ds, 'train-*.record'))) == 2
  out = str(tmp_path / 'run_tfr')
  hp = _args(ds, out, '--epochs', '2')
  cli.main(hp)
  gen_file = os.path.join(out, 'generated', 'epoch001_signals.h5')
  assert h5_helper.get(gen_file, 'signals').shape == (6, 256, 16)
  mhp = cm.build_parser().parse_args(['--output_dir', out, '--num_processors',
                                      '1', '--verbose', '0'])
  rep = cm.main(mhp)
  assert list(rep) == [1]                       # the last generated epoch
  r = rep[1]
  spikes = h5_helper.get(gen_file, 'spikes')
  assert spikes.shape == (6, 256, 16) and set(np.unique(spikes)) <= {0, 1}
  for k in ('firing_rate_kl', 'correlation_kl', 'van_rossum_kl'):
    assert np.isfinite(r[k]['mean']) and r[k]['mean'] >= 0, (k, r[k])
  # (the reference draws its plot neurons WITH replacement: np.random.choice)
  assert 1 <= len(r['firing_rate_kl']['neurons']) <= 6
  saved = json.load(open(os.path.join(out, 'spike_metrics.json')))
  assert saved['1']['van_rossum_kl']['mean'] == r['van_rossum_kl']['mean']


def test_recorded_data_pipeline_at_configs3_shapes(tmp_path, capsys):
  """BASELINE configs[3] at ITS shapes: 256 training + 32
  validation segments of sl2048 x 102 neurons as TFRecord shards ->
  main.py --batch_size 128 --num_units 64 --m 10 --layer_norm --epochs 1 (two
  train() calls at the benchmark's geometry, one ragged validation batch, the
  generated set) -> compute_metrics.py over the 32 generated trials: OASIS
  deconvolution of 32 x 102 traces, firing-rate / correlation / van Rossum KLs
  (128 trials are two minutes of host time on the test box: 32 keep the suite
  short, the shapes per trial are the config's).  Recorded
  calcium does not exist here: DG calcium stands in for it.  Reader contract
  dataset_helper.py:147-182, spike_helper.py:23-54, compute_metrics.py:35-57."""
  import time
  import compute_metrics as cm
  L, C, B, V = 2048, 102, 128, 32
  d = dg.make_dataset(num_neurons=C, sequence_length=L, num_segments=2 * B + V)
  info = {k: v for k, v in d['info'].items() if k != 'rates_hz'}
  ds = str(tmp_path / 'ds_cfg4')
  dataset_helper.write_dataset(ds, d['signals'], d['spikes'], info,
                               validation_size=V, tfrecords=True, num_per_shard=128)
  assert len(glob.glob(os.path.join(ds, 'train-*.record'))) == 2
  assert len(glob.glob(os.path.join(ds, 'validation-*.record'))) == 1
  out = str(tmp_path / 'run_cfg4')
  a = cli.build_parser().parse_args([
      '--input_dir', ds, '--output_dir', out, '--model', 'calciumgan',
      '--algorithm', 'wgan-gp', '--batch_size', str(B), '--num_units', '64', '--m',
      '10', '--layer_norm', '--epochs', '1', '--save_generated', 'last',
      '--verbose', '0'])
  a.global_step = 0
  a.surrogate_ds = False
  t0 = time.time()
  cli.main(a)
  t_train = time.time() - t0
  assert a.global_step == 2                      # 256 segments / batch 128
  tr = _scalars(os.path.join(out, 'scalars.jsonl'))
  assert all(np.isfinite(r['value']) for r in tr)
  rate = [r['value'] for r in tr if r['tag'] == 'samples_per_sec']
  assert rate and rate[-1] > 0                   # the throughput line is there
  gen_file = os.path.join(out, 'generated', 'epoch000_signals.h5')
  gen = h5_helper.get(gen_file, 'signals')
  assert gen.shape == (V, L, C) and np.isfinite(gen).all()
  val = h5_helper.get(os.path.join(out, 'generated', 'validation.h5'), 'signals')
  assert val.shape == (V, L, C)
  t0 = time.time()
  # (one process: a Pool would fork this GPU-holding test process)
  mhp = cm.build_parser().parse_args(['--output_dir', out, '--num_processors',
                                      '1', '--verbose', '0'])
  rep = cm.main(mhp)
  t_metrics = time.time() - t0
  r = rep[0]
  spikes = h5_helper.get(gen_file, 'spikes')
  assert spikes.shape == (V, L, C) and spikes.dtype == np.int8
  assert set(np.unique(spikes)) <= {0, 1}
  for k in ('firing_rate_kl', 'correlation_kl', 'van_rossum_kl'):
    assert np.isfinite(r[k]['mean']) and r[k]['mean'] >= 0, (k, r[k])
  assert len(r['van_rossum_heatmap_min']) >= 1
  with capsys.disabled():
    print('\nconfigs[3] at its shapes: main.py (dataset load, 2 train() + validation '
          'at B = 128, generated set) %.1f s; compute_metrics.py over 32 x 102 '
          'traces %.1f s; KL firing rate %.3f, correlation %.3f, van Rossum %.3f '
          '(two training steps: the numbers only show the chain runs)' % (
              t_train, t_metrics, r['firing_rate_kl']['mean'],
              r['correlation_kl']['mean'], r['van_rossum_kl']['mean']))
