#!/usr/bin/env python
"""Training CLI with the reference's flag surface and epoch loop (main.py:33-265
of bryanlimy/CalciumGAN), driving the MI355X hot path.

  python main.py --input_dir dataset/dg_sl2048 --output_dir runs/001 \
      --epochs 400 --batch_size 128 --model calciumgan --algorithm wgan-gp \
      --noise_dim 32 --num_units 64 --kernel_size 24 --strides 2 --m 10 \
      --layer_norm --mixed_precision
  torchrun --nproc-per-node 8 main.py ...        (data parallel over RCCL)

Differences, all documented: without --mixed_precision activations are bf16
(f32 accumulation, f32 master weights) where the reference computes in f32;
with it they are fp16 with the reference's dynamic loss scaling
(mixed_float16).  Scalars go to TensorBoard event files (written without
TensorFlow) and JSON lines; trace plots / spike deconvolution in the loop
(main.py:142-154) are out of scope; under torchrun every rank trains on its
own shard of each batch and rank 0 writes the files.
"""
import argparse
import os
from shutil import rmtree
from time import time

import numpy as np

np.random.seed(1234)  # main.py:11


def train(hparams, train_ds, gan, summary, epoch):
  """main.py:33-75."""
  gen_losses, dis_losses, gradient_penalties = [], [], []
  batch_count = 0
  start = time()
  for signal, _ in train_ds:
    signal = _shard(hparams, signal)
    if signal is None:  # ragged last batch smaller than the world size
      continue
    # --profile: batches 2..6 of the second epoch (main.py:45-52)
    profiling = hparams.profile and epoch == 1 and summary is not None
    if profiling and batch_count == 2:
      summary.profiler_trace(gan)
    gen_loss, dis_loss, gradient_penalty, metrics = gan.train(signal)
    if profiling and batch_count == 6:
      summary.profiler_export()
    batch_count += 1
    gen_losses.append(gen_loss)
    dis_losses.append(dis_loss)
    if gradient_penalty is not None:
      gradient_penalties.append(gradient_penalty)
    hparams.global_step += 1
  if summary is not None and getattr(summary, '_profile', None) is not None:
    # an epoch of 3 .. 6 batches never reaches batch 6: close the window here
    # (graph replay comes back, the launch profiler is switched off)
    summary.profiler_export()
  gen_loss = float(np.mean([float(v) for v in gen_losses]))
  dis_loss = float(np.mean([float(v) for v in dis_losses]))
  gp = float(np.mean([float(v) for v in gradient_penalties
                     ])) if gradient_penalties else None
  end = time()
  if summary is not None:
    summary.log(gen_loss, dis_loss, gp, elapse=end - start, gan=gan, step=epoch,
                training=True)
    summary.scalar('samples_per_sec', hparams.train_size / (end - start),
                   step=epoch, training=True)
  return gen_loss, dis_loss


def validate(hparams, validation_ds, gan, summary, epoch):
  """main.py:78-122."""
  from calciumgan_amd import parallel
  from calciumgan_amd.gan.utils import utils
  gen_losses, dis_losses, gradient_penalties, results = [], [], [], {}
  save_generated = (hparams.save_generated == 'all' and
                    (epoch % 10 == 0 or epoch == hparams.epochs - 1)) or (
                        hparams.save_generated == 'last' and
                        epoch == hparams.epochs - 1)
  start = time()
  for signal, _ in validation_ds:
    signal = _shard(hparams, signal)
    if signal is None:
      continue
    fake, gen_loss, dis_loss, gradient_penalty, metrics = gan.validate(signal)
    gen_losses.append(float(gen_loss))
    dis_losses.append(float(dis_loss))
    if gradient_penalty is not None:
      gradient_penalties.append(float(gradient_penalty))
    for key, item in metrics.items():
      results.setdefault(key, []).append(float(item))
    if save_generated:
      # every rank generated its shard: the whole batch goes to rank 0's file
      fake = parallel.gather_batch(fake)
      if hparams.rank == 0:
        utils.save_fake_signals(hparams, epoch, signals=fake)
  gen_loss, dis_loss = float(np.mean(gen_losses)), float(np.mean(dis_losses))
  results = {key: float(np.mean(item)) for key, item in results.items()}
  end = time()
  if summary is not None:
    summary.log(gen_loss, dis_loss,
                float(np.mean(gradient_penalties)) if gradient_penalties else
                None, metrics=results, elapse=end - start, step=epoch,
                training=False)
  return gen_loss, dis_loss


def train_and_validate(hparams, train_ds, validation_ds, gan, summary):
  """main.py:125-165."""
  from calciumgan_amd.gan.utils import utils
  for epoch in range(hparams.start_epoch, hparams.epochs):
    if hparams.verbose and hparams.rank == 0:
      print('Epoch {:03d}/{:03d}'.format(epoch, hparams.epochs))
    start = time()
    train_gen_loss, train_dis_loss = train(hparams, train_ds, gan, summary,
                                           epoch)
    val_gen_loss, val_dis_loss = validate(hparams, validation_ds, gan, summary,
                                          epoch)
    if epoch % 10 == 0 or epoch == hparams.epochs - 1:
      if not hparams.skip_checkpoints and hparams.rank == 0:
        utils.save_models(hparams, gan, epoch)
    end = time()
    if hparams.verbose and hparams.rank == 0:
      print('Train: generator loss {:.04f} discriminator loss {:.04f}\n'
            'Eval: generator loss {:.04f} discriminator loss {:.04f}\n'
            'Elapse: {:.02f} mins\n'.format(train_gen_loss, train_dis_loss,
                                            val_gen_loss, val_dis_loss,
                                            (end - start) / 60))


def test(validation_ds, gan, hparams):
  """main.py:168-181."""
  results = {}
  for signal, _ in validation_ds:
    signal = _shard(hparams, signal)
    if signal is None:
      continue
    _, _, _, _, metrics = gan.validate(signal)
    for key, item in metrics.items():
      results.setdefault(key, []).append(float(item))
  return {key: float(np.mean(item)) for key, item in results.items()}


def _shard(hparams, batch):
  """Data parallel: rank r takes samples r::world of every batch, equal
  shares on every rank (parallel.shard_batch: up to world - 1 samples of a
  ragged last batch are dropped, None = skip the batch on every rank).  The
  losses / metrics train() and validate() return are already means over the
  ranks."""
  if hparams.world_size > 1:
    from calciumgan_amd import parallel
    return parallel.shard_batch(batch, hparams.rank, hparams.world_size)
  return batch


def main(hparams, return_metrics=False):
  """main.py:184-224."""
  from calciumgan_amd import parallel
  from calciumgan_amd.gan.algorithms.registry import get_algorithm
  from calciumgan_amd.gan.models.registry import get_models
  import calciumgan_amd.gan.algorithms  # noqa: F401  (registers algorithms)
  import calciumgan_amd.gan.models  # noqa: F401  (registers models)
  from calciumgan_amd.gan.utils import utils
  from calciumgan_amd.gan.utils.dataset_helper import get_dataset
  from calciumgan_amd.gan.utils.summary_helper import Summary

  parallel.init_process_group()
  hparams.rank, hparams.world_size = parallel.rank(), parallel.world_size()

  if hparams.rank == 0:
    if hparams.clear_output_dir and os.path.exists(hparams.output_dir):
      rmtree(hparams.output_dir)
    os.makedirs(hparams.output_dir, exist_ok=True)
  # the other ranks touch the output directory only after rank 0 has set it up
  parallel.barrier()

  hparams.focus_neurons = [87, 58, 90, 39, 7, 60, 14, 5, 13]
  summary = Summary(hparams) if hparams.rank == 0 else None
  train_ds, validation_ds = get_dataset(hparams, summary)
  parallel.barrier()  # rank 0 wrote generated/validation.h5
  generator, discriminator = get_models(hparams, summary)
  if hparams.rank == 0:
    utils.save_hparams(hparams)
  gan = get_algorithm(hparams, generator, discriminator, summary)
  utils.load_models(hparams, gan)
  # dataset resident in HBM: no per-step host-to-device copy
  train_ds.to_device(gan.device)
  if hparams.world_size == 1 and hasattr(gan, 'batch_buffer'):
    # batches are gathered straight into the buffer train()'s hipGraph reads
    train_ds.gather_into = gan.batch_buffer
  validation_ds.to_device(gan.device)

  start = time()
  train_and_validate(hparams, train_ds, validation_ds, gan, summary)
  end = time()
  if summary is not None:
    summary.scalar('elapse/total', end - start)

  if hparams.surrogate_ds and hparams.rank == 0:
    utils.generate_dataset(hparams, gan=gan, num_samples=2 * 10**6)
  if return_metrics:
    return test(validation_ds, gan, hparams)


def build_parser():
  """main.py:227-262 -- same flags and defaults (the reference's default
  --model 'wavegan' is not registered there either; use --model calciumgan)."""
  parser = argparse.ArgumentParser()
  parser.add_argument('--input_dir', default='dataset/tfrecords')
  parser.add_argument('--output_dir', default='runs')
  parser.add_argument('--batch_size', default=64, type=int)
  parser.add_argument('--num_units', default=32, type=int)
  parser.add_argument('--kernel_size', default=24, type=int)
  parser.add_argument('--strides', default=2, type=int)
  parser.add_argument('--m', default=2, type=int, help='phase shuffle m')
  parser.add_argument('--n', default=2, type=int, help='phase shuffle n')
  parser.add_argument('--epochs', default=20, type=int)
  parser.add_argument('--dropout', default=0.2, type=float)
  parser.add_argument('--learning_rate', default=0.0001, type=float)
  parser.add_argument('--noise_dim', default=32, type=int)
  parser.add_argument('--gradient_penalty', default=10.0, type=float)
  parser.add_argument('--model', default='wavegan', type=str)
  parser.add_argument('--activation', default='leakyrelu', type=str)
  parser.add_argument('--batch_norm', action='store_true')
  parser.add_argument('--layer_norm', action='store_true')
  parser.add_argument('--algorithm', default='wgan-gp', type=str)
  parser.add_argument('--n_critic', default=5, type=int,
                      help='number of steps between each generator update')
  parser.add_argument('--clear_output_dir', action='store_true')
  parser.add_argument('--save_generated', default='',
                      choices=['', 'last', 'all'], type=str)
  parser.add_argument('--plot_weights', action='store_true')
  parser.add_argument('--skip_checkpoints', action='store_true')
  parser.add_argument('--mixed_precision', action='store_true')
  parser.add_argument('--profile', action='store_true',
                      help='time every MFMA-kernel launch of batches 2-6 of the '
                      'second epoch -> <output_dir>/profiler/mfma_kernels.json')
  parser.add_argument('--dpi', default=120, type=int)
  parser.add_argument('--verbose', default=1, type=int)
  return parser


if __name__ == '__main__':
  params = build_parser().parse_args()
  params.global_step = 0
  params.surrogate_ds = True if 'surrogate' in params.input_dir else False
  main(params)
