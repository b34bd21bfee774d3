#!/usr/bin/env python
"""Headline benchmark: WGAN-GP training samples/sec of the CalciumGAN 1-D conv
stack at BASELINE.json configs[1] (DG sl2048, 102 neurons, batch 128 per GPU,
num_units 64, k 24, s 2, layer_norm, n_critic 5), bf16 MFMA / f32 accumulate.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

One step = one WGAN_GP.train() = 5 critic updates + 1 generator update on one
batch (reference gan/algorithms/wgan_gp.py:82-95).  Inputs are resident in HBM
before the timed region; timing is barrier + synchronize on both sides, MAX
over ranks; rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

MFMA_BF16_PEAK = 2.5e15  # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md


def make_hparams(L, C, U, m, mixed_precision=False):
  from types import SimpleNamespace
  return SimpleNamespace(
      signal_shape=(L, C), sequence_length=L, num_channels=C, num_neurons=C,
      num_units=U, kernel_size=24, strides=2, noise_dim=32, noise_shape=(32,),
      m=m, layer_norm=True, batch_norm=False, normalize=True,
      activation='leakyrelu', gradient_penalty=10.0, n_critic=5,
      learning_rate=1e-4, signals_min=0.0, signals_max=1.0, conv2d=False,
      mixed_precision=bool(mixed_precision), model='calciumgan',
      algorithm='wgan-gp', verbose=0)


def algorithmic_flops(hp, l1_mix=False):
  """FLOPs per training sample of one train(): 8 F_G + 52 F_D - 10 F_D1
  (SURVEY 3.3 / BASELINE.md 2), split by kernel family.  l1_mix (round 5): the
  critic's first layer on x^ comes from the layer's outputs on real and fake
  (cg_lrelu_mix) -- n F_D1 that the step does NOT execute leave the cg_swconv
  family's count (a rate is credited only with work done)."""
  from calciumgan_amd import geometry as geo
  k = hp.kernel_size
  g_layers, d_layers = geo.generator_layers(hp), geo.discriminator_layers(hp)
  w0 = g_layers[0].lin
  macs_g = hp.noise_dim * w0 * hp.noise_dim
  macs_g += sum(l.lin * k * l.cin * l.cout for l in g_layers)
  macs_g += hp.signal_shape[0] * hp.num_channels * hp.num_channels
  macs_d = sum(l.lout * k * l.cin * l.cout for l in d_layers)
  macs_d += d_layers[-1].lout * d_layers[-1].cout
  f_g, f_d = 2.0 * macs_g, 2.0 * macs_d
  f_d1 = 2.0 * d_layers[0].lout * k * d_layers[0].cin * d_layers[0].cout
  n = hp.n_critic
  # the generator's last Dense runs in the streaming cg_dense_rows kernel
  # (HBM-bound, not part of the swconv family) whenever its shape allows
  f_dense = 2.0 * hp.signal_shape[0] * hp.num_channels * hp.num_channels
  cp = geo.pitch(hp.num_channels)
  streaming = geo.dense_streams(cp)
  # ((n + 1) forwards; + the input gradient dh = dz W^T of the generator update,
  # streaming for pitches >= 128)
  dense_fwd_bwd = ((n + 1) + (1 if cp >= 128 else 0)) * f_dense if streaming else 0.0
  # its weight gradient always runs in cg_dense_wgrad (not a cg_wgrad launch)
  dense_rows = dense_fwd_bwd + f_dense
  skipped = n * f_d1 if l1_mix else 0.0
  swconv = (n * (f_g + 7 * f_d - 2 * f_d1) + 2 * f_g + 2 * f_d - dense_fwd_bwd -
            skipped)
  wgrad = n * 3 * f_d + f_g - f_dense
  total = (n + 3) * f_g + (10 * n + 2) * f_d - 2 * n * f_d1 - skipped
  assert abs(total - swconv - wgrad - dense_rows) < 1e-3 * total
  return dict(total=total, swconv=swconv, wgrad=wgrad, f_g=f_g, f_d=f_d,
              dense_rows=dense_rows)


def algorithmic_bytes_swconv(hp, B, batched_g=False, l1_mix=False):
  """Algorithmic HBM bytes of all cg_swconv launches of one train(): every
  launch reads its bf16 source tensor and its packed bf16 weights once and
  writes its output once (bf16; f32 for the generator output).  batched_g:
  the fake batches of all critic updates come from one generator pass (single
  rank) -- same activation bytes, the generator's weights read once instead of
  n_critic times, n_critic - 1 fewer launches per generator layer.  Returns
  (bytes per step, launches per step)."""
  from calciumgan_amd import geometry as geo
  k, n = hp.kernel_size, hp.n_critic
  g_l, d_l = geo.generator_layers(hp), geo.discriminator_layers(hp)
  w0, nd, L, C = g_l[0].lin, hp.noise_dim, hp.signal_shape[0], hp.num_channels
  cp = geo.pitch(C)

  def conv(nb, lay, out_bytes=2):  # D conv fwd / tangent (stride 2)
    return nb * (lay.lin * lay.cinp * 2 + lay.lout * lay.coutp * out_bytes) + \
        k * lay.cinp * lay.cout * 2

  def dgrad(nb, lay, out_bytes=2, masked=False):  # D input gradient (2 phases)
    # masked: the phase-unshuffle + LeakyReLU' mask fused into the epilogue
    # reads the activation it masks with
    return nb * (lay.lout * lay.coutp * 2 + lay.lin * lay.cinp * out_bytes +
                 (lay.lin * lay.cinp * 2 if masked else 0)) + \
        k * lay.coutp * lay.cin * 2

  def convT(nb, lay, keep):  # G conv-transpose fwd (two phases)
    # rows of <= 128 channels: LayerNorm + LeakyReLU fused -- the launch writes
    # the activation, and (keep: a backward follows) also the pre-activation
    # and the row statistics
    out = lay.lout * lay.coutp * 2
    if hp.layer_norm and lay.cout <= 128 and keep:
      out += lay.lout * (lay.coutp * 2 + 8)
    return nb * (lay.lin * lay.cinp * 2 + out) + k * lay.cinp * lay.cout * 2

  def convT_dgrad(nb, lay):
    return nb * (lay.lout * lay.coutp * 2 + lay.lin * lay.cinp * 2) + \
        k * lay.coutp * lay.cin * 2

  streaming = geo.dense_streams(cp)  # last Dense in cg_dense_rows, not a swconv launch
  streaming_dgrad = streaming and cp >= 128  # ... and its input gradient

  def g_forward(keep):
    t = B * (nd * 2 + w0 * nd * 2) + nd * w0 * nd * 2
    t += sum(convT(B, l, keep) for l in g_l)
    if not streaming:
      t += B * L * (cp * 2 + cp * 4) + cp * C * 2
    return t

  g_bwd = 0 if streaming_dgrad else B * L * (cp * 2 + cp * 2) + cp * C * 2
  g_bwd += sum(convT_dgrad(B, l) for l in g_l)
  d_fwd = lambda nb: sum(conv(nb, l) for l in d_l)
  fusable = lambda l: 2 * max(1, hp.m) + 1 <= l.lin
  critic = g_forward(False) + d_fwd(3 * B) + sum(dgrad(3 * B, l, masked=fusable(l))
                                      for l in d_l[1:])
  critic += dgrad(B, d_l[0]) + d_fwd(B)  # x^ input gradient + tangent chain
  # (the tangent chain runs in place over the x^ segment's activations and masks
  # with their sign: every launch also READS the bf16 activation it overwrites --
  # counted since round 5, as tools/traffic_by_geometry.py does per launch)
  critic += sum(B * l.lout * l.coutp * 2 for l in d_l)
  if l1_mix:  # layer 1 forward runs over [real | fake] only
    critic -= B * (d_l[0].lin * d_l[0].cinp * 2 + d_l[0].lout * d_l[0].coutp * 2)
  gen = g_forward(True) + d_fwd(B) + sum(dgrad(B, l, masked=fusable(l))
                               for l in d_l[1:]) + \
      dgrad(B, d_l[0]) + g_bwd
  g_launches = 6 if streaming else 7
  launches = n * (g_launches + 5 + 4 + 1 + 5) + (g_launches + 5 + 4 + 1 + 6)
  if streaming_dgrad:
    launches -= 1
  total = n * critic + gen
  if batched_g and n > 1:
    launches -= (n - 1) * g_launches
    g_weights = nd * w0 * nd * 2 + sum(k * l.cinp * l.cout * 2 for l in g_l)
    if not streaming:
      g_weights += cp * C * 2
    total -= (n - 1) * g_weights
  return total, launches


def cpu_baseline(hp, batch, steps):
  """The oracle (CPU restatement, NOT TensorFlow) timed on this box's host
  cores on a bounded sample of the same workload."""
  import oracle as O
  ohp = O.make_hparams(hp.signal_shape[0], hp.num_channels, hp.num_units,
                       kernel_size=hp.kernel_size, m=hp.m)
  rng = np.random.RandomState(0)
  gan = O.OracleGAN(ohp, O.init_generator(ohp, rng),
                    O.init_discriminator(ohp, rng))
  real = rng.uniform(0, 1, (batch,) + hp.signal_shape).astype(np.float32)
  gan.train(real, O.draw_randomness(ohp, batch, 0))  # warm-up
  t0 = time.time()
  for i in range(steps):
    gan.train(real, O.draw_randomness(ohp, batch, i + 1))
  dt = time.time() - t0
  return dict(
      value=batch * steps / dt, unit='samples/s',
      cores=torch.get_num_threads(), kind='port',
      sample='same shapes (L={}, C={}, U={}), batch {} of the benchmark batch, '
      '1 warm-up + {} timed train() steps of the torch-CPU f32 oracle '
      '(restatement of the reference graph, not TensorFlow); a reported '
      'baseline, not the target: three steps of batch 8 on a shared host swing '
      '0.23-1.10 samples/s between boxes of the pool (+- 2.6 x)'.format(
          hp.signal_shape[0], hp.num_channels, hp.num_units, batch, steps))


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=20)
  ap.add_argument('--warmup', type=int, default=3)
  ap.add_argument('--batch', type=int, default=128, help='per-GPU batch')
  ap.add_argument('--seq_len', type=int, default=2048)
  ap.add_argument('--neurons', type=int, default=102)
  ap.add_argument('--num_units', type=int, default=64)
  ap.add_argument('--m', type=int, default=10)
  ap.add_argument('--mixed_precision', action='store_true',
                  help='fp16 activations + dynamic loss scaling (the '
                  "reference's mixed_float16; BASELINE.json configs[4])")
  ap.add_argument('--no_cpu_baseline', action='store_true')
  ap.add_argument('--cpu_batch', type=int, default=8)
  ap.add_argument('--cpu_steps', type=int, default=3)
  ap.add_argument('--no_kernel_timing', action='store_true')
  ap.add_argument('--backend', default='nccl',
                  help="torch.distributed backend ('nccl' = RCCL; 'gloo' only "
                  'for single-GPU rehearsals of the multi-rank path)')
  args = ap.parse_args()

  from calciumgan_amd import nets, parallel
  world = parallel.env_world()
  if world > 1:
    parallel.init_process_group(args.backend)
  else:
    torch.cuda.set_device(0)
  rank = parallel.rank()
  if world != args.gpus and rank == 0:
    print('warning: --gpus {} but WORLD_SIZE {}'.format(args.gpus, world),
          file=sys.stderr)

  from calciumgan_amd.data import dg
  from calciumgan_amd.gan.algorithms import get_algorithm
  from calciumgan_amd.gan.models import get_models

  hp = make_hparams(args.seq_len, args.neurons, args.num_units, args.m,
                    args.mixed_precision)
  gen, dis = get_models(hp, None)
  gan = get_algorithm(hp, gen, dis, None)

  B = args.batch
  # A resident dataset of a few batches; every step gathers ITS batch into the
  # buffer train()'s hipGraph reads, exactly as main.py's loader does
  # (dataset_helper.ArrayDataset.gather_into): the step's input delivery -- one
  # index_select of B segments inside HBM, ~35 us at cfg2 -- is inside the timed
  # region (with the batch parked in the buffer once, the timed step
  # carried no delivery at all).  cfg5's 4.3 GB batches keep one resident batch.
  nseg = B * (4 if B * args.seq_len * args.neurons * 4 < (1 << 30) else 1)
  data = dg.make_dataset(args.neurons, args.seq_len, num_segments=nseg,
                         seed=1234 + rank)
  dataset = torch.from_numpy(data['signals']).to(gan.device)
  real = gan.batch_buffer(B)
  gsteps = torch.Generator().manual_seed(99 + rank)
  total = args.warmup + args.steps
  index = [torch.randperm(nseg, generator=gsteps)[:B].to(gan.device)
           for _ in range(min(total, 64))]

  def next_batch(i):
    torch.index_select(dataset, 0, index[i % len(index)], out=real)
    return real

  def barrier():
    if world > 1:
      dist.barrier()
    torch.cuda.synchronize()

  for i in range(args.warmup):
    gan.train(next_batch(i))
  barrier()
  t0 = time.perf_counter()
  for i in range(args.steps):
    out = gan.train(next_batch(args.warmup + i))
  barrier()
  dt = time.perf_counter() - t0
  losses = [float(out[0]), float(out[1]), float(out[2])]

  # Kernel-duration leg of the roofline: the SAME K steps again with every
  # MFMA-kernel launch timed by the library's launch profiler
  # (cg_profile_enable: hipExtLaunchKernelGGL event pairs that carry the
  # kernel's own begin / end timestamps on the launch stream, i.e. the figure
  # rocprofv3 --kernel-trace reports).  Single-process runs replay train() as
  # one hipGraph in the timed region above, where launches cannot be
  # instrumented, so this pass runs the identical kernels eagerly right after
  # it (rank 0's numbers are reported).
  records = []
  dt_prof = None
  if not args.no_kernel_timing:
    import ctypes
    from calciumgan_amd import _lib
    lib = _lib.load()
    cap = 512 * args.steps
    graphed = getattr(gan, '_use_graph', False)
    gan._use_graph = False
    _lib.check(lib.cg_profile_enable(cap), 'cg_profile_enable')
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(args.steps):
      gan.train(real)
    torch.cuda.synchronize()
    dt_prof = time.perf_counter() - t1
    ms = (ctypes.c_float * cap)()
    famid = (ctypes.c_int * cap)()
    n = lib.cg_profile_collect(ms, famid, cap)
    if n < 0:
      raise RuntimeError('cg_profile_collect: HIP error {}'.format(-n))
    records = [(('swconv', 'wgrad')[famid[i]], ms[i] * 1e-3) for i in range(n)]
    gan._use_graph = graphed

  t = torch.tensor([dt], dtype=torch.float64, device=gan.device)
  if world > 1:
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
  dt = float(t.item())

  # the workload label follows the arguments: only BASELINE.json configs[1]
  # shapes are called cfg2
  default_workload = (args.seq_len, args.neurons, args.num_units, args.m, B,
                      args.mixed_precision) == (2048, 102, 64, 10, 128, False)
  cfg_name = 'cfg2' if default_workload else (
      'cfg1 shapes' if (args.seq_len, args.neurons) == (256, 16) else
      'cfg5' if (args.seq_len, args.neurons, B, args.mixed_precision) ==
      (8192, 512, 256, True) else
      'cfg5 shapes' if (args.seq_len, args.neurons) == (8192, 512) else
      'cfg2 shapes, mixed_float16' if (args.seq_len, args.neurons, args.num_units,
                                       args.m, B) == (2048, 102, 64, 10, 128) else
      'custom shapes (not a BASELINE.json config)')
  if rank == 0:
    l1_mix = bool(gan._get_state(B)['critic'].mixes_layer1)
    fl = algorithmic_flops(hp, l1_mix=l1_mix)
    value = world * B * args.steps / dt
    roofline = None
    fam = {}
    for name, sec in records:
      d = fam.setdefault(name, [0.0, 0])
      d[0] += sec
      d[1] += 1
    if 'swconv' in fam:
      sec, cnt = fam['swconv']
      flops = fl['swconv'] * B * args.steps
      roofline = dict(
          kernel='swconv_kernel (all MFMA conv / dgrad / dense launches)',
          bound='mfma', achieved=flops / sec / 1e12,
          peak=MFMA_BF16_PEAK / 1e12, unit='TFLOP/s',
          frac=flops / sec / MFMA_BF16_PEAK, traffic=None,
          launches_per_step=cnt / args.steps,
          avg_launch_us=sec / cnt * 1e6,
          flop_per_launch=flops / cnt,
          share_of_step=sec / dt_prof,
          timing='kernel begin/end timestamps of every launch '
          '(hipExtLaunchKernelGGL event pairs), eager pass of the same '
          '{} steps run right after the timed region ({:.2f} ms/step '
          'eager vs {:.2f} ms/step timed)'.format(
              args.steps, dt_prof / args.steps * 1e3, dt / args.steps * 1e3))
      from calciumgan_amd.gan.algorithms import wgan_gp as _w
      ab, alaunch = algorithmic_bytes_swconv(
          hp, B, batched_g=world == 1 and _w._BATCH_G and not _w._FORCE_SPLIT,
          l1_mix=l1_mix)
      roofline['algorithmic_hbm_bytes_per_launch'] = ab / alaunch
      # context, not the roofline's denominator: what dense bf16 MFMA SUSTAINS on
      # random operands under the package power cap (tools/probe/mfma_sustained.hip,
      # profiles/r04_mfma_sustained.txt: 2.04 PFLOP/s at 2.07 GHz; `peak` is the
      # 2.4 GHz figure the chip only holds on zero operands)
      roofline['sustained_mfma_measured'] = dict(
          tflops=2043.0, frac_of_it=flops / sec / 2043.0e12,
          source='profiles/r04_mfma_sustained.txt')
      import glob
      # (measured per workload: the default one and BASELINE configs[4])
      pmcs = sorted(glob.glob(os.path.join(
          ROOT, 'profiles', 'r??_cfg5_pmc_traffic.json'
          if cfg_name == 'cfg5' else 'r??_pmc_traffic.json')))
      if pmcs and (default_workload or cfg_name == 'cfg5'):
        # HBM bytes per launch from rocprofv3 PMC passes of this same command
        # (tools/pmc_traffic.sh: FETCH_SIZE x2 per the gfx950 correction +
        # WRITE_SIZE, separate passes).  A committed measurement, not a live
        # one: the file names the commit its kernels were built from
        pj = json.load(open(pmcs[-1]))
        roofline['traffic'] = pj.get('swconv', {}).get('hbm_bytes_per_launch')
        roofline['traffic_source'] = 'profiles/' + os.path.basename(pmcs[-1])
        roofline['traffic_measured_at_commit'] = pj.get('commit')
      if 'wgrad' in fam:
        wsec, wcnt = fam['wgrad']
        wfl = fl['wgrad'] * B * args.steps
        roofline['wgrad_kernel'] = dict(
            achieved=wfl / wsec / 1e12, frac=wfl / wsec / MFMA_BF16_PEAK,
            launches_per_step=wcnt / args.steps, avg_launch_us=wsec / wcnt * 1e6,
            share_of_step=wsec / dt_prof)
    line = {
        'metric': 'training samples/sec (seq_len={}, n_critic={})'.format(
            args.seq_len, hp.n_critic),
        'value': value,
        'unit': 'samples/s',
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': dt / args.steps * 1e3,
        'higher_is_better': True,
        'scaling': 'weak',
        'vs_baseline': None,
        'dtype': 'f16' if args.mixed_precision else 'bf16',
        'data': 'synthetic',
        'config': {
            'workload': '{}: dichotomised-Gaussian sl{} calcium signals, {} '
                        'neurons, batch {}/GPU, calciumgan num_units {} k 24 s 2 '
                        'm {} layer_norm, wgan-gp n_critic 5 lambda 10, Keras '
                        'Adam 1e-4'.format(cfg_name, args.seq_len, args.neurons,
                                           B, args.num_units, args.m),
            'global_batch': world * B,
            'seq_len': args.seq_len,
            'parallelism': 'dp{}'.format(world),
            'launch': 'hipGraph replay of train()' if getattr(
                gan, '_use_graph', False) else 'eager launches',
            # (round 5; CALCIUMGAN_L1_LINEAR=0 convolves x^ like the other segments;
            # FLOP counts below are of the work executed either way)
            'critic_layer1_on_interpolate':
                'a*y_real + (1-a)*y_fake from the stored outputs (cg_lrelu_mix)'
                if l1_mix else 'convolved',
        },
        'gflop_per_sample_step': fl['total'] / 1e9,
        'model_tflops': value * fl['total'] / 1e12,
        'final_losses': losses,
        'roofline': roofline,
    }
    if world == 1 and not args.no_cpu_baseline:
      line['cpu_baseline'] = cpu_baseline(hp, args.cpu_batch, args.cpu_steps)
    else:
      line['cpu_baseline'] = None
    print(json.dumps(line))
  if world > 1:
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
