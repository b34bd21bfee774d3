"""Scalar logger with the reference's tags and directories
(gan/utils/summary_helper.py: writers :32-40, Summary.scalar :98-101,
Summary.log :559-588, profiler hooks :115-119).

Scalars go to TensorBoard event files -- <output_dir>/events.out.tfevents.* for
training, <output_dir>/validation/... for validation, the reference's two
writer directories, written by tb_events.py without TensorFlow -- and, for
grep-ability, to JSON lines next to them (scalars.jsonl).  Images, histograms
and trace plots (matplotlib) are outside the hot path and not produced.

--profile: the reference traces batches 2-6 of the second epoch with the TF
profiler (main.py:45-52).  Here profiler_trace() switches that window to eager
launches with the kernel library's launch profiler on (cg_profile_enable: every
MFMA-kernel launch carries its own begin / end timestamps) and
profiler_export() writes <output_dir>/profiler/mfma_kernels.json: launches,
total and mean duration per kernel family per batch of the window.  For a
whole-process trace run the same command under
`rocprofv3 --kernel-trace --stats -d <output_dir>/profiler -- python3 main.py ...`
(the command line is printed).
"""
import json
import os
import sys

from . import tb_events


def _to_float(v):
  if hasattr(v, 'item'):
    return float(v.item())
  return float(v)


class Summary(object):

  def __init__(self, hparams, policy=None):
    self._hparams = hparams
    self._policy = policy
    self._train_dir = hparams.output_dir
    self._validation_dir = os.path.join(hparams.output_dir, 'validation')
    self._profiler_dir = os.path.join(hparams.output_dir, 'profiler')
    os.makedirs(self._validation_dir, exist_ok=True)
    self._files = {
        True: os.path.join(self._train_dir, 'scalars.jsonl'),
        False: os.path.join(self._validation_dir, 'scalars.jsonl')
    }
    self._writers = {
        True: tb_events.EventFileWriter(self._train_dir),
        False: tb_events.EventFileWriter(self._validation_dir)
    }
    self._profile = None

  def scalar(self, tag, value, step=0, training=True):
    value = _to_float(value)
    self._writers[bool(training)].scalar(tag, value, step=step)
    with open(self._files[bool(training)], 'a') as f:
      f.write(json.dumps({'tag': tag, 'value': value, 'step': int(step)}) + '\n')

  # -- profiler window (main.py:45-52) ------------------------------------------
  def profiler_trace(self, gan=None, capacity=4096):
    """Start of the profiled window: eager launches with per-launch kernel
    timestamps (a captured graph cannot be instrumented)."""
    from ... import _lib
    if self._hparams.verbose:
      print('profiling window open; for a whole-process kernel trace run:\n'
            '  rocprofv3 --kernel-trace --stats -d {} -- python3 {}'.format(
                self._profiler_dir, ' '.join(
                    a for a in sys.argv if a != '--profile')))
    self._profile = dict(gan=gan, graph=getattr(gan, '_use_graph', None),
                         capacity=capacity)
    if gan is not None:
      gan._use_graph = False
    _lib.check(_lib.load().cg_profile_enable(capacity), 'cg_profile_enable')

  def profiler_export(self):
    """End of the window: durations of the cg_swconv / cg_wgrad launches since
    profiler_trace() -> <output_dir>/profiler/mfma_kernels.json."""
    import ctypes
    from ... import _lib
    if self._profile is None:
      return None
    cap = self._profile['capacity']
    ms = (ctypes.c_float * cap)()
    fam = (ctypes.c_int * cap)()
    n = _lib.load().cg_profile_collect(ms, fam, cap)
    if n < 0:
      raise RuntimeError('cg_profile_collect: HIP error {}'.format(-n))
    gan = self._profile['gan']
    if gan is not None and self._profile['graph'] is not None:
      gan._use_graph = self._profile['graph']
    self._profile = None
    out = {}
    for i in range(n):
      d = out.setdefault(('cg_swconv', 'cg_wgrad')[fam[i]],
                         dict(launches=0, total_ms=0.0))
      d['launches'] += 1
      d['total_ms'] += float(ms[i])
    for d in out.values():
      d['mean_us'] = d['total_ms'] / d['launches'] * 1e3
    os.makedirs(self._profiler_dir, exist_ok=True)
    path = os.path.join(self._profiler_dir, 'mfma_kernels.json')
    with open(path, 'w') as f:
      json.dump(dict(launches_recorded=n, capacity=cap, families=out), f,
                indent=1)
    return path

  def plot_traces(self, *args, **kwargs):
    pass

  def log(self, gen_loss, dis_loss, gradient_penalty, metrics=None, elapse=None,
          gan=None, step=0, training=True):
    """summary_helper.py:559-588 (scalar part)."""
    self.scalar('loss/generator', gen_loss, step=step, training=training)
    self.scalar('loss/discriminator', dis_loss, step=step, training=training)
    if gradient_penalty is not None:
      self.scalar('loss/gradient_penalty', gradient_penalty, step=step,
                  training=training)
    if metrics is not None:
      for tag, value in metrics.items():
        self.scalar(tag, value, step=step, training=training)
    if elapse is not None:
      self.scalar('elapse', elapse, step=step, training=training)
    if training and gan is not None and getattr(
        gan.gen_optimizer, 'loss_scale', None) is not None:
      # summary_helper.py:77-78,586-588 logs the loss scale under mixed precision
      self.scalar('model/loss_scale', gan.dis_optimizer.loss_scale[0], step=step,
                  training=training)
