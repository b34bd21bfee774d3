"""Scalar logger with the reference's tags (gan/utils/summary_helper.py:
Summary.scalar :98-113, Summary.log :559-588).  TensorBoard / matplotlib are
not part of the hot path (and not installed here): scalars go to JSON-lines
files, training at <output_dir>/scalars.jsonl and validation at
<output_dir>/validation/scalars.jsonl (the reference's two writer dirs,
:32-40).  --profile maps to rocprofv3 (run the command under it); the
trace/export hooks are accepted and ignored."""
import json
import os


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
    os.makedirs(self._validation_dir, exist_ok=True)
    self._files = {
        True: os.path.join(self._train_dir, 'scalars.jsonl'),
        False: os.path.join(self._validation_dir, 'scalars.jsonl')
    }

  def scalar(self, tag, value, step=0, training=True):
    with open(self._files[bool(training)], 'a') as f:
      f.write(json.dumps({'tag': tag, 'value': _to_float(value),
                          'step': int(step)}) + '\n')

  def profiler_trace(self):
    pass

  def profiler_export(self):
    pass

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
