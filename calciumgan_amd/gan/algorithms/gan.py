"""Base GAN object -- gan/algorithms/gan.py:13-97 (the pieces WGAN-GP uses:
get_noise, metrics, _step/validate, generate).  The vanilla BCE GAN train
step (gan.py:72-85) is outside the north-star path and not implemented.
"""
import torch

from ... import _lib
from ... import nets
from ... import parallel
from .optimizer import Optimizer
from .registry import register

# main.py:11-12 (CALCIUMGAN_SEED: another draw of the noise / interpolation /
# shift streams, for seed-to-seed comparisons -- tools/e2e_seeds.sh)
_SEED = int(__import__('os').environ.get('CALCIUMGAN_SEED', '1234'))


@register('gan')
class GAN(object):

  def __init__(self, hparams, generator, discriminator, summary=None):
    self.generator = generator
    self.discriminator = discriminator
    self._summary = summary
    self.noise_shape = tuple(hparams.noise_shape)
    self.signal_shape = tuple(hparams.signal_shape)
    self._normalize = hparams.normalize
    self._signals_min = float(getattr(hparams, 'signals_min', 0.0))
    self._signals_max = float(getattr(hparams, 'signals_max', 1.0))
    if not hparams.normalize:
      self._signals_min, self._signals_max = 0.0, 1.0

    self.device = generator.net.device
    # the build of the kernel library both models compute with ('f16' under
    # hparams.mixed_precision); every entry point re-selects it
    self.precision = generator.net.precision
    if discriminator.net.precision != self.precision:
      raise ValueError('generator and discriminator differ in precision')
    self.gen_optimizer = Optimizer(hparams, self.device)
    self.dis_optimizer = Optimizer(hparams, self.device)

    self._sync = parallel.GradSync()
    self._streams = parallel.RandomStreams(_SEED, self.device, hparams.m)
    self._metrics_buf = torch.zeros(4, dtype=torch.float32, device=self.device)

  # -- helpers ---------------------------------------------------------------
  def _to_device(self, x):
    if not torch.is_tensor(x):
      x = torch.as_tensor(x)
    return x.to(device=self.device, dtype=torch.float32).contiguous()

  def get_noise(self, batch_size):
    """gan.py:29-30: N(0,1) of shape (batch,) + noise_shape."""
    return self._streams.noise(batch_size, self.noise_shape[0])

  def metrics(self, real, fake, fake_pitch=None):
    """gan.py:32-41 + signals_metrics.py:9-28: MSE between real and fake of the
    per-(sample, timestep) min / max / mean / std over channels, after
    denormalisation.  real (B, L, C) f32 contiguous; fake f32 with row pitch
    fake_pitch (defaults to C)."""
    B, L, C = real.shape
    rows = B * L
    rws = nets.reduce_ws(self.device)
    if rws is not None:
      # ordered reduction: the finishing launch stores the means
      buf = torch.empty(4, dtype=torch.float32, device=self.device)
    else:
      buf = torch.zeros(4, dtype=torch.float32, device=self.device)
    _lib.call('cg_signal_metrics', nets._p(real), nets._p(fake), nets._p(buf),
              rows, C, C, fake_pitch or C, self._signals_min,
              self._signals_max, nets._p(rws), nets._stream())
    if rws is None:
      buf.mul_(1.0 / rows)
    return {
        'signals_metrics/min': buf[0],
        'signals_metrics/max': buf[1],
        'signals_metrics/mean': buf[2],
        'signals_metrics/std': buf[3],
    }

  def train(self, inputs):
    raise NotImplementedError(
        "calciumgan_amd implements the 'wgan-gp' algorithm only; the vanilla "
        'BCE GAN step (gan/algorithms/gan.py:72-85) is out of scope')

  def validate(self, inputs):
    raise NotImplementedError("use algorithm 'wgan-gp'")

  def generate(self, noise, denorm=False):
    """gan.py:92-97."""
    fake = self.generator(noise, training=False)
    if denorm:
      fake = fake * (self._signals_max - self._signals_min) + self._signals_min
    return fake
