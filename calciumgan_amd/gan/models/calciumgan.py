"""1-D CalciumGAN generator / discriminator behind the reference's model API
(gan/models/calciumgan.py:10-12, :22-103, :141-192).

The objects expose what the reference's callers use on a Keras Model:
``model(x, training=...)``, ``trainable_variables``, ``get_weights`` /
``set_weights`` (Keras order, TensorFlow kernel layouts -- SURVEY Appendix C)
and ``summary``.  Compute runs in the gfx950 kernel library only.
"""
import numpy as np
import torch

from ... import geometry as geo
from ... import nets
from ..._lib import use as _lib_use
from .registry import register

# every model of one process draws initial weights from this stream so that
# data-parallel ranks start identical (main.py:11-12 seeds with 1234)
_INIT_SEED = 1234


def _device():
  if not torch.cuda.is_available():
    raise RuntimeError('calciumgan_amd needs a HIP device (no CPU fallback)')
  return torch.device('cuda', torch.cuda.current_device())


@register('calciumgan')
def get_calciumgan(hparams):
  return generator(hparams), discriminator(hparams)


calculate_noise_shape = geo.calculate_noise_shape


class _Model(object):
  name = 'model'

  def __init__(self, net):
    self.net = net

  @property
  def trainable_variables(self):
    """(BatchNormalization's moving statistics are weights -- get_weights() --
    but not trainable variables, as in Keras)"""
    p = self.net.params
    return [v for i, v in enumerate(p.views) if i not in p.frozen]

  trainable_weights = trainable_variables

  def get_weights(self):
    return self.net.params.get_weights()

  def set_weights(self, weights):
    self.net.params.set_weights(weights)
    self.net.repack()

  def count_params(self):
    """Keras Model.count_params(): every weight, the non-trainable ones
    (BatchNormalization's moving statistics) included; utils.count_trainable_params
    / count_trainable_params() give the trainable figure."""
    return int(sum(int(np.prod(v.shape)) for v in self.net.params.views))

  def count_trainable_params(self):
    return self.net.params.count

  def summary(self):
    print('Model: "{}"'.format(self.name))
    for i, v in enumerate(self.net.params.views):
      print('  [{:02d}] {:<22} {}'.format(i, str(tuple(v.shape)),
                                          int(np.prod(v.shape))))
    total, trainable = self.count_params(), self.count_trainable_params()
    print('Total params: {:,}'.format(total))
    print('Trainable params: {:,}'.format(trainable))
    print('Non-trainable params: {:,}'.format(total - trainable))


class Generator(_Model):
  name = 'generator'

  def __call__(self, noise, training=False):
    """noise (B, noise_dim) -> (B, L, C) float32 (sigmoid when normalize).
    training defaults to False as for a Keras model called without it: with
    --batch_norm the moving statistics normalise and are left alone (the
    algorithms pass training explicitly; calciumgan.py:22-103)."""
    _lib_use(self.net.precision)
    noise = torch.as_tensor(noise, dtype=torch.float32).to(
        self.net.device).contiguous()
    ws = self.net.workspace(noise.shape[0])
    fake = ws.forward(noise, training=training)
    return fake[:, :, :self.net.C].clone()


class Discriminator(_Model):
  name = 'discriminator'

  def __init__(self, net, hparams):
    super().__init__(net)
    self._m = hparams.m
    self._gen = torch.Generator().manual_seed(_INIT_SEED + 17)

  def __call__(self, signals, training=True, shifts=None):
    """signals (B, L, C) -> (B, 1) float32.  PhaseShuffle draws one shift per
    layer per call, in training and inference alike (calciumgan.py:117)."""
    net = self.net
    _lib_use(net.precision)
    x = torch.as_tensor(signals, dtype=torch.float32).to(net.device).contiguous()
    B = x.shape[0]
    ws = net.workspace(B)
    plan = ws.plan(B, B, None)
    if shifts is None:
      shifts = torch.randint(-self._m, self._m + 1, (4,), generator=self._gen)
    plan.shifts.copy_(
        torch.as_tensor(shifts, dtype=torch.int32).reshape(4, 1))
    lay = net.layers[0]
    from ..._lib import call
    call('cg_cast_pad', nets._p(x), nets._p(ws.act[0]), B * lay.lin, lay.cin,
         lay.cin, lay.cinp, nets._stream())
    plan.forward()
    return ws.d_out[:B].clone().reshape(B, 1)


def generator(hparams, padding='same'):
  """calciumgan.py:22-103."""
  rng = np.random.RandomState(_INIT_SEED)
  return Generator(nets.GeneratorNet(hparams, _device(), rng))


def discriminator(hparams, padding='same'):
  """calciumgan.py:141-192."""
  rng = np.random.RandomState(_INIT_SEED + 1)
  return Discriminator(nets.DiscriminatorNet(hparams, _device(), rng), hparams)
