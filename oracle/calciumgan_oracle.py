"""torch-CPU restatement of the CalciumGAN 1-D conv stack + WGAN-GP train step.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED upstream:
no TensorFlow here, no golden vectors in the reference.

Every function cites the reference file:line it restates (paths relative to
/root/reference).  All randomness (noise z, interpolation alpha, phase-shuffle
shifts) is INJECTED so that the HIP path and this oracle can be compared
element-wise on identical draws.

Layouts are the reference's: activations channels-last (B, L, C); weights in
Keras ``get_weights()`` order and TensorFlow kernel layouts:
  generator      [dense.W (nd, w*nd), dense.b,
                  (convT.W (k,1,Co,Ci), convT.b, ln.gamma, ln.beta) x5,
                  dense_out.W (C,C), dense_out.b]                 -> 24 arrays
  discriminator  [(conv.W (k,Ci,Co), conv.b) x5, dense.W (w*5U,1), dense.b]
                                                                  -> 12 arrays
"""
import math
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn.functional as F

__all__ = [
    'make_hparams', 'calculate_noise_shape', 'same_padding', 'conv1d_same',
    'conv1d_transpose_same', 'layer_norm', 'leaky_relu', 'activation_fn',
    'phase_shuffle',
    'phase_shuffle_index', 'init_generator', 'init_discriminator',
    'count_params', 'generator_nontrainable', 'BN_EPS', 'BN_MOMENTUM',
    'generator_forward', 'discriminator_forward',
    'interpolation', 'gradient_penalty', 'layer1_mix_pre', 'discriminator_loss',
    'generator_loss', 'd_step_grads', 'g_step_grads', 'keras_adam',
    'signal_metrics', 'draw_randomness', 'OracleGAN', 'bf16_round',
    'f16_round', 'DynamicLossScale',
    'LEAKY_ALPHA', 'LN_EPS', 'NUM_CONVS',
]

LEAKY_ALPHA = 0.3  # Keras LeakyReLU() default alpha; gan/models/utils.py:6-8
LN_EPS = 1e-3  # Keras LayerNormalization default epsilon; calciumgan.py:44-45
NUM_CONVS = 5  # calciumgan.py:26 (num_convolutions=5) and the 5 D layers


# ---------------------------------------------------------------------------
# hparams
# ---------------------------------------------------------------------------
def make_hparams(sequence_length,
                 num_channels,
                 num_units=32,
                 kernel_size=24,
                 strides=2,
                 noise_dim=32,
                 m=2,
                 layer_norm=True,
                 batch_norm=False,
                 normalize=True,
                 gradient_penalty=10.0,
                 n_critic=5,
                 learning_rate=1e-4,
                 signals_min=0.0,
                 signals_max=1.0):
  """The hparams fields the hot path reads (main.py:227-262 defaults;
  dataset_helper.py:113-144 for the dataset-derived ones)."""
  return SimpleNamespace(
      signal_shape=(sequence_length, num_channels),
      sequence_length=sequence_length,
      num_channels=num_channels,
      num_neurons=num_channels,
      num_units=num_units,
      kernel_size=kernel_size,
      strides=strides,
      noise_dim=noise_dim,
      noise_shape=(noise_dim,),
      m=m,
      layer_norm=layer_norm,
      batch_norm=batch_norm,
      normalize=normalize,
      activation='leakyrelu',
      gradient_penalty=gradient_penalty,
      n_critic=n_critic,
      learning_rate=learning_rate,
      signals_min=signals_min,
      signals_max=signals_max,
      conv2d=False,
      mixed_precision=False,
      model='calciumgan',
      algorithm='wgan-gp')


def calculate_noise_shape(output_shape, noise_dim, num_convolutions, strides):
  """calciumgan.py:15-19 -- w = L / strides**5 must be an integer."""
  w = output_shape[0] / (strides**num_convolutions)
  if not float(w).is_integer():
    raise ValueError('Conv1D: w {} is not an integer.'.format(w))
  return (int(w), noise_dim)


# ---------------------------------------------------------------------------
# bf16 storage emulation (used to tighten tolerances against the bf16 HIP path)
# ---------------------------------------------------------------------------
class _RoundBF16(torch.autograd.Function):
  """Round-to-nearest-even to bf16 in forward AND in backward (the HIP path
  stores both activations and their gradients in bf16)."""

  @staticmethod
  def forward(ctx, x):
    return x.to(torch.bfloat16).to(x.dtype)

  @staticmethod
  def backward(ctx, g):
    return _RoundBF16.apply(g)


def bf16_round(x):
  return _RoundBF16.apply(x)


class _RoundF16(torch.autograd.Function):
  """Straight-through IEEE fp16 rounding (mixed_float16 storage points; values
  past 65504 become infinity, as the fp16 kernels store them)."""

  @staticmethod
  def forward(ctx, x):
    return x.to(torch.float16).to(x.dtype)

  @staticmethod
  def backward(ctx, g):
    return g


def f16_round(x):
  return _RoundF16.apply(x)


class DynamicLossScale(object):
  """tf.mixed_precision.experimental.DynamicLossScale as LossScaleOptimizer
  drives it (gan/algorithms/optimizer.py:10-12,23-34) [ext, TF 2.3]: initial
  scale 2**15; an update whose gradients are all finite is applied, and
  `increment_period` (2000) consecutive ones double the scale; a non-finite
  gradient skips the update (the inner optimizer's iteration count does not
  advance), halves the scale (floor 1) and restarts the count."""

  def __init__(self, initial=2.0**15, increment_period=2000, multiplier=2.0):
    self.scale = float(initial)
    self.period = int(increment_period)
    self.multiplier = float(multiplier)
    self.good_steps = 0

  def update(self, grads):
    """grads: the (unscaled) gradients of this update.  Returns True when the
    update is to be applied."""
    finite = all(bool(torch.isfinite(g).all()) for g in grads)
    if finite:
      if self.good_steps + 1 >= self.period:
        new = self.scale * self.multiplier
        if math.isfinite(new):
          self.scale = new
        self.good_steps = 0
      else:
        self.good_steps += 1
    else:
      self.scale = max(self.scale / self.multiplier, 1.0)
      self.good_steps = 0
    return finite


def _is_rounding(q):
  """Whether q is one of the storage-point emulations (tests pass `lambda x: x`
  for the f32 oracle as well as the default _ident)."""
  return q in (bf16_round, f16_round)


def _ident(x):
  return x


# ---------------------------------------------------------------------------
# layer semantics
# ---------------------------------------------------------------------------
def same_padding(length, kernel_size, strides):
  """TF 'same' padding for a strided conv: L_out = ceil(L/s), total pad
  max((L_out-1)*s + k - L, 0), the extra element goes to the right."""
  out = -(-length // strides)
  total = max((out - 1) * strides + kernel_size - length, 0)
  left = total // 2
  return out, left, total - left


def conv1d_same(x, kernel, bias, strides):
  """layers.Conv1D(padding='same') -- calciumgan.py:145-149.
  x (B,L,Ci); kernel (k,Ci,Co) cross-correlation; returns (B,ceil(L/s),Co)."""
  k = kernel.shape[0]
  _, left, right = same_padding(x.shape[1], k, strides)
  xt = F.pad(x.transpose(1, 2), (left, right))
  w = kernel.permute(2, 1, 0)  # (Co,Ci,k)
  y = F.conv1d(xt, w, bias, stride=strides)
  return y.transpose(1, 2)


def conv1d_transpose_same(x, kernel, bias, strides):
  """Conv1DTranspose -- gan/models/utils.py:65-94: expand to (B,L,1,Ci),
  Conv2DTranspose((k,1),(s,1),'same'), squeeze.  kernel (k,1,Co,Ci).  It is the
  input-gradient of the 'same' strided conv over an input of length s*L:
  y[o] += x[i] * W[kk], o = s*i + kk - pad_left."""
  k = kernel.shape[0]
  l_out = x.shape[1] * strides
  _, left, right = same_padding(l_out, k, strides)
  w = kernel[:, 0].permute(2, 1, 0)  # (Ci,Co,k)
  # full transposed conv has length (L-1)*s + k; crop [left, left + l_out)
  full = F.conv_transpose1d(x.transpose(1, 2), w, None, stride=strides)
  need = left + l_out
  if full.shape[2] < need:
    full = F.pad(full, (0, need - full.shape[2]))
  y = full[:, :, left:need]
  if bias is not None:
    y = y + bias[None, :, None]
  return y.transpose(1, 2)


def layer_norm(x, gamma, beta, eps=LN_EPS):
  """layers.LayerNormalization() -- axis=-1, biased variance, eps inside the
  sqrt (calciumgan.py:44-45)."""
  mean = x.mean(dim=-1, keepdim=True)
  var = ((x - mean)**2).mean(dim=-1, keepdim=True)
  return (x - mean) * torch.rsqrt(var + eps) * gamma + beta


def leaky_relu(x, alpha=LEAKY_ALPHA):
  """activation_fn('leakyrelu') -- gan/models/utils.py:6-8."""
  return torch.where(x > 0, x, alpha * x)


def activation_fn(name):
  """gan/models/utils.py:6-8: LeakyReLU() for 'leakyrelu', else
  layers.Activation(name) (Keras activation names)."""
  table = {
      'leakyrelu': leaky_relu,
      'relu': torch.relu,
      'linear': lambda x: x,
      'tanh': torch.tanh,
      'sigmoid': torch.sigmoid,
      'elu': torch.nn.functional.elu,
      'selu': torch.nn.functional.selu,
      'softplus': torch.nn.functional.softplus,
      'swish': torch.nn.functional.silu,
  }
  if name not in table:
    raise ValueError('unknown activation {}'.format(name))
  return table[name]


def phase_shuffle_index(w, shift):
  """Source index map of PhaseShuffle (calciumgan.py:117-138) for one shift:
  out[t] = x[idx[t]].  shift>0: reflect-pad right by shift, take
  [shift, w+shift); else reflect-pad left by |shift|, take [0, w).  tf.pad
  'reflect' mirrors without repeating the edge sample."""
  t = np.arange(w)
  if shift > 0:
    u = t + shift
    return np.where(u < w, u, 2 * (w - 1) - u).astype(np.int64)
  a = -shift
  return np.where(t < a, a - t, t - a).astype(np.int64)


def phase_shuffle(x, shift):
  """PhaseShuffle.call with the random shift injected; x (B,w,C)."""
  shift = int(shift)
  if shift == 0:
    return x
  idx = torch.from_numpy(phase_shuffle_index(x.shape[1], shift))
  return x.index_select(1, idx)


# ---------------------------------------------------------------------------
# parameters
# ---------------------------------------------------------------------------
def _glorot(rng, shape, fan_in, fan_out):
  limit = math.sqrt(6.0 / (fan_in + fan_out))
  return rng.uniform(-limit, limit, size=shape).astype(np.float32)


def generator_filters(hp):
  u = hp.num_units
  return [u * 5, u * 4, u * 3, u * 2, hp.num_channels]  # calciumgan.py:37-87


def discriminator_filters(hp):
  u = hp.num_units
  return [u, u * 2, u * 3, u * 4, u * 5]  # calciumgan.py:145-185


def init_generator(hp, rng):
  """Keras default initialisers (glorot_uniform kernels, zero biases, LN
  gamma=1 beta=0), arrays in get_weights() order."""
  w, nd = calculate_noise_shape(hp.signal_shape, hp.noise_dim, NUM_CONVS,
                                hp.strides)
  k = hp.kernel_size
  ws = [_glorot(rng, (nd, w * nd), nd, w * nd), np.zeros(w * nd, np.float32)]
  cin = nd
  for cout in generator_filters(hp):
    ws.append(_glorot(rng, (k, 1, cout, cin), k * cout, k * cin))
    ws.append(np.zeros(cout, np.float32))
    if getattr(hp, 'batch_norm', False):
      # layers.BatchNormalization(): gamma 1, beta 0, moving_mean 0,
      # moving_variance 1 (get_weights() order) [ext]
      ws += [np.ones(cout, np.float32), np.zeros(cout, np.float32),
             np.zeros(cout, np.float32), np.ones(cout, np.float32)]
    if hp.layer_norm:
      ws.append(np.ones(cout, np.float32))
      ws.append(np.zeros(cout, np.float32))
    cin = cout
  c = hp.num_channels
  ws.append(_glorot(rng, (c, c), c, c))
  ws.append(np.zeros(c, np.float32))
  return ws


def init_discriminator(hp, rng):
  k = hp.kernel_size
  ws = []
  cin = hp.num_channels
  for cout in discriminator_filters(hp):
    ws.append(_glorot(rng, (k, cin, cout), k * cin, k * cout))
    ws.append(np.zeros(cout, np.float32))
    cin = cout
  length = hp.signal_shape[0]
  for _ in range(NUM_CONVS):
    length = -(-length // hp.strides)
  flat = length * cin
  ws.append(_glorot(rng, (flat, 1), flat, 1))
  ws.append(np.zeros(1, np.float32))
  return ws


def count_params(weights):
  """count_trainable_params -- gan/models/utils.py:11-14."""
  return int(sum(int(np.prod(w.shape)) for w in weights))


BN_EPS = 1e-3        # layers.BatchNormalization() defaults [ext]
BN_MOMENTUM = 0.99


def generator_nontrainable(hp):
  """Indices (get_weights() order) of the generator's non-trainable arrays: the
  moving mean / variance of every BatchNormalization layer."""
  out = []
  i = 2
  for _ in range(NUM_CONVS):
    i += 2
    if getattr(hp, 'batch_norm', False):
      out += [i + 2, i + 3]
      i += 4
    if hp.layer_norm:
      i += 2
  return out


# ---------------------------------------------------------------------------
# models
# ---------------------------------------------------------------------------
def generator_forward(weights, z, hp, q=_ident, wq=_ident, training=True,
                      bn_updates=None):
  """generator -- calciumgan.py:22-103.  z (B, noise_dim) -> (B, L, C).
  q rounds stored activations, wq rounds the weight operands (bf16 emulation);
  both identity for the plain fp32 oracle.  batch_norm (calciumgan.py:42-43):
  training = batch statistics over (B, L) per channel, biased variance;
  bn_updates (a dict) receives {weight index: new moving statistic} -- Keras
  updates the moving averages on every training=True call; training = False
  normalises with the moving statistics (GAN.generate / validate)."""
  shape = calculate_noise_shape(hp.signal_shape, hp.noise_dim, NUM_CONVS,
                                hp.strides)
  it = iter(weights)
  pos = [0]

  def take():
    pos[0] += 1
    return next(it)

  act = activation_fn(getattr(hp, 'activation', 'leakyrelu'))
  dw, db = take(), take()
  x = q(act(q(z) @ wq(dw) + db))  # :32-33
  x = x.reshape(z.shape[0], shape[0], shape[1])  # :34
  for _ in range(NUM_CONVS):
    cw, cb = take(), take()
    x = q(conv1d_transpose_same(x, wq(cw), cb, hp.strides))
    if getattr(hp, 'batch_norm', False):
      g, b, mm, mv = take(), take(), take(), take()
      if training:
        mean = x.mean(dim=(0, 1))
        var = ((x - mean)**2).mean(dim=(0, 1))
        if bn_updates is not None:
          bn_updates[pos[0] - 2] = (mm * BN_MOMENTUM +
                                    mean.detach() * (1 - BN_MOMENTUM))
          bn_updates[pos[0] - 1] = (mv * BN_MOMENTUM +
                                    var.detach() * (1 - BN_MOMENTUM))
      else:
        mean, var = mm, mv
      x = (x - mean) * torch.rsqrt(var + BN_EPS) * g + b
      if hp.layer_norm:
        x = q(x)  # (stored between the two normalisations)
    if hp.layer_norm:
      g, b = take(), take()
      x = layer_norm(x, g, b)
    x = q(act(x))
  ow, ob = take(), take()
  x = x @ wq(ow) + ob  # Dense on the last axis, :96
  if hp.normalize:
    x = torch.sigmoid(x)  # :98-99
  return x


def discriminator_forward(weights, x, shifts, hp, q=_ident, wq=_ident, pre1=None):
  """discriminator -- calciumgan.py:141-192.  x (B, L, C) -> (B, 1).
  shifts: 4 ints, the PhaseShuffle draws after layers 1-4 of THIS call.
  pre1 (storage-point emulation only, see layer1_mix_pre): VALUES of the first
  layer's pre-activation; derivatives stay those of the convolution."""
  assert len(shifts) == NUM_CONVS - 1
  it = iter(weights)
  act = activation_fn(getattr(hp, 'activation', 'leakyrelu'))
  x = q(x)
  for layer in range(NUM_CONVS):
    cw, cb = next(it), next(it)
    y = conv1d_same(x, wq(cw), cb, hp.strides)
    if layer == 0 and pre1 is not None:
      y = y + (pre1 - y).detach()
    x = q(act(y))
    if layer < NUM_CONVS - 1:
      x = phase_shuffle(x, shifts[layer])
  dw, db = next(it), next(it)
  x = x.reshape(x.shape[0], -1)  # Flatten: row-major (t, c), :188
  return x @ wq(dw) + db


# ---------------------------------------------------------------------------
# WGAN-GP (gan/algorithms/wgan_gp.py)
# ---------------------------------------------------------------------------
def generator_loss(fake_output):
  """wgan_gp.py:19-20."""
  return -fake_output.mean()


def interpolation(real, fake, alpha):
  """wgan_gp.py:38-41 with alpha (B,) injected."""
  a = alpha.reshape(-1, 1, 1)
  return a * real + (1 - a) * fake


# The emulation of the HIP path's storage points (q != identity) follows where that
# path rounds, not what the reference computes.  Since round 5 the HIP path forms
# the critic's first layer on x^ from the layer's STORED outputs on real and fake
# at large batches (a convolution is linear; calciumgan_amd.nets._L1_LINEAR); set
# this flag and the emulation takes the same values there (tools/probe/
# critic_noise.py, tests of that form).  The f32 oracle (q = identity: the
# restatement of the reference every parity bar is stated against) never does.
EMULATE_LAYER1_MIX = False


def layer1_mix_pre(dis_weights, real, fake, alpha, hp, q, wq):
  """Pre-activation of the critic's first Conv1D on x^ = a real + (1 - a) fake as
  the HIP path forms it: a y_real + (1 - a) y_fake with y = act^-1 of the stored
  (q-rounded) activations.  None when the activation has no such inverse."""
  name = getattr(hp, 'activation', 'leakyrelu')
  if name not in ('leakyrelu', 'linear'):
    return None
  slope = LEAKY_ALPHA if name == 'leakyrelu' else 1.0
  act = activation_fn(name)
  with torch.no_grad():
    cw, cb = wq(dis_weights[0].detach()), dis_weights[1].detach()
    hr = q(act(conv1d_same(q(real.detach()), cw, cb, hp.strides)))
    hf = q(act(conv1d_same(q(fake.detach()), cw, cb, hp.strides)))
    inv = lambda h: torch.where(h > 0, h, h * (1.0 / slope))
    a = alpha.reshape(-1, 1, 1)
    return a * inv(hr) + (1 - a) * inv(hf)


def gradient_penalty(dis_weights, real, fake, alpha, shifts, hp, q=_ident,
                     wq=_ident, create_graph=True):
  """wgan_gp.py:43-50.  Returns (gp, per-sample norms, gradient)."""
  inter = interpolation(real, fake, alpha)
  if not inter.requires_grad:
    inter = inter.detach().requires_grad_(True)
  pre1 = None
  if EMULATE_LAYER1_MIX and q is not _ident and _is_rounding(q):
    pre1 = layer1_mix_pre(dis_weights, real, fake, alpha, hp, q, wq)
  out = discriminator_forward(dis_weights, inter, shifts, hp, q, wq, pre1=pre1)
  grad, = torch.autograd.grad(out.sum(), inter, create_graph=create_graph)
  norm = grad.reshape(grad.shape[0], -1).pow(2).sum(dim=1).sqrt()  # tf.norm
  return ((norm - 1.0)**2).mean(), norm, grad


def discriminator_loss(real_output, fake_output, gp, penalty):
  """wgan_gp.py:52-62."""
  return -real_output.mean() + fake_output.mean() + penalty * gp


def d_step_grads(gen_weights, dis_weights, real, z, alpha, shifts_real,
                 shifts_fake, shifts_inter, hp, q=_ident, wq=_ident):
  """_train_discriminator up to (not including) the optimizer update --
  wgan_gp.py:64-80.  Returns dict(loss, gp, grads, fake, real_out, fake_out,
  norm)."""
  dis = [w.detach().clone().requires_grad_(True) for w in dis_weights]
  bn_updates = {}
  with torch.no_grad():
    fake = generator_forward(gen_weights, z, hp, q, wq, bn_updates=bn_updates)
  real_out = discriminator_forward(dis, real, shifts_real, hp, q, wq)
  fake_out = discriminator_forward(dis, fake, shifts_fake, hp, q, wq)
  gp, norm, grad = gradient_penalty(dis, real, fake, alpha, shifts_inter, hp, q,
                                    wq)
  loss = discriminator_loss(real_out, fake_out, gp, hp.gradient_penalty)
  grads = torch.autograd.grad(loss, dis, allow_unused=True)
  grads = [
      torch.zeros_like(w) if g is None else g for g, w in zip(grads, dis)
  ]
  return dict(
      bn_updates=bn_updates,
      loss=loss.detach(),
      gp=gp.detach(),
      grads=[g.detach() for g in grads],
      fake=fake,
      real_out=real_out.detach(),
      fake_out=fake_out.detach(),
      norm=norm.detach(),
      gradient=grad.detach())


def g_step_grads(gen_weights, dis_weights, z, shifts, hp, q=_ident, wq=_ident):
  """_train_generator up to the optimizer update -- wgan_gp.py:22-36."""
  gen = [w.detach().clone().requires_grad_(True) for w in gen_weights]
  bn_updates = {}
  fake = generator_forward(gen, z, hp, q, wq, bn_updates=bn_updates)
  fake_in = fake
  if q is not _ident:
    fake_in = fake  # D rounds its own input
  out = discriminator_forward(dis_weights, fake_in, shifts, hp, q, wq)
  loss = generator_loss(out)
  # (the moving statistics of BatchNormalization take no gradient)
  grads = torch.autograd.grad(loss, gen, allow_unused=True)
  grads = [torch.zeros_like(w) if g is None else g for g, w in zip(grads, gen)]
  bn_updates = {k: v.detach() for k, v in bn_updates.items()}
  return dict(
      bn_updates=bn_updates,
      loss=loss.detach(), grads=[g.detach() for g in grads],
      fake=fake.detach(), fake_out=out.detach())


def keras_adam(p, g, m, v, t, lr, beta1=0.9, beta2=0.999, eps=1e-7):
  """tf.keras.optimizers.Adam (optimizer.py:9) dense update, step t>=1:
  lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m,v EMA; p -= lr_t*m/(sqrt(v)+eps).
  NOT torch.optim.Adam's epsilon placement.  In-place."""
  lr_t = lr * math.sqrt(1.0 - beta2**t) / (1.0 - beta1**t)
  m.mul_(beta1).add_(g, alpha=1.0 - beta1)
  v.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
  p.sub_(lr_t * m / (v.sqrt() + eps))


def signal_metrics(real, fake, signals_min=0.0, signals_max=1.0,
                   normalize=True):
  """GAN.metrics -- gan.py:32-41 + signals_metrics.py:9-28 + utils.py:30-32:
  denormalise, reduce over the channel axis, MSE over (B, L)."""
  if normalize:
    scale = signals_max - signals_min
    real = real * scale + signals_min
    fake = fake * scale + signals_min

  def mse(a, b):
    return ((a - b)**2).mean()

  return {
      'signals_metrics/min':
          mse(real.min(dim=-1).values, fake.min(dim=-1).values),
      'signals_metrics/max':
          mse(real.max(dim=-1).values, fake.max(dim=-1).values),
      'signals_metrics/mean':
          mse(real.mean(dim=-1), fake.mean(dim=-1)),
      'signals_metrics/std':
          mse(real.std(dim=-1, unbiased=False), fake.std(dim=-1,
                                                        unbiased=False)),
  }


# ---------------------------------------------------------------------------
# randomness + full train()
# ---------------------------------------------------------------------------
def draw_randomness(hp, batch_size, seed, n_critic=None):
  """All random draws of one WGAN_GP.train call (wgan_gp.py:82-95): per critic
  step z (gan.py:29-30), alpha (wgan_gp.py:40) and 3x4 phase shifts
  (calciumgan.py:121-124, U{-m..m}); for the generator step z and 4 shifts."""
  rng = np.random.RandomState(seed)
  n_critic = hp.n_critic if n_critic is None else n_critic
  m = hp.m

  def shifts():
    return rng.randint(-m, m + 1, size=NUM_CONVS - 1).astype(np.int32)

  critic = []
  for _ in range(n_critic):
    critic.append(
        dict(
            z=rng.standard_normal((batch_size, hp.noise_dim)).astype(
                np.float32),
            alpha=rng.uniform(0, 1, size=batch_size).astype(np.float32),
            shifts_real=shifts(),
            shifts_fake=shifts(),
            shifts_inter=shifts()))
  gen = dict(
      z=rng.standard_normal((batch_size, hp.noise_dim)).astype(np.float32),
      shifts=shifts())
  return dict(critic=critic, gen=gen)


class OracleGAN(object):
  """Stateful oracle of WGAN_GP (weights + Keras-Adam state), mirroring the
  reference object surface used by main.py (train / validate / generate)."""

  def __init__(self, hp, gen_weights, dis_weights, dtype=torch.float32,
               emulate_bf16=False, emulate_f16=False, loss_scaling=False):
    """emulate_bf16 / emulate_f16: round stored activations and weight
    operands like the bf16 / mixed_float16 kernels.  loss_scaling: wrap both
    optimizers in DynamicLossScale (the backward here is f32, so the scale
    itself changes nothing -- its skip / halve / grow state machine does)."""
    self.hp = hp
    self.dtype = dtype
    self.gen = [torch.tensor(np.asarray(w), dtype=dtype) for w in gen_weights]
    self.dis = [torch.tensor(np.asarray(w), dtype=dtype) for w in dis_weights]
    self.gen_m = [torch.zeros_like(w) for w in self.gen]
    self.gen_v = [torch.zeros_like(w) for w in self.gen]
    self.dis_m = [torch.zeros_like(w) for w in self.dis]
    self.dis_v = [torch.zeros_like(w) for w in self.dis]
    self.gen_steps = 0
    self.dis_steps = 0
    rnd = bf16_round if emulate_bf16 else f16_round if emulate_f16 else _ident
    self.q = rnd
    self.wq = rnd
    self.gen_scale = DynamicLossScale() if loss_scaling else None
    self.dis_scale = DynamicLossScale() if loss_scaling else None

  def _t(self, a):
    return torch.as_tensor(np.asarray(a), dtype=self.dtype)

  def train_discriminator(self, inputs, r):
    res = d_step_grads(self.gen, self.dis, self._t(inputs), self._t(r['z']),
                       self._t(r['alpha']), r['shifts_real'], r['shifts_fake'],
                       r['shifts_inter'], self.hp, self.q, self.wq)
    self._apply_bn(res)  # (the forward pass ran: Keras has moved the averages)
    if self.dis_scale is not None and not self.dis_scale.update(res['grads']):
      return res  # non-finite gradients: update skipped
    self.dis_steps += 1
    for p, g, m, v in zip(self.dis, res['grads'], self.dis_m, self.dis_v):
      keras_adam(p, g, m, v, self.dis_steps, self.hp.learning_rate)
    return res

  def _apply_bn(self, res):
    for i, v in res.get('bn_updates', {}).items():
      self.gen[i] = v.to(self.dtype)

  def train_generator(self, inputs, r):
    res = g_step_grads(self.gen, self.dis, self._t(r['z']), r['shifts'],
                       self.hp, self.q, self.wq)
    self._apply_bn(res)
    if self.gen_scale is None or self.gen_scale.update(res['grads']):
      self.gen_steps += 1
      frozen = set(generator_nontrainable(self.hp))
      for i, (p, g, m, v) in enumerate(zip(self.gen, res['grads'], self.gen_m,
                                           self.gen_v)):
        if i not in frozen:
          keras_adam(p, g, m, v, self.gen_steps, self.hp.learning_rate)
    res['metrics'] = signal_metrics(
        self._t(inputs), res['fake'], self.hp.signals_min, self.hp.signals_max,
        self.hp.normalize)
    return res

  def train(self, inputs, rand):
    """WGAN_GP.train -- wgan_gp.py:82-95: n_critic critic updates on the SAME
    batch, then one generator update.  Returns (gen_loss, dis_loss, gp,
    metrics) as python floats / dict of floats."""
    dis_losses, gps = [], []
    for r in rand['critic']:
      res = self.train_discriminator(inputs, r)
      dis_losses.append(float(res['loss']))
      gps.append(float(res['gp']))
    res = self.train_generator(inputs, rand['gen'])
    metrics = {k: float(v) for k, v in res['metrics'].items()}
    return (float(res['loss']), float(np.mean(dis_losses)),
            float(np.mean(gps)), metrics)

  def generate(self, z, denorm=False):
    """GAN.generate -- gan.py:92-97."""
    with torch.no_grad():
      fake = generator_forward(self.gen, self._t(z), self.hp, self.q, self.wq,
                               training=False)
    if denorm:
      fake = fake * (self.hp.signals_max -
                     self.hp.signals_min) + self.hp.signals_min
    return fake
