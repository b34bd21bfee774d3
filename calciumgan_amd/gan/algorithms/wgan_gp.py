"""WGAN-GP train step -- gan/algorithms/wgan_gp.py:12-95 -- as a hand-scheduled
sequence of gfx950 kernels (no autograd graph).

Critic step (reference :64-80).  real, fake and the interpolate x^ run through
the discriminator as ONE batch of 3B samples (per-segment phase shifts):
  1. G forward                       -> fake
  2. interpolate + pack              -> X0 = [real | fake | x^]       (:38-41)
  3. D forward over 3B               -> outputs, activations h_l
  4. delta_5 = c_seg * w_d * lrelu'(h_5), c = (-1/B, +1/B, 1); input-gradient
     chain over 3B down to layer 2, layer 1 only for the x^ segment -> g
  5. ||g||, gp = mean((||g||-1)^2), v = lambda*dgp/dg                 (:48-50)
  6. tangent-forward of v through the SAME masked linear chain (the second
     backward of the penalty; D is piecewise linear, so the mask derivative
     vanishes and biases get no penalty gradient), in place over the x^
     segment's activations
  7. weight gradients over the 3B batch: inputs [h_real | h_fake | tangent],
     output-gradients [delta_real | delta_fake | delta_x^]  == autodiff of
     -E[D(real)] + E[D(fake)] + lambda*gp
  8. (data parallel) all-reduce, Keras Adam, re-pack bf16 operands
Generator step (reference :22-36): G forward (activations kept), D forward and
input-gradient chain on fake, generator backward, Adam, signal metrics.
"""
import torch

from ... import _lib
from ... import nets
from .gan import GAN
from .registry import register


@register('wgan-gp')
class WGAN_GP(GAN):

  def __init__(self, hparams, generator, discriminator, summary=None):
    super().__init__(hparams, generator, discriminator, summary)
    self.penalty = float(hparams.gradient_penalty)
    self.n_critic = int(hparams.n_critic)
    self.conv2d = getattr(hparams, 'conv2d', False)
    if self.conv2d:
      raise ValueError('calciumgan_amd: conv2d models are out of scope')
    self._state = {}

  # -- per-batch-size state ---------------------------------------------------
  def _get_state(self, B):
    st = self._state.get(B)
    if st is None:
      dev = self.device
      dws = self.discriminator.net.workspace(3 * B)
      st = dict(
          gws=self.generator.net.workspace(B),
          dws=dws,
          critic=dws.plan(3 * B, B, 2 * B),
          gen=dws.plan(B, B, 0),
          norm=torch.zeros(B, dtype=torch.float32, device=dev),
          coef_gp=torch.zeros(B, dtype=torch.float32, device=dev),
          gp=torch.zeros(max(self.n_critic, 1), dtype=torch.float32, device=dev),
          loss=torch.zeros(max(self.n_critic, 1), 2, dtype=torch.float32,
                           device=dev),
          gen_loss=torch.zeros(1, dtype=torch.float32, device=dev))
      st['critic'].coef.copy_(torch.tensor([-1.0 / B, 1.0 / B, 1.0]))
      st['critic'].bias_coef.copy_(torch.tensor([-1.0 / B, 1.0 / B, 0.0]))
      st['critic'].build_jvp(2)
      st['gen'].coef.copy_(torch.tensor([-1.0 / B]))
      st['gen'].bias_coef.zero_()
      self._state[B] = st
    return st

  # -- losses (API parity; the fused kernels compute the same values) ---------
  def generator_loss(self, fake_output):
    """wgan_gp.py:19-20."""
    return -fake_output.mean()

  def _critic_forward(self, st, real, z, alpha, shifts, slot):
    """Steps 1-5 of the critic schedule; leaves g in st['critic'].gin."""
    net_d = self.discriminator.net
    B = real.shape[0]
    lay = net_d.layers[0]
    plan = st['critic']
    s = nets._stream()
    plan.shifts.copy_(shifts, non_blocking=True)
    fake = st['gws'].forward(z)
    _lib.call('cg_interp_pack', nets._p(real), nets._p(fake), nets._p(alpha),
              nets._p(st['dws'].act[0]), B, lay.lin, lay.cin, lay.cin, lay.cinp,
              lay.cinp, s)
    plan.forward()
    plan.backward_chain()
    n = lay.lin * lay.cinp
    _lib.call('cg_rownorm', nets._p(plan.gin), nets._p(st['norm']), B, n, s)
    _lib.call('cg_gp_finalize', nets._p(st['norm']), nets._p(st['gp'][slot:]),
              nets._p(st['coef_gp']), B, self.penalty, s)
    _lib.call('cg_critic_loss', nets._p(st['dws'].d_out),
              nets._p(st['gp'][slot:]), self.penalty,
              nets._p(st['loss'][slot]), B, s)
    return fake

  def _train_discriminator(self, inputs, r=None, slot=0):
    """wgan_gp.py:64-80."""
    real = self._to_device(inputs)
    B = real.shape[0]
    st = self._get_state(B)
    net_d = self.discriminator.net
    lay = net_d.layers[0]
    if r is None:
      z = self.get_noise(B)
      alpha = self._streams.alpha(B)
      shifts = self._streams.shifts(3)
    else:
      z = self._to_device(r['z'])
      alpha = self._to_device(r['alpha'])
      shifts = torch.stack([
          torch.as_tensor(r['shifts_real'], dtype=torch.int32),
          torch.as_tensor(r['shifts_fake'], dtype=torch.int32),
          torch.as_tensor(r['shifts_inter'], dtype=torch.int32)
      ], dim=1)
    self._critic_forward(st, real, z, alpha, shifts, slot)
    plan = st['critic']
    s = nets._stream()
    n = lay.lin * lay.cinp
    # v = lambda * dgp/dg, written over the x^ segment of X0
    _lib.call('cg_scale_rows', nets._p(plan.gin), nets._p(st['coef_gp']),
              nets._p(st['dws'].act[0][2 * B:]), B, n, s)
    plan.jvp_forward()
    net_d.params.grad.zero_()
    plan.weight_grads(bias_rows=2 * B)
    self._sync.all_reduce(net_d.params.grad)
    self.dis_optimizer.update(self.discriminator, self._sync.grad_scale)
    return st['loss'][slot, 0], st['gp'][slot]

  def _train_generator(self, inputs, r=None):
    """wgan_gp.py:22-36."""
    real = self._to_device(inputs)
    B = real.shape[0]
    st = self._get_state(B)
    net_g, net_d = self.generator.net, self.discriminator.net
    lay = net_d.layers[0]
    plan = st['gen']
    if r is None:
      z = self.get_noise(B)
      shifts = self._streams.shifts(1)
    else:
      z = self._to_device(r['z'])
      shifts = torch.as_tensor(r['shifts'], dtype=torch.int32).reshape(4, 1)
    s = nets._stream()
    plan.shifts.copy_(shifts, non_blocking=True)
    fake = st['gws'].forward(z)
    _lib.call('cg_cast_pad', nets._p(fake), nets._p(st['dws'].act[0]),
              B * lay.lin, lay.cin, lay.cinp, lay.cinp, s)
    plan.forward()
    _lib.call('cg_neg_mean', nets._p(st['dws'].d_out), nets._p(st['gen_loss']),
              B, s)
    plan.backward_chain()
    net_g.params.grad.zero_()
    st['gws'].backward(plan.gin)
    self._sync.all_reduce(net_g.params.grad)
    self.gen_optimizer.update(self.generator, self._sync.grad_scale)
    metrics = self.metrics(real, fake, fake_pitch=net_g.Cp)
    return st['gen_loss'][0], metrics

  def train(self, inputs, rand=None):
    """wgan_gp.py:82-95: n_critic critic updates on the SAME batch, then one
    generator update.  Returns (gen_loss, dis_loss, gradient_penalty, metrics)
    as 0-d device tensors (no host sync inside).  `rand` optionally injects the
    random draws (same structure as oracle.draw_randomness) for parity tests."""
    real = self._to_device(inputs)
    B = real.shape[0]
    st = self._get_state(B)
    for i in range(self.n_critic):
      self._train_discriminator(
          real, None if rand is None else rand['critic'][i], slot=i)
    gen_loss, metrics = self._train_generator(
        real, None if rand is None else rand['gen'])
    dis_loss = st['loss'][:self.n_critic, 0].mean()
    gradient_penalty = st['gp'][:self.n_critic].mean()
    return gen_loss.clone(), dis_loss, gradient_penalty, metrics

  def validate(self, inputs, rand=None):
    """gan.py:87-90 / :58-70 with the WGAN-GP loss (inner-gradient penalty, no
    parameter update).  Returns (fake, gen_loss, dis_loss, gp, metrics)."""
    real = self._to_device(inputs)
    B = real.shape[0]
    st = self._get_state(B)
    if rand is None:
      z = self.get_noise(B)
      alpha = self._streams.alpha(B)
      shifts = self._streams.shifts(3)
    else:
      z = self._to_device(rand['z'])
      alpha = self._to_device(rand['alpha'])
      shifts = torch.stack([
          torch.as_tensor(rand['shifts_real'], dtype=torch.int32),
          torch.as_tensor(rand['shifts_fake'], dtype=torch.int32),
          torch.as_tensor(rand['shifts_inter'], dtype=torch.int32)
      ], dim=1)
    fake = self._critic_forward(st, real, z, alpha, shifts, 0)
    C = self.generator.net.C
    metrics = self.metrics(real, fake, fake_pitch=self.generator.net.Cp)
    loss = st['loss'][0].clone()
    return (fake[:, :, :C].clone(), loss[1], loss[0], st['gp'][0].clone(),
            metrics)
