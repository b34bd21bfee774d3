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
import os

import torch

from ... import _lib
from ... import nets
from .gan import GAN
from .registry import register

# hipGraph capture of train(): the ~370 launches of one step replay as ONE
# graph on a single rank, and under data parallelism as 2 * n_critic + 3 graphs
# cut around the gradient all-reduces (the RCCL calls stay eager between
# replays and overlap the graphs that do not need their result), removing
# launch gaps.  CALCIUMGAN_GRAPH=0 disables it.
_GRAPH_WARMUP_CALLS = 2
# development knob: keep the multi-rank segmentation on a single rank
_FORCE_SPLIT = os.environ.get('CALCIUMGAN_SPLIT_SEGMENTS', '0') == '1'
# single rank: one generator pass for the fake batches of all critic updates
# of a step (CALCIUMGAN_BATCH_G=0: one pass per update, as under data
# parallelism, where each pass hides the previous update's all-reduce)
_BATCH_G = os.environ.get('CALCIUMGAN_BATCH_G', '1') != '0'
# data parallel, A/B only: wait for every gradient all-reduce right after it is
# started instead of overlapping it with the next segment
_DP_OVERLAP = os.environ.get('CALCIUMGAN_DP_OVERLAP', '1') != '0'
# the penalty norm's finishing sum, gp / coef / critic loss and v's per-sample
# scale as ONE launch (cg_gp_loss_scale) instead of three (A/B: =0)
_FUSE_GP = os.environ.get('CALCIUMGAN_FUSE_GP', '1') != '0'
# single rank, batched generator pass: the output Dense of that pass writes the
# critic's input buffers of ALL updates itself -- interpolation and packing in its
# epilogue (cg_dense_rows_interp), one input buffer per update -- instead of an
# f32 fake batch that n_critic cg_interp_pack launches read back (A/B: =0)
_FUSE_INTERP = os.environ.get('CALCIUMGAN_FUSE_INTERP', '1') != '0'
_METRIC_KEYS = ('signals_metrics/min', 'signals_metrics/max',
                'signals_metrics/mean', 'signals_metrics/std')
# pinned staging slots for the host-drawn inputs of a graph replay: the host may
# run this many steps ahead of the GPU before it waits for a slot's copy
_STAGING_SLOTS = 4


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
    self._use_graph = os.environ.get('CALCIUMGAN_GRAPH', '1') != '0'

  # -- per-batch-size state ---------------------------------------------------
  def _get_state(self, B):
    st = self._state.get(B)
    if st is None:
      dev = self.device
      dws = self.discriminator.net.workspace(3 * B)
      nc = self.n_critic
      # host-drawn inputs of a step on the device: [shifts int32 x (12 n + 4) |
      # Adam step sizes f32 x (n + 1)].  The plans read their PhaseShuffle draws
      # straight from it (update k: 12 ints at 12 k, the generator update's 4
      # behind them), so a graph replay needs ONE staged copy and no per-update
      # copies into the plans
      stage = torch.zeros(nc * 13 + 5, dtype=torch.int32, device=dev)
      st = dict(
          gws=self.generator.net.workspace(B),
          dws=dws,
          stage_dev=stage,
          critic=dws.plan(3 * B, B, 2 * B,
                          shifts=stage[:12].view(4, 3) if nc > 0 else None),
          gen=dws.plan(B, B, 0, want_norm=False,
                       shifts=stage[12 * nc:12 * nc + 4].view(4, 1)),
          norm=torch.zeros(B, dtype=torch.float32, device=dev),
          coef_gp=torch.zeros(B, dtype=torch.float32, device=dev),
          gp=torch.zeros(max(self.n_critic, 1), dtype=torch.float32, device=dev),
          loss=torch.zeros(max(self.n_critic, 1), 2, dtype=torch.float32,
                           device=dev),
          gen_loss=torch.zeros(1, dtype=torch.float32, device=dev),
          # the step's outputs [gen_loss, dis_loss, gp, metrics x 4], written
          # by the last launches of train(); train() hands out a COPY
          out=torch.zeros(7, dtype=torch.float32, device=dev))
      st['critic'].coef.copy_(torch.tensor([-1.0 / B, 1.0 / B, 1.0]))
      st['critic'].bias_coef.copy_(torch.tensor([-1.0 / B, 1.0 / B, 0.0]))
      st['critic'].build_jvp(2, st['coef_gp'])
      st['gen'].coef.copy_(torch.tensor([-1.0 / B]))
      st['gen'].bias_coef.zero_()
      if self.dis_optimizer.loss_scale is not None:
        # mixed_float16: the seeds of the backward chains carry the loss scale
        # (get_scaled_loss, wgan_gp.py:32,76).  The x^ segment's chain is the
        # INNER gradient of the penalty (its own tape, wgan_gp.py:45-48): its
        # seed stays 1, the scale enters its second backward through v
        st['coef_base'] = dict(
            critic=(st['critic'].coef.clone(), st['critic'].bias_coef.clone(),
                    torch.tensor([1.0, 1.0, 0.0], device=dev)),
            gen=(st['gen'].coef.clone(), st['gen'].bias_coef.clone(),
                 torch.tensor([1.0], device=dev)))
      # (fp16: the loss scale multiplies coef on the device afterwards -- three
      # launches as before)
      if _FUSE_GP and self.dis_optimizer.loss_scale is None:
        st['critic'].defer_norm()
      self._state[B] = st
    return st

  def _critic_plan(self, st, k):
    """The critic plan whose layer-1 launches read input buffer k of the step
    (_DisWorkspace.x0): plan 0 is st['critic']; the others are built on first use
    with the same seeds and tangent chain."""
    if k == 0:
      return st['critic']
    plans = st.setdefault('critic_alt', {})
    pl = plans.get(k)
    if pl is None:
      B = st['coef_gp'].shape[0]
      pl = plans[k] = st['dws'].plan(
          3 * B, B, 2 * B, x0_index=k,
          shifts=st['stage_dev'][12 * k:12 * k + 12].view(4, 3))
      pl.coef.copy_(st['critic'].coef)
      pl.bias_coef.copy_(st['critic'].bias_coef)
      pl.build_jvp(2, st['coef_gp'])
      if st['critic'].norm_deferred:
        pl.defer_norm()
    return pl

  def _can_fuse_interp(self, B, n, gws=None):
    """cg_dense_rows_interp applies: the streaming output Dense (register form at
    pitch 128, LDS-panel form beyond), n <= 8 updates.  gws: the generator
    workspace of the pass (default: the forward-only one over n * B samples)."""
    if not _FUSE_INTERP:
      return False
    if gws is None:
      gws = self.generator.net.workspace(n * B, forward_only=True)
    lay = self.discriminator.net.layers[0]
    return gws.can_interp(n) and lay.cinp == self.generator.net.Cp

  def batch_buffer(self, B):
    """The device buffer train() reads a batch of B samples from when it replays
    its hipGraph: (B,) + signal_shape, f32.  A data loader that gathers every
    batch INTO it (torch.index_select(..., out=buffer)) saves the copy train()
    otherwise makes in front of each replay; passing any other tensor stays
    valid.  One buffer per batch size, alive as long as this object."""
    st = self._get_state(B)
    g = st.get('graph')
    if g is not None:
      return g['real']
    if st.get('batch_buf') is None:
      st['batch_buf'] = torch.empty((B,) + self.signal_shape,
                                    dtype=torch.float32, device=self.device)
    return st['batch_buf']

  def _scale_seeds(self, st, which, optimizer, plan=None):
    """coef = base * (S where the segment's loss term is scaled, else 1), on the
    plan the coming launches use (default: st[which])."""
    S = optimizer.loss_scale
    if S is None:
      return
    plan = st[which] if plan is None else plan
    coef, bias_coef, scaled = st['coef_base'][which]
    f = scaled * S + (1.0 - scaled)
    torch.mul(coef, f, out=plan.coef)
    torch.mul(bias_coef, f, out=plan.bias_coef)

  # -- losses (API parity; the fused kernels compute the same values) ---------
  def generator_loss(self, fake_output):
    """wgan_gp.py:19-20."""
    return -fake_output.mean()

  def _critic_forward(self, st, real, z, alpha, shifts, slot,
                      real_cached=False, fake=None, training=True, scale=None,
                      plan=None, packed=False):
    """Steps 1-5 of the critic schedule; leaves g in st['critic'].gin.  `fake`
    is G(z) when the generator forward already ran (_critic_generate).
    training=False (validate, gan.py:87-90): BatchNormalization layers use their
    moving statistics."""
    net_d = self.discriminator.net
    B = real.shape[0]
    lay = net_d.layers[0]
    plan = st['critic'] if plan is None else plan
    s = nets._stream()
    # (host-drawn shifts of an eager step are a pageable temporary: a
    # non-blocking copy could read it after it is gone once the host runs ahead
    # of the GPU; the graph path hands a device view)
    if shifts.data_ptr() != plan.shifts.data_ptr():  # (else staged in place)
      plan.shifts.copy_(shifts, non_blocking=shifts.is_cuda)
    self._scale_seeds(st, 'critic', self.dis_optimizer, plan)
    if not packed:  # (packed: plan.x0 already holds [real | fake | x^])
      if fake is None:
        fake = st['gws'].forward(z, keep=False, training=training)
      # (a plan that mixes layer 1 of x^ never reads x^: [real | fake] only)
      _lib.call('cg_interp_pack', nets._p(real), nets._p(fake),
                nets._p(None if plan.mixes_layer1 else alpha),
                nets._p(plan.x0), B, lay.lin, lay.cin, lay.cin,
                self.generator.net.Cf, lay.cinp, 0 if real_cached else 1, s)
    plan.forward(seed_backward=True,
                 mix=alpha.reshape(-1) if plan.mixes_layer1 else None)
    plan.backward_chain(seeded=True)
    n = lay.lin * lay.cinp
    if plan.norm_deferred:
      # slots of ||g||^2 -> norm, gp, coef, critic loss and (scale = (g, dst, n
      # per sample): the caller's v = coef_b * g pass) in one launch
      norm = plan.sumsq
      slots, P = plan.ssq
      g, dst, ns = scale if scale is not None else (None, None, 0)
      _lib.call('cg_gp_loss_scale', nets._p(slots), P, nets._p(norm),
                nets._p(st['gp'][slot:]), nets._p(st['coef_gp']),
                nets._p(st['dws'].d_out), nets._p(st['loss'][slot]), B,
                self.penalty, 1.0, nets._p(g), nets._p(dst), ns, s)
      st['norm_out'] = norm
      return fake
    if plan.sumsq is not None:  # ||g||^2 came out of the dgrad epilogue
      norm = plan.sumsq
      squared = 1
    else:
      norm = st['norm']
      squared = 0
      _lib.call('cg_rownorm', nets._p(plan.gin), nets._p(norm), B, n,
                nets._p(nets.reduce_ws(self.device)), s)
    # gp = mean((||g|| - 1)^2), v's per-sample factor and the critic loss: one
    # launch (two single-block reductions over the batch)
    _lib.call('cg_gp_critic_loss', nets._p(norm), nets._p(st['gp'][slot:]),
              nets._p(st['coef_gp']), nets._p(st['dws'].d_out),
              nets._p(st['loss'][slot]), B, self.penalty, squared, 1.0, s)
    if self.dis_optimizer.loss_scale is not None:
      st['coef_gp'].mul_(self.dis_optimizer.loss_scale)  # d(S * lambda * gp)/dg
    st['norm_out'] = norm
    return fake

  # -- the step, cut at the all-reduce points ---------------------------------
  def _critic_generate(self, real, r=None, keep=False):
    """fake = G(z) of one critic update (wgan_gp.py:65-67).  It reads no
    discriminator state, so train() runs it while the previous update's
    gradient all-reduce is still in flight.  keep=True (the generator update's
    own forward): activations stay for GeneratorNet's backward."""
    B = real.shape[0]
    st = self._get_state(B)
    if r is None or 'shifts_dev' in r:
      z = self.get_noise(B)
    else:
      z = self._to_device(r['z'])
    return st['gws'].forward(z, keep=keep)

  def _critic_generate_all(self, real, z, n, alphas=None):
    """G(z) of ALL n critic updates of one train() as one forward-only pass
    over n * B samples: the generator does not change between them
    (wgan_gp.py:84-90 updates only the critic), and the small early layers run
    far better at 5 B.  Returns one (B, L, Cf) view per update -- or, with
    alphas (f32, n * B: the interpolation factors of all updates), a list of
    None: the output Dense then writes [real | fake_k | x^_k] into update k's
    input buffer itself (cg_dense_rows_interp; no f32 fake batch)."""
    B = real.shape[0]
    ws = self.generator.net.workspace(n * B, forward_only=True)
    if alphas is not None:
      st = self._get_state(B)
      # (plans that form layer 1 of x^ from the other two segments' never read x^)
      ws.forward(z, keep=False,
                 interp=(real, None if st['critic'].mixes_layer1 else alphas,
                         [st['dws'].x0(k) for k in range(n)]))
      return [None] * n
    fake = ws.forward(z, keep=False)
    return [fake[i * B:(i + 1) * B] for i in range(n)]

  def _critic_compute(self, real, r=None, slot=0, real_cached=False,
                      fake=None, alpha=None, x0_index=None):
    """wgan_gp.py:64-80 up to (not including) the optimizer update: leaves the
    critic gradients in discriminator.net.params.grad.  alpha: this update's
    interpolation draws when the caller drew all updates' at once."""
    B = real.shape[0]
    st = self._get_state(B)
    net_d = self.discriminator.net
    lay = net_d.layers[0]
    packed = x0_index is not None  # (the step's generator pass packed X0_k)
    if fake is None and not packed:
      fake = self._critic_generate(real, r)
    if r is None:
      alpha = self._streams.alpha(B) if alpha is None else alpha
      shifts = self._streams.shifts(3)
    elif 'shifts_dev' in r:  # graph replay: draws staged in device memory
      alpha = self._streams.alpha(B) if alpha is None else alpha
      shifts = r['shifts_dev']
    else:
      alpha = self._to_device(r['alpha'])
      shifts = torch.stack([
          torch.as_tensor(r['shifts_real'], dtype=torch.int32),
          torch.as_tensor(r['shifts_fake'], dtype=torch.int32),
          torch.as_tensor(r['shifts_inter'], dtype=torch.int32)
      ], dim=1)
    plan = self._critic_plan(st, x0_index or 0)
    n = lay.lin * lay.cinp
    if plan.jvp_folds:
      # g already sits over the x^ segment of X0; v = lambda * dgp/dg = coef_b * g
      # enters the tangent chain as a per-sample scale of its first launch and
      # the layer-1 weight gradient through delta_1's x^ segment (g (x) coef
      # delta == coef g (x) delta): a pass over 1/4 of the bytes
      d1 = st['dws'].delta[1][2 * B:3 * B]
      scale = (d1, d1, d1[0].numel())
    else:
      # v = lambda * dgp/dg, written over the x^ segment of X0 (in place when g
      # already sits there)
      scale = (plan.gin, plan.x0[2 * B:], n)
    self._critic_forward(st, real, None, alpha, shifts, slot, real_cached,
                         fake=fake, scale=scale, plan=plan, packed=packed)
    s = nets._stream()
    if not plan.norm_deferred:  # (else cg_gp_loss_scale has scaled the rows)
      _lib.call('cg_scale_rows', nets._p(scale[0]), nets._p(st['coef_gp']),
                nets._p(scale[1]), B, scale[2], s)
    plan.jvp_forward()
    if not nets.DETERMINISTIC:  # (the ordered reductions store every gradient)
      net_d.params.grad.zero_()
    # bias gradients: real + fake segments only (the penalty has none)
    plan.weight_grads(bias_rows=2 * B)

  def _critic_apply(self, lr_t_dev=None):
    self.dis_optimizer.update(self.discriminator, self._sync.grad_scale,
                              lr_t_dev=lr_t_dev)

  def _train_discriminator(self, inputs, r=None, slot=0, real_cached=False,
                           lr_t_dev=None):
    """wgan_gp.py:64-80."""
    real = self._to_device(inputs)
    st = self._get_state(real.shape[0])
    self._critic_compute(real, r, slot, real_cached)
    self._sync.all_reduce(self.discriminator.net.params.grad)
    self._critic_apply(lr_t_dev)
    return st['loss'][slot, 0], st['gp'][slot]

  def _gen_compute(self, real, r=None, fake=None):
    """wgan_gp.py:22-36 up to the optimizer update."""
    B = real.shape[0]
    st = self._get_state(B)
    net_g, net_d = self.generator.net, self.discriminator.net
    lay = net_d.layers[0]
    plan = st['gen']
    if fake is None:
      fake = self._critic_generate(real, r, keep=True)  # same op: fake = G(z)
    if r is None:
      shifts = self._streams.shifts(1)
    elif 'shifts_dev' in r:
      shifts = r['shifts_dev']
    else:
      shifts = torch.as_tensor(r['shifts'], dtype=torch.int32).reshape(4, 1)
    s = nets._stream()
    if shifts.data_ptr() != plan.shifts.data_ptr():
      plan.shifts.copy_(shifts, non_blocking=shifts.is_cuda)
    self._scale_seeds(st, 'gen', self.gen_optimizer)
    _lib.call('cg_cast_pad', nets._p(fake), nets._p(st['dws'].act[0]),
              B * lay.lin, lay.cin, self.generator.net.Cf, lay.cinp, s)
    plan.forward(seed_backward=True)
    _lib.call('cg_neg_mean', nets._p(st['dws'].d_out), nets._p(st['gen_loss']),
              B, s)
    plan.backward_chain(seeded=True)
    if not nets.DETERMINISTIC:
      net_g.params.grad.zero_()
    st['gws'].backward(plan.gin)

  def _gen_metrics(self, real):
    """gan.py:32-41 on the fake batch of the generator update."""
    st = self._get_state(real.shape[0])
    return self.metrics(real, st['gws'].fake, fake_pitch=self.generator.net.Cf)

  def _gen_apply(self, real, lr_t_dev=None, metrics=None):
    self.gen_optimizer.update(self.generator, self._sync.grad_scale,
                              lr_t_dev=lr_t_dev)
    return self._gen_metrics(real) if metrics is None else metrics

  def _train_generator(self, inputs, r=None, lr_t_dev=None):
    """wgan_gp.py:22-36."""
    real = self._to_device(inputs)
    st = self._get_state(real.shape[0])
    self._gen_compute(real, r)
    self._sync.all_reduce(self.generator.net.params.grad)
    metrics = self._gen_apply(real, lr_t_dev)
    return st['gen_loss'][0], metrics

  def _segments(self, real, rand=None, lr_dev=None, out=None):
    """One train() (wgan_gp.py:82-95) as launch segments cut around the
    gradient all-reduces.  Returns [(callable, flat_grad_or_None, wait)]:
    after a segment with a gradient buffer its all-reduce is STARTED; a segment
    with wait=True needs the pending all-reduce finished first.  Work that
    does not read the reduced gradients -- the next update's G(z), the signal
    metrics -- sits in wait=False segments and overlaps the collective:

      [G(z0) D-step0] ar | [G(z1)] wait [adam0 D-step1] ar | ... |
      [G(zg)] wait [adam D fwd/bwd, G bwd] ar | [metrics] wait [adam_G, outputs]

    The last callable stores the step's outputs in out['value']."""
    n = self.n_critic
    st = self._get_state(real.shape[0])
    out = {} if out is None else out
    lr = (lambda i: None) if lr_dev is None else (lambda i: lr_dev[i:])
    rc = (lambda i: None) if rand is None else (lambda i: rand['critic'][i])
    rg = None if rand is None else rand['gen']
    d_grad = self.discriminator.net.params.grad
    g_grad = self.generator.net.params.grad
    box = {}

    def generate(key, r):
      def run():
        B = real.shape[0]
        if key != 'g' and self._can_fuse_interp(B, 1, st['gws']):
          # a critic update's own generator pass (data parallel, BatchNorm,
          # CALCIUMGAN_BATCH_G=0): its output Dense writes [real | fake | x^]
          # into X0 itself (cg_dense_rows_interp, n = 1) -- no f32 fake batch, no
          # cg_interp_pack.  X0 is free: every launch of the previous update that
          # read it is ahead of this one on the stream
          drawn = r is None or 'shifts_dev' in r
          z = self.get_noise(B) if drawn else self._to_device(r['z'])
          a = (self._streams.alpha(B) if drawn
               else self._to_device(r['alpha']).reshape(-1))
          st['gws'].forward(
              z, keep=False,
              interp=(real, None if st['critic'].mixes_layer1 else a,
                      [st['dws'].x0(0)]))
          box[key] = None
          box[('alpha', key)] = a
          box[('packed', key)] = True
          return
        # only the generator update's own forward ('g') is followed by a
        # backward through G
        box[key] = self._critic_generate(real, r, keep=key == 'g')
      return run

    def critic_seg(i, own_g):
      def run():
        if i > 0:
          self._critic_apply(lr(i - 1))
        # the bf16 copy of `real` in X0[0:B] survives a critic step (only the
        # x^ segment is overwritten): converted once per train()
        self._critic_compute(real, rc(i), slot=i, real_cached=i > 0,
                             fake=None if own_g else box.pop(i),
                             alpha=box.pop(('alpha', i), None),
                             x0_index=(i if box.get('packed') else
                                       0 if box.pop(('packed', i), False)
                                       else None))
      return run

    def gen_seg():
      if n > 0:
        self._critic_apply(lr(n - 1))
      self._gen_compute(real, rg, fake=box.pop('g'))

    def metrics_seg():
      box['metrics'] = self._gen_metrics(real)

    def last_seg():
      metrics = self._gen_apply(real, lr(n), metrics=box.pop('metrics'))
      o = st['out']
      # (metrics are views of one 4-float buffer: GAN.metrics)
      _lib.call('cg_step_outputs', nets._p(st['gen_loss']), nets._p(st['loss']),
                nets._p(st['gp']), nets._p(metrics[_METRIC_KEYS[0]]), n,
                nets._p(o), nets._stream())
      out['value'] = o

    segs = []
    def first_seg():
      generate(0, rc(0))()
      critic_seg(0, False)()

    for i in range(n):
      if i == 0:
        segs.append((first_seg, d_grad, False))
      else:
        segs.append((generate(i, rc(i)), None, False))
        segs.append((critic_seg(i, False), d_grad, True))
    segs.append((generate('g', rg), None, False))
    segs.append((gen_seg, g_grad, True))
    segs.append((metrics_seg, None, False))
    segs.append((last_seg, None, True))
    if self._sync.world == 1 and not _FORCE_SPLIT:
      # no collective to cut around: the whole step is one segment (one graph)
      fns = [fn for fn, _, _ in segs]
      # (BatchNormalization: the batch statistics are those of ONE update's
      # fake batch -- no batched pass)
      if _BATCH_G and n > 1 and not self.generator.net.batch_norm:
        # ... and no all-reduce for G(z_i) to hide behind: all critic updates'
        # fake batches come from ONE generator pass at the start of the step
        def generate_all():
          B = real.shape[0]
          rs = [rc(i) for i in range(n)]
          drawn = rs[0] is None or 'shifts_dev' in rs[0]
          fuse = self._can_fuse_interp(B, n)
          # one draw each for the noise and (they are i.i.d. U[0, 1): four RNG
          # launches fewer per step) the interpolation factors of all updates
          if drawn:
            z = self.get_noise(n * B)
            a = self._streams.alpha(n * B)
          else:  # injected draws (parity tests)
            z = torch.cat([self._to_device(r['z']) for r in rs], 0)
            a = (torch.cat([self._to_device(r['alpha']).reshape(-1) for r in rs])
                 if fuse else None)
          fakes = self._critic_generate_all(real, z, n, alphas=a if fuse else None)
          box['packed'] = fuse
          for i in range(n):
            box[i] = fakes[i]
            if a is not None:
              box[('alpha', i)] = a[i * B:(i + 1) * B]
        fns = ([generate_all] + [critic_seg(i, False) for i in range(n)] +
               [generate('g', rg), gen_seg, metrics_seg, last_seg])

      def run_all():
        for fn in fns:
          fn()
      segs = [(run_all, None, False)]
    return segs, out

  def _run_segments(self, segs, launch):
    """Drive (callable, grad, wait) segments; `launch` runs one callable."""
    pending = None
    for fn, grad, wait in segs:
      if wait and pending is not None:
        pending.wait()
        pending = None
      launch(fn)
      if grad is not None:
        pending = self._sync.all_reduce_async(grad)
        if not _DP_OVERLAP and pending is not None:
          pending.wait()
          pending = None
    if pending is not None:
      pending.wait()

  def _outputs(self, o):
    """(gen_loss, dis_loss, gradient_penalty, metrics) as views of a fresh
    COPY of the step's output buffer: the buffer itself is rewritten by the next
    train() -- in place, when the step replays as a graph -- so callers may keep
    the returned tensors across steps without a host sync (main.py averages
    them at the end of the epoch).  Data parallel: averaged over the ranks, one
    7-float all-reduce per step."""
    o = self._sync.mean_scalars(o.clone())
    return (o[0], o[1], o[2],
            {k: o[3 + i] for i, k in enumerate(_METRIC_KEYS)})

  def _train_body(self, real, rand=None):
    segs, out = self._segments(real, rand)
    self._run_segments(segs, lambda fn: fn())
    return self._outputs(out['value'])

  def _capture(self, real, st):
    """Capture one train() as hipGraphs, one per segment (RCCL all-reduces stay
    eager between replays, overlapped with the wait=False segments).
    Host-drawn inputs of a replay (phase shifts, Adam step sizes) are copied
    to fixed device buffers EAGERLY ahead of the replay, from a ring of pinned
    staging slots (_stage_host_inputs); z / alpha come from the
    graph-registered device generator."""
    dev = self.device
    n = self.n_critic
    g = dict(
        # (the caller's own buffer when it gathers its batches into
        # batch_buffer(): no copy in front of a replay then)
        real=(st['batch_buf'] if st.get('batch_buf') is not None and
              st['batch_buf'].shape == real.shape else torch.empty_like(real)),
        # one staging word array per slot: [shifts int32 x (12 n + 4) |
        # lr_t f32 x (n + 1)] (the f32 part travels as its bit pattern)
        stage_host=[torch.zeros(n * 13 + 5, dtype=torch.int32).pin_memory()
                    for _ in range(_STAGING_SLOTS)],
        stage_event=[None] * _STAGING_SLOTS,
        stage_next=0,
        stage_dev=st['stage_dev'])
    g['shifts_dev'] = g['stage_dev'][:n * 12 + 4]
    g['lr_dev'] = g['stage_dev'][n * 12 + 4:].view(torch.float32)
    if g['real'].data_ptr() != real.data_ptr():
      g['real'].copy_(real)
    rand = dict(
        critic=[dict(shifts_dev=g['shifts_dev'][12 * i:12 * i + 12].view(4, 3))
                for i in range(n)],
        gen=dict(shifts_dev=g['shifts_dev'][12 * n:].view(4, 1)))
    segs, out = self._segments(g['real'], rand, g['lr_dev'])
    it_d, it_g = (self.dis_optimizer.host_steps, self.gen_optimizer.host_steps)
    graphs = []
    pool = None
    torch.cuda.synchronize()
    try:
      for k, (fn, grad, wait) in enumerate(segs):
        graph = torch.cuda.CUDAGraph()
        graph.register_generator_state(self._streams.local)
        # thread_local: the RCCL watchdog thread may touch the HIP runtime
        # while this thread captures
        with torch.cuda.graph(graph, pool=pool,
                              capture_error_mode='thread_local'):
          fn()
        pool = graph.pool()
        graphs.append((graph.replay, grad, wait))
    finally:
      # capture only records: undo the host-side step counters it advanced
      self.dis_optimizer.host_steps, self.gen_optimizer.host_steps = it_d, it_g
    g['graphs'] = graphs
    g['out'] = out['value']
    return g

  def _train_graphed(self, real, st):
    g = st.get('graph')
    if g is None:
      try:
        g = st['graph'] = self._capture(real, st)
      except Exception as e:  # noqa: BLE001 -- any capture failure
        # the step itself is unaffected: keep training with eager launches
        import warnings
        warnings.warn('calciumgan_amd: hipGraph capture of train() failed '
                      '({}: {}); continuing with eager launches'.format(
                          type(e).__name__, e))
        self._use_graph = False
        torch.cuda.synchronize()
        return self._train_body(real)
    n = self.n_critic
    # the graphs read their batch from a fixed buffer.  A caller that gathers
    # its batches into batch_buffer() wrote it already; any other tensor is
    # copied (107 MB at cfg2, ~35 us; 4.3 GB at cfg5, 2 ms)
    if g['real'].data_ptr() != real.data_ptr():
      g['real'].copy_(real)
    self._stage_host_inputs(g)
    self._run_segments(g['graphs'], lambda replay: replay())
    self.dis_optimizer.host_steps += n
    self.gen_optimizer.host_steps += 1
    return self._outputs(g['out'])

  def _stage_host_inputs(self, g):
    """Phase shifts and Adam step sizes of the coming replay -> device.  The
    host writes them into the next pinned slot of a ring and enqueues the copy
    on the launch stream (ordered after the previous replay, which still reads
    the device buffer); an event per slot keeps the host from rewriting a slot
    whose copy has not executed yet -- train() never syncs, so the host may
    run several steps ahead of the GPU."""
    n = self.n_critic
    k = g['stage_next']
    g['stage_next'] = (k + 1) % _STAGING_SLOTS
    if g['stage_event'][k] is not None:
      g['stage_event'][k].synchronize()
    host = g['stage_host'][k]
    for i in range(n):
      host[12 * i:12 * i + 12] = self._streams.shifts(3).reshape(-1)
    host[12 * n:12 * n + 4] = self._streams.shifts(1).reshape(-1)
    lr = host[12 * n + 4:].view(torch.float32)
    for i in range(n):
      lr[i] = self.dis_optimizer.lr_t(self.dis_optimizer.host_steps + i + 1)
    lr[n] = self.gen_optimizer.lr_t(self.gen_optimizer.host_steps + 1)
    g['stage_dev'].copy_(host, non_blocking=True)
    ev = g['stage_event'][k] = torch.cuda.Event()
    ev.record()

  def train(self, inputs, rand=None):
    """wgan_gp.py:82-95: n_critic critic updates on the SAME batch, then one
    generator update.  Returns (gen_loss, dis_loss, gradient_penalty, metrics)
    as 0-d device tensors (no host sync inside; each call returns views of its
    own small buffer, so they stay valid across later steps; under data
    parallelism they are the means over all ranks).  `rand` optionally injects
    the random draws (same structure as oracle.draw_randomness) for parity
    tests.  After two eager calls per batch size the step replays as
    hipGraphs."""
    _lib.use(self.precision)
    real = self._to_device(inputs)
    st = self._get_state(real.shape[0])
    if rand is None and self._use_graph:
      st['calls'] = st.get('calls', 0) + 1
      if st['calls'] > _GRAPH_WARMUP_CALLS:
        return self._train_graphed(real, st)
    # an eager step between replays (injected randomness, main.py's --profile
    # window).  The captured graphs stay valid: they hold pointers to buffers
    # that live as long as this object, and nothing in them depends on what ran
    # in between.  (Round 2 dropped them here after "stale graph" penalties of
    # 1e25; the cause was a hipMemsetAsync NODE inside the captured step --
    # cg_rownorm's -- not stale memory: DESIGN.md section 8.)
    return self._train_body(real, rand)

  def validate(self, inputs, rand=None):
    """gan.py:87-90 / :58-70 with the WGAN-GP loss (inner-gradient penalty, no
    parameter update).  Returns (fake, gen_loss, dis_loss, gp, metrics)."""
    _lib.use(self.precision)
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
    fake = self._critic_forward(st, real, z, alpha, shifts, 0, training=False)
    C = self.generator.net.C
    metrics = self.metrics(real, fake, fake_pitch=self.generator.net.Cf)
    loss = st['loss'][0]
    gen_loss, dis_loss, gp, metrics = self._outputs(
        torch.stack([loss[1], loss[0], st['gp'][0]] +
                    [metrics[k] for k in _METRIC_KEYS]))
    return fake[:, :, :C].clone(), gen_loss, dis_loss, gp, metrics
