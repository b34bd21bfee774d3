"""Optimizer wrapper -- gan/algorithms/optimizer.py:5-34.

Keras Adam (lr from hparams, beta1 .9, beta2 .999, eps 1e-7 outside the bias
correction) as one fused HIP launch over the model's flat parameter buffer.
With hparams.mixed_precision (fp16 activations) it is wrapped in the
reference's dynamic loss scaling (LossScaleOptimizer(Adam, 'dynamic'),
optimizer.py:10-12): the caller multiplies its loss -- here: the seeds of the
hand-scheduled backward chains -- by `loss_scale`, update() checks the scaled
gradients, divides them by the scale inside the Adam launch or skips the update,
and advances the scale; all of it on the device.  The bf16 default needs no
scaling (f32 exponent range) and get_scaled_loss / get_unscaled_gradients are
the identity there, as in the reference without the policy.
"""
from ... import nets


class Optimizer(object):

  def __init__(self, hparams, device=None):
    self.learning_rate = hparams.learning_rate
    self._mixed_precision = bool(getattr(hparams, 'mixed_precision', False))
    self._iterations = 0
    # [S, finite updates in a row, applied steps, finite flag] on the device
    self.loss_scale_state = (nets.new_loss_scale_state(device)
                             if self._mixed_precision else None)

  @property
  def iterations(self):
    """Applied Adam steps (a skipped non-finite update does not count:
    LossScaleOptimizer never reaches the inner optimizer then).  Reading it
    under mixed precision syncs with the device."""
    if self._mixed_precision:
      return int(self.loss_scale_state[2].item())
    return self._iterations

  @iterations.setter
  def iterations(self, value):
    self._iterations = int(value)
    if self._mixed_precision:
      self.loss_scale_state[2] = float(value)

  @property
  def host_steps(self):
    """update() calls counted on the host (the step count of the bf16 path's
    host-computed Adam step size; under mixed precision the device counts the
    APPLIED steps itself and this number is informational)."""
    return self._iterations

  @host_steps.setter
  def host_steps(self, value):
    self._iterations = int(value)

  @property
  def loss_scale(self):
    """Device scalar (1-element view) or None without mixed precision."""
    return None if not self._mixed_precision else self.loss_scale_state[0:1]

  def get_scaled_loss(self, loss):
    return loss * self.loss_scale if self._mixed_precision else loss

  def get_unscaled_gradients(self, scaled_gradients):
    if not self._mixed_precision:
      return scaled_gradients
    return [g / self.loss_scale for g in scaled_gradients]

  def lr_t(self, step):
    return nets.adam_lr_t(step, self.learning_rate)

  def update(self, model, grad_scale=1.0, lr_t_dev=None):
    """Apply the gradients already accumulated in model.net.params.grad
    (optimizer.py:31-34) and refresh the packed MFMA operands."""
    self._iterations += 1
    if self._mixed_precision:
      nets.adam_update_scaled(model.net.params, self.learning_rate,
                              self.loss_scale_state, grad_scale)
    else:
      nets.adam_update(model.net.params, self._iterations, self.learning_rate,
                       grad_scale, lr_t_dev=lr_t_dev)
    model.net.repack()
