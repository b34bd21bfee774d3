"""Optimizer wrapper -- gan/algorithms/optimizer.py:5-34.

Keras Adam (lr from hparams, beta1 .9, beta2 .999, eps 1e-7 outside the bias
correction) as one fused HIP launch over the model's flat parameter buffer.
The bf16 path needs no loss scaling, so get_scaled_loss / unscale are the
identity (the reference only scales under mixed_float16).
"""
from ... import nets


class Optimizer(object):

  def __init__(self, hparams):
    self.learning_rate = hparams.learning_rate
    self._iterations = 0

  @property
  def iterations(self):
    return self._iterations

  @iterations.setter
  def iterations(self, value):
    self._iterations = int(value)

  def get_scaled_loss(self, loss):
    return loss

  def get_unscaled_gradients(self, scaled_gradients):
    return scaled_gradients

  def lr_t(self, step):
    return nets.adam_lr_t(step, self.learning_rate)

  def update(self, model, grad_scale=1.0, lr_t_dev=None):
    """Apply the gradients already accumulated in model.net.params.grad
    (optimizer.py:31-34) and refresh the packed bf16 operands."""
    self._iterations += 1
    nets.adam_update(model.net.params, self._iterations, self.learning_rate,
                     grad_scale, lr_t_dev=lr_t_dev)
    model.net.repack()
