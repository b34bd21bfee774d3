"""Model registry -- same contract as gan/models/registry.py:6-33."""
from .utils import count_trainable_params

_MODELS = dict()


def register(name):

  def add_to_dict(fn):
    _MODELS[name] = fn
    return fn

  return add_to_dict


def get_models(hparams, summary=None):
  """registry.py:16-33: unknown names print and exit, parameter counts are
  logged under model/trainable_parameters/*."""
  if hparams.model not in _MODELS:
    print('models {} not found'.format(hparams.model))
    exit()

  generator, discriminator = _MODELS[hparams.model](hparams)

  if summary is not None:
    summary.scalar('model/trainable_parameters/generator',
                   count_trainable_params(generator))
    summary.scalar('model/trainable_parameters/discriminator',
                   count_trainable_params(discriminator))

  if getattr(hparams, 'verbose', 0):
    generator.summary()
    print('')
    discriminator.summary()

  return generator, discriminator
