"""Algorithm registry -- same contract as gan/algorithms/registry.py:4-19."""
_ALGORITHMS = dict()


def register(name):

  def add_to_dict(fn):
    _ALGORITHMS[name] = fn
    return fn

  return add_to_dict


def get_algorithm(hparams, generator, discriminator, summary=None):
  if hparams.algorithm not in _ALGORITHMS:
    print('Algorithm {} not found'.format(hparams.algorithm))
    exit()
  return _ALGORITHMS[hparams.algorithm](hparams, generator, discriminator,
                                        summary)
