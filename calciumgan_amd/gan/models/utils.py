"""gan/models/utils.py counterparts that the hot path needs."""
import numpy as np


def count_trainable_params(model):
  """gan/models/utils.py:11-14."""
  return int(np.sum([int(np.prod(v.shape)) for v in model.trainable_variables]))
