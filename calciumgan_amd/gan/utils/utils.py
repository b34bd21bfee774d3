"""gan/utils/utils.py counterparts used by the hot path."""


def normalize(x, x_min, x_max):
  """utils.py:25-27."""
  return (x - x_min) / (x_max - x_min)


def denormalize(x, x_min, x_max):
  """utils.py:30-32."""
  return x * (x_max - x_min) + x_min
