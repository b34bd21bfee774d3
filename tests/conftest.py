import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)


def pytest_configure(config):
  config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


def pytest_collection_modifyitems(config, items):
  """GPU tests are skipped (not failed) when no device is visible and the run
  did not ask for them explicitly with ``-m gpu``."""
  import torch
  if torch.cuda.is_available():
    return
  skip = pytest.mark.skip(reason='no GPU visible')
  for item in items:
    if 'gpu' in item.keywords:
      item.add_marker(skip)


@pytest.fixture
def fixed_tiles():
  """Static tile choices for the duration of a test (no per-process tuning, no
  earlier test's cached choices): with the ordered reductions the HIP path is
  then reproducible bit for bit, so a bar can sit just above ONE measured value
  instead of above a run-to-run spread."""
  from calciumgan_amd import nets
  saved = (nets._AUTOTUNE, dict(nets._TILE_CACHE))
  nets._AUTOTUNE = False
  nets._TILE_CACHE.clear()
  yield
  nets._AUTOTUNE = saved[0]
  nets._TILE_CACHE.clear()
  nets._TILE_CACHE.update(saved[1])
