import os
import sys

import pytest

try:
  # Test infrastructure only (tests/dev.py uses torch as an independent calculator on the device; the product never
  # imports it).  Loaded FIRST on purpose: torch ships its own copy of the HIP runtime, and a process in which
  # libspartan_hip.so brought the system's runtime up before torch loads its own ends with two runtimes and torch
  # seeing no GPU.
  import torch  # noqa: F401
except ImportError:
  pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)


def pytest_configure(config):
  config.addinivalue_line('markers', 'gpu: test needs a real MI355X (run with -m gpu on the GPU box)')
  config.addinivalue_line('markers', 'host_logic: pure host arithmetic pinned by reference goldens (extents, tiling, '
                                     'fusion); runs in the CPU suite here AND in the -m gpu suite on the GPU box')


def _have_gpu():
  try:
    from spartan_amd import comm
    return comm.gpu_count() > 0
  except Exception:
    return False


@pytest.hookimpl(tryfirst=True)
def pytest_collection_modifyitems(config, items):
  if _have_gpu():
    # on the GPU box the golden-pinned host logic is part of the `-m gpu` run as well
    for item in items:
      if 'host_logic' in item.keywords:
        item.add_marker(pytest.mark.gpu)
    return
  skip = pytest.mark.skip(reason='no GPU in this container')
  for item in items:
    if 'gpu' in item.keywords:
      item.add_marker(skip)
