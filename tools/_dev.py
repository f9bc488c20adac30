"""Shared by the tools: device operands from the library's generator and launch timing by HIP events."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spartan_amd import devarray as D  # noqa: E402
from spartan_amd import kernels  # noqa: E402


def rand(shape, lo=0.0, hi=1.0, dtype=np.float32, seed=1):
  out = D.empty(tuple(shape), dtype)
  kernels.random_fill(out, 'uniform', seed, 0)
  if (lo, hi) != (0.0, 1.0):
    out = out * np.dtype(dtype).type(hi - lo) + np.dtype(dtype).type(lo)
  return out


def timeit(fn, reps, warmup=3):
  """Average milliseconds per call of fn()."""
  for _ in range(warmup):
    fn()
  D.synchronize()
  e0, e1 = D.Event(), D.Event()
  e0.record()
  for _ in range(reps):
    fn()
  e1.record()
  e1.synchronize()
  return e0.elapsed_ms(e1) / reps
