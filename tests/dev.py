"""Test-side helpers for the GPU legs: device arrays of the product (spartan_amd.devarray) and -- where a test
wants an INDEPENDENT calculator on the device at sizes NumPy cannot finish in seconds -- a zero-copy torch view of
the same HBM bytes (`T`).  torch is test infrastructure here; the product never imports it."""
import numpy as np

from spartan_amd import devarray as D
from spartan_amd import kernels


def T(d):
  """A torch tensor aliasing the device array (through __cuda_array_interface__): reads see the product's bytes,
  writes plant values into the tile."""
  import torch
  return torch.as_tensor(d, device='cuda')


def uniform_tile(shape, seed, lo=0.0, hi=1.0, dtype=np.float32):
  """A device array of uniform [lo, hi) values from the library's counter-based generator."""
  out = D.empty(tuple(shape), dtype)
  kernels.random_fill(out, 'uniform', seed, 0)
  if (lo, hi) != (0.0, 1.0):
    out = out * np.dtype(dtype).type(hi - lo) + np.dtype(dtype).type(lo)
  return out


def uniform(sp, shape, seed, lo=0.0, hi=1.0, **kw):
  """A distributed array whose tiles are uniform_tile()s (tile at row r seeded with seed + r)."""
  return sp.from_tile_fn(shape, np.float32, lambda ex: uniform_tile(ex.shape, seed + 1000003 * (ex.ul[0] if ex.ul else 0), lo, hi), **kw)
