"""stencil / maxpool / _convolve (spartan/expr/operator/stencil.py; reference tests/test_stencil.py only times them).
Expected values are the reference's loops written out in NumPy; integer-valued data, so the GEMM formulation on the
GPU must reproduce them exactly."""
import numpy as np
import pytest

import spartan_amd as sp


def _conv_ref(img, flt):
  n, c, w, h = img.shape
  f, _, fw, fh = flt.shape
  out = np.zeros((n, f, w, h), dtype=np.result_type(img.dtype, flt.dtype))
  for x in range(w):
    for y in range(h):
      for i in range(fw):
        for j in range(fh):
          if x + i < w and y + j < h:
            out[:, :, x, y] += np.einsum('nc,fc->nf', img[:, :, x + i, y + j], flt[:, :, i, j])
  return out


def _pool_ref(x, pool, stride):
  n, c, w, h = x.shape
  W, H = -(-w // stride), -(-h // stride)
  out = np.full((n, c, W, H), -1e12, dtype=x.dtype)
  for a in range(w):
    for b in range(h):
      X, Y = a // stride, b // stride
      if a - X * stride < pool and b - Y * stride < pool:
        out[:, :, X, Y] = np.maximum(out[:, :, X, Y], x[:, :, a, b])
  return out


def _check(backend_factory, workers):
  rng = np.random.RandomState(workers)
  sp.initialize(backend=backend_factory(), num_workers=workers)
  try:
    for dtype in (np.float32, np.float64):
      img = rng.randint(-3, 4, size=(8, 3, 12, 12)).astype(dtype)
      flt = rng.randint(-2, 3, size=(5, 3, 4, 4)).astype(dtype)
      want = _conv_ref(img, flt)
      # the tile body on its own (tests/test_stencil.py:test_local_convolve passes NumPy arrays)
      got = sp.get_context().backend.to_numpy(sp._convolve(img, flt))
      np.testing.assert_array_equal(got, want)
      # the operator: images tiled over the spatial dimensions like tests/test_stencil.py:test_stencil
      side = 12 if workers == 1 else 6
      images = sp.from_numpy(img, tile_hint=(8, 3, side, side))
      filters = sp.from_numpy(flt, tile_hint=(5, 3, 4, 4))
      res = sp.stencil(images, filters, 1).force()
      assert res.shape == (8, 5, 12, 12) and res.dtype == np.dtype(dtype)
      if workers == 1:
        np.testing.assert_array_equal(res.glom(), want)
      else:
        # every tile convolves only its own pixels (the reference does not exchange halos, stencil.py:76-100)
        tiles = np.zeros_like(want)
        for a in range(0, 12, side):
          for b in range(0, 12, side):
            tiles[:, :, a:a + side, b:b + side] = _conv_ref(img[:, :, a:a + side, b:b + side], flt)
        np.testing.assert_array_equal(res.glom(), tiles)
      x = rng.randint(-9, 10, size=(4, 2, 10, 7)).astype(dtype)
      for pool, stride in ((2, 2), (3, 2), (2, 3)):
        p = sp.maxpool(sp.from_numpy(x, tile_hint=(4, 2, 10, 7)), pool, stride).force()
        np.testing.assert_array_equal(p.glom(), _pool_ref(x, pool, stride))
  finally:
    sp.shutdown()


@pytest.mark.parametrize('workers', [1, 4])
def test_stencil_cpu(workers):
  from oracle.np_backend import NumpyBackend
  _check(NumpyBackend, workers)


@pytest.mark.gpu
@pytest.mark.parametrize('workers', [1, 4])
def test_stencil_gpu(workers):
  from spartan_amd.backend_hip import HipBackend
  _check(HipBackend, workers)
