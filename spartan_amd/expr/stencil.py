"""stencil (convolution) and maxpool: mirror of the reference's spartan/expr/operator/stencil.py.

`_convolve` is the tile body: out[n, f, x, y] = sum over c, i, j (x+i < w, y+j < h) of
image[n, c, x+i, y+j] * filter[f, c, i, j] (stencil.py:29-45, a Parakeet `imap` over a scalar Python function in the
reference).  Here it is a GEMM: the shifted image planes are laid side by side (one strided box copy per (c, i, j),
zero beyond the edge) into P[(n, x, y), (c, i, j)], multiplied with the filters reshaped to [(c, i, j), f] on the MFMA
GEMM, and the product is brought back to [n, f, x, y].

`maxpool`: the reference's tile body is disabled by a Parakeet workaround (it writes only target[0, 0, 0, 0],
stencil.py:54-73); the loop it disables is implemented: pixel (a, b) goes to window (a // stride, b // stride) when
its offset inside the window is below pool_size, out = the maximum per window, -1e12 where a window is empty.
"""
import math

import numpy as np

from .base import eager
from .ndarray import ndarray
from .shuffle import shuffle
from .. import context
from ..array import distarray, extent, tile as tile_mod
from ..util import Assert, divup


def tiles_like(array, target_shape):
  """stencil.py:18-26."""
  return [int(math.ceil(t * (float(want) / have)))
          for t, want, have in zip(array.tile_shape(), target_shape, array.shape)]


def _convolve(local_image, local_filters):
  """stencil.py:29-45 on backend tensors (NumPy arrays are uploaded)."""
  be = context.get().backend
  img = be.contiguous(be._as_device(local_image)) if hasattr(be, '_as_device') else local_image
  flt = be.contiguous(be._as_device(local_filters)) if hasattr(be, '_as_device') else local_filters
  return be.convolve(img, flt)


def stencil_mapper(array, ex, filters=None, images=None, target_shape=None):
  """stencil.py:76-100."""
  ctx = context.get()
  local_filters = filters.fetch(extent.from_shape(filters.shape))
  local_image = images.fetch(ex)
  num_img, n_col, w, h = images.shape
  num_filt, f_col, fw, fh = filters.shape
  Assert.eq(n_col, f_col)
  target_ex = extent.create((ex.ul[0], 0, ex.ul[2], ex.ul[3]), (ex.lr[0], num_filt, ex.lr[2], ex.lr[3]), target_shape)
  if ctx.executing:
    result = ctx.backend.convolve(local_image, local_filters)
  else:
    result = distarray.Absent(target_ex.shape, images.dtype)
  yield (target_ex, result)


def stencil(images, filters, stride=1):
  """stencil.py:103-132."""
  images, filters = eager(images).evaluate(), eager(filters)
  (n_img, n_col, w, h), (n_filt, f_col, fw, fh) = images.shape, filters.shape
  tile_hint = tiles_like(images, (n_img, n_filt, w, h))
  target = ndarray((n_img, n_filt, w, h), dtype=images.dtype, reduce_fn=np.add, tile_hint=tile_hint)
  return shuffle(images, stencil_mapper, target=target,
                 kw=dict(images=images, filters=filters, target_shape=target.shape))


def _maxpool_mapper(array, ex, pool_size, stride, target_shape):
  """stencil.py:135-150."""
  ctx = context.get()
  region = array.fetch(ex)
  ul, lr = ex.ul, ex.lr
  t_ul = tuple(ul[:2]) + tuple(int(v) // stride for v in ul[2:])
  t_lr = tuple(lr[:2]) + tuple(divup(tuple(lr[2:]), stride))
  target_ex = extent.create(t_ul, t_lr, target_shape)
  if ctx.executing:
    pooled = ctx.backend.maxpool(region, pool_size, stride, target_ex.shape)
  else:
    pooled = distarray.Absent(target_ex.shape, array.dtype)
  yield (target_ex, pooled)


def maxpool(images, pool_size=2, stride=2):
  """stencil.py:153-172."""
  images = images.evaluate() if hasattr(images, 'evaluate') else images
  n_img, n_col = images.shape[:2]
  tgt_shape = divup(tuple(images.shape[2:]), stride)
  tile_hint = tiles_like(images, (n_img, n_col,) + tgt_shape)
  target = ndarray((n_img, n_col) + tgt_shape, dtype=images.dtype, tile_hint=tile_hint, reduce_fn=np.maximum)
  return shuffle(images, _maxpool_mapper, target=target,
                 kw=dict(target_shape=target.shape, stride=stride, pool_size=pool_size))
