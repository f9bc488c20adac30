"""bincount / norm / normalize / concatenate (reference spartan/expr/statistics.py:108-228,
spartan/expr/manipulation.py:45-88).  Tile bodies run on backend (HBM) tensors."""
import builtins

import numpy as np

from . import builtins as B
from .assign import region_map  # noqa: F401  (kept importable from here as well)
from .base import Expr
from .map import map2, map_with_location
from .. import context
from ..array import distarray, extent


def _tile_op(method, out_shape, out_dtype, *tiles, **kw):
  if builtins.any(isinstance(t, distarray.Absent) for t in tiles):
    return distarray.Absent(tuple(out_shape), np.dtype(out_dtype))
  return getattr(context.get().backend, method)(*tiles, **kw)


# ------------------------------------------------------------------ bincount (statistics.py:108-140)
def _bincount_mapper(ex, tiles, minlength=None):
  """statistics.py:108-114: np.bincount(tile[, weights], minlength) -> whole-result partial."""
  if len(tiles) > 1:
    be = context.get().backend
    wdt = np.float64 if isinstance(tiles[1], distarray.Absent) else be.dtype_of(tiles[1])
    wdt = wdt if np.dtype(wdt).kind == 'f' else np.float64
    result = _tile_op('weighted_bincount', (minlength,), wdt, tiles[0], tiles[1], k=minlength)
  else:
    result = _tile_op('bincount', (minlength,), np.int64, tiles[0], k=minlength)
  yield extent.from_shape((minlength,)), result


def bincount(v, weights=None, minlength=None):
  """Count the values of a non-negative integer array (statistics.py:117-140).
  (The reference asserts `min(v) > 0`, which rejects arrays containing 0; zeros are accepted here.)"""
  minval = B.min(v).glom()
  maxval = int(B.max(v).glom())
  assert minval >= 0
  minlength = maxval + 1 if minlength is None else builtins.max(maxval + 1, int(minlength))
  if weights is not None:
    return map2((v, weights), fn=_bincount_mapper, fn_kw={'minlength': minlength}, shape=(minlength,),
                reducer=np.add, dtype=np.float64)
  return map2(v, fn=_bincount_mapper, fn_kw={'minlength': minlength}, shape=(minlength,), reducer=np.add,
              dtype=np.int64)


# ------------------------------------------------------------------ norm (statistics.py:186-221)
def norm(array, ord=2):
  """ord=1: max column sum of |a| (sum |a| for vectors); ord=2: Euclidean norm of a vector.
  Returns driver-side NumPy values like the reference (it gloms the reduction)."""
  assert ord == 1 or ord == 2
  if ord == 1:
    result = B.sum(B.abs(array), axis=0).optimized().glom()
    return np.max(result)
  elif len(array.shape) == 1 or len(array.shape) == 2 and array.shape[1] == 1:
    result = B.sum(B.square(array), axis=0).optimized().glom()
    return np.sqrt(result)
  assert False, "matrix norm-2 is not support!"


# ------------------------------------------------------------------ normalize (statistics.py:143-183)
def normalize(array, axis=None):
  """After normalisation sum(array, axis) == 1 (statistics.py:166-183).  For axis 0 / 1 the reference's
  tile body divides only the tile's first column / row (statistics.py:158-161); this divides every
  element by its column / row sum, which is what its docstring promises."""
  axis_sum = B.sum(array, axis=axis).glom()
  if axis is None:
    return array / axis_sum
  s = np.asarray(axis_sum)
  s = s.reshape((1, -1)) if axis == 0 else s.reshape((-1, 1))
  return array / B.from_numpy(s)


# ------------------------------------------------------------------ concatenate (manipulation.py:45-88)
def _concatenate_mapper(extents, tiles, shape=None, axis=0):
  """manipulation.py:45-58."""
  if len(extents[0].shape) > 1:
    ul = extents[0].ul
    lr = list(extents[0].lr)
    lr[axis] += extents[1].shape[axis]
    ex = extent.create(ul, lr, shape)
    out_shape = list(extents[0].shape)
    out_shape[axis] += extents[1].shape[axis]
    dt = np.result_type(*[t.dtype if isinstance(t, distarray.Absent) else context.get().backend.dtype_of(t)
                          for t in tiles])
    yield ex, _tile_op('concat', out_shape, dt, tiles[0], tiles[1], axis=axis)
  else:
    ex = extent.create(extents[0].ul, extents[0].lr, shape)
    yield ex, tiles[0]
    ul = (extents[0].array_shape[0] + extents[1].ul[0],)
    lr = (extents[0].array_shape[0] + extents[1].lr[0],)
    ex = extent.create(ul, lr, shape)
    yield ex, tiles[1]


def concatenate(a, b, axis=0):
  """Join two arrays along `axis` (manipulation.py:61-88)."""
  new_shape = [0] * len(a.shape)
  for index, (dim1, dim2) in enumerate(zip(a.shape, b.shape)):
    if index == axis:
      new_shape[index] = dim1 + dim2
      continue
    new_shape[index] = dim1
    if dim1 != dim2:
      raise ValueError('all the input array dimensions except for the'
                       'concatenation axis must match exactly')
  if len(a.shape) > 1:
    partition_axis = extent.largest_dim_axis(a.shape, exclude_axes=[axis])
  else:
    partition_axis = 0
  return map2((a, b), (partition_axis, partition_axis), fn=_concatenate_mapper,
              fn_kw={'axis': axis, 'shape': new_shape}, shape=new_shape)


# ------------------------------------------------------------------ diagonal / diag / diagflat (creation.py:222-330)
def _diagflat_mapper(extents, tiles, shape=None):
  """creation.py:222-246: this tile's elements, ravelled, on the diagonal of rows head..tail."""
  ex = extents[0]
  head = extent.ravelled_pos(ex.ul, ex.array_shape)
  tail = extent.ravelled_pos([l - 1 for l in ex.lr], ex.array_shape)
  rows = tail - head + 1
  dt = tiles[0].dtype if isinstance(tiles[0], distarray.Absent) else context.get().backend.dtype_of(tiles[0])
  target_ex = extent.create((head, 0), (tail + 1, shape[1]), shape)
  yield target_ex, _tile_op('diag_embed', (rows, shape[1]), dt, tiles[0], width=shape[1], col0=head)


def diagflat(array):
  """A 2-D array with the flattened input on its diagonal (creation.py:249-261)."""
  n = int(np.prod(array.shape))
  shape = (n, n)
  return map2(array, 0, fn=_diagflat_mapper, fn_kw={'shape': shape}, shape=shape)


def _diagonal_mapper(ex, tiles, shape=None):
  """creation.py:264-279: the part of the main diagonal that crosses this tile."""
  ex = ex[0] if isinstance(ex, (list, tuple)) else ex
  tile = tiles[0]
  max_dim = builtins.max(*ex.ul)
  first_point = [max_dim for _ in range(len(ex.ul))]
  slices = []
  for i in range(len(ex.ul)):
    if first_point[i] >= ex.lr[i]:
      return
    slices.append(slice(first_point[i] - ex.ul[i], ex.shape[i]))
  if len(slices) != 2:
    raise NotImplementedError('diagonal of a %d-d array' % len(slices))
  n = builtins.min(s.stop - s.start for s in slices)
  dt = tile.dtype if isinstance(tile, distarray.Absent) else context.get().backend.dtype_of(tile)
  result = _tile_op('diag_extract', (n,), dt, tile, slices=tuple(slices))
  target_ex = extent.create((first_point[0],), (first_point[0] + n,), shape)
  yield target_ex, result


def diagonal(a):
  """Main diagonal (creation.py:282-299)."""
  if len(a.shape) < 2:
    raise ValueError("diag requires an array of at least two dimensions")
  shape = (builtins.min(a.shape),)
  return map2(a, fn=_diagonal_mapper, fn_kw={'shape': shape}, shape=shape)


def diag(array, offset=0):
  """Extract a diagonal or construct a diagonal array (creation.py:302-330)."""
  if offset != 0:
    raise NotImplementedError
  if len(array.shape) == 1:
    return diagflat(array)
  elif len(array.shape) == 2:
    return diagonal(array)
  raise ValueError("Input must be 1- or 2-d.")
