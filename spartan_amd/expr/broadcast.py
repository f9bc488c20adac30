"""NumPy-style broadcasting of DistArrays without copying
(mirror of the reference's spartan/expr/operator/broadcast.py)."""
import numpy as np

from .. import context
from ..array import distarray, extent
from ..util import Assert


def _broadcast_invoke(self, tile_id, blob, mapper_fn, kw):
  """broadcast.py:11-26 (_broadcast_mapper): map a base tile to its broadcast extent."""
  array = self
  base_ex = array.base.extent_for_blob(tile_id)
  ul = [0 for _ in array.shape]
  lr = [dim for dim in array.shape]
  for i in range(len(base_ex.ul) - 1, -1, -1):
    broadcast_i = i + array.prepend_dim
    if array.base.shape[i] != array.shape[broadcast_i]:
      assert ul[broadcast_i] == 0
    else:
      ul[broadcast_i] = base_ex.ul[i]
      lr[broadcast_i] = base_ex.lr[i]
  ex = extent.create(ul, lr, array.shape)
  return mapper_fn(ex, **kw)


class Broadcast(distarray.DistArray):
  """broadcast.py:28-109."""

  def __init__(self, base, shape):
    Assert.isinstance(base, (np.ndarray, distarray.DistArray))
    Assert.isinstance(shape, tuple)
    if isinstance(base, Broadcast):
      self.base = base.base
    else:
      self.base = base
    self.shape = shape
    self.tiles = self.base.tiles
    self.dtype = base.dtype
    self.sparse = self.base.sparse
    self.bad_tiles = []
    self.prepend_dim = len(shape) - len(base.shape)

  def __repr__(self):
    return 'Broadcast(%s -> %s)' % (self.base, self.shape)

  def real_size(self):
    """broadcast.py:54-59: offset by one to prefer direct arrays."""
    return int(np.prod(self.base.shape, dtype=np.int64)) - 1

  def extent_for_blob(self, tile_id):
    return self.base.extent_for_blob(tile_id)

  _invoke_mapper = _broadcast_invoke

  def foreach_tile(self, mapper_fn, kw=None):
    if kw is None:
      kw = {}
    return distarray.run_kernel(self, list(self.base.tiles.values()), mapper_fn, kw)

  def _base_ex(self, ex):
    """broadcast.py:72-91."""
    while len(ex.shape) > len(self.base.shape):
      ex = extent.drop_axis(ex, 0)
    ul, lr = [], []
    for i in range(len(self.base.shape)):
      size = self.base.shape[i]
      if size == 1:
        ul.append(0)
        lr.append(1)
      else:
        ul.append(ex.ul[i])
        lr.append(ex.lr[i])
    return extent.create(ul, lr, self.base.shape)

  def fetch(self, ex):
    """broadcast.py:93-104.  The slab is returned un-broadcast with size-1 axes;
    kernels broadcast through zero strides (no copy is ever made)."""
    return self.fetch_base_tile(ex)

  def fetch_base_tile(self, ex):
    """broadcast.py:106-109."""
    ex = self._base_ex(ex)
    return self.base.fetch(ex)


def broadcast(args):
  """broadcast.py:111-158."""
  if len(args) == 1:
    return args
  orig_shapes = [list(x.shape) for x in args]
  dims = [len(shape) for shape in orig_shapes]
  max_dim = max(dims)
  new_shapes = []
  for i in range(len(orig_shapes)):
    diff = max_dim - len(orig_shapes[i])
    new_shapes.append([1] * diff + orig_shapes[i])
  for axis in range(max_dim):
    axis_shape = set(shp[axis] for shp in new_shapes)
    assert len(axis_shape) <= 2, 'Mismatched shapes for broadcast: %s' % orig_shapes
    if len(axis_shape) == 2:
      assert 1 in axis_shape, 'Mismatched shapes for broadcast: %s' % orig_shapes
    max_size = max(shp[axis] for shp in new_shapes)
    for shp in new_shapes:
      shp[axis] = max_size
  results = []
  for i in range(len(args)):
    if new_shapes[i] == orig_shapes[i]:
      results.append(args[i])
    else:
      results.append(Broadcast(args[i], tuple(new_shapes[i])))
  return results
