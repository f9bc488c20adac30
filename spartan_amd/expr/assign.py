"""Region updates: write / assign / region_map / retile (reference
spartan/expr/operator/write_array.py:32-94, spartan/expr/assign.py, operator/region_map.py,
spartan/expr/retile.py).  Tile bodies work on backend (HBM) tensors: the box copies are
sp_slice_copy launches, fills are fused-map launches."""
import numpy as np

from . import base
from .base import Expr
from .map import map_with_location
from .ndarray import ndarray
from .shuffle import shuffle
from .views import Slice
from .. import context
from ..array import distarray, extent
from ..context import LocalKernelResult
from ..util import Assert


# ------------------------------------------------------------------ write (write_array.py:32-94)
def _write_mapper(ex, source=None, sregion=None, dst_slice=None):
  """write_array.py:32-44: the part of `sregion` inside this tile is fetched from the data slice."""
  intersection = extent.intersection(ex, sregion)
  if intersection is not None:
    dst_lr = np.asarray(intersection.lr) - np.asarray(sregion.ul)
    dst_ul = np.asarray(intersection.ul) - np.asarray(sregion.ul)
    dst_ex = extent.create(tuple(dst_ul), tuple(dst_lr), dst_slice.shape)
    v = dst_slice.fetch(dst_ex)
    source.update(intersection, v, wait=False)
  return LocalKernelResult(result=None, futures=None)


class WriteArrayExpr(Expr):
  """write_array.py:47-81.  (Like the reference, this mutates `array` in place.)"""
  members = ('array', 'src_slices', 'data', 'dst_slices')

  def dependencies(self):
    return {'array': self.array, 'data': self.data}

  def visit(self, visitor):
    def v(x):
      return visitor.visit(x) if isinstance(x, Expr) else x
    return base.expr_like(self, array=v(self.array), src_slices=self.src_slices, data=v(self.data),
                          dst_slices=self.dst_slices)

  def pretty_str(self):
    return 'WriteArrayExpr[%d] %s %s' % (self.expr_id, self.array, self.data)

  def _evaluate(self, ctx, deps):
    array, data = deps['array'], deps['data']
    sregion = extent.from_slice(self.src_slices, array.shape)
    if isinstance(data, np.ndarray):
      piece = data if sregion.shape == data.shape else data[self.dst_slices]
      be = ctx.backend
      array.update(sregion, be.astype(be.from_numpy(np.ascontiguousarray(piece)), array.dtype))
    elif isinstance(data, distarray.DistArray):
      dst_slice = Slice(data, self.dst_slices)
      Assert.eq(sregion.shape, dst_slice.shape)
      array.foreach_tile(mapper_fn=_write_mapper,
                         kw={'source': array, 'sregion': sregion, 'dst_slice': dst_slice})
    else:
      raise TypeError('write: data must be a numpy array or a distributed array, not %s' % type(data))
    return array

  def compute_shape(self):
    return self.array.shape


def write(array, src_slices, data, dst_slices):
  """array[src_slices] = data[dst_slices] (write_array.py:84-94)."""
  return WriteArrayExpr(array=array, src_slices=src_slices, data=data, dst_slices=dst_slices)


# ------------------------------------------------------------------ region_map (region_map.py:8-66)
def _region_mapper(tile, ex, region, user_fn, fn_kw=None):
  """region_map.py:8-40: tiles outside every region are returned as they are; inside, a copy of the
  tile gets `user_fn(view of the intersection, ex, **fn_kw)` pasted over the intersection."""
  be = context.get().backend
  ex = extent.from_tuple(ex)
  if fn_kw is None:
    fn_kw = {}
  for area in region:
    intersection = extent.intersection(area, ex)
    if intersection:
      result = be.copy(tile)
      subslice = extent.offset_slice(ex, intersection)
      value = user_fn(result[subslice], ex, **fn_kw)
      be.assign_box(result, subslice, value)
      return result
  return tile


_region_mapper._sp_tile_fn = True   # runs on backend tensors (not lowered to a register program)


def region_map(array, region, fn, fn_kw={}):
  """region_map.py:43-66.  `fn(tile_view, extent, **kw)` receives a backend tensor view and returns a
  scalar, a NumPy array or a backend tensor of the view's shape."""
  if isinstance(region, extent.TileExtent):
    region = list([region])
  kw = {'fn_kw': fn_kw, 'user_fn': fn, 'region': region}
  return map_with_location(array, fn=_region_mapper, fn_kw=kw)


# ------------------------------------------------------------------ assign (assign.py:11-52)
def _assign_mapper(tile, ex, assign_region, value):
  """assign.py:11-33."""
  if np.isscalar(value):
    return value
  # the part of `value` that lands on this tile: the tile's share of the region, as slices of the region
  part = extent.offset_slice(assign_region, extent.intersection(assign_region, ex))
  if len(assign_region.shape) != len(value.shape):
    # value has fewer axes than the region (a[2, :, :] = v): match its axes to the region's by length, in order
    # (the reference's rule, assign.py:41-47)
    picked, at = [], -1
    for n in value.shape:
      at = assign_region.shape.index(n, at + 1)
      picked.append(part[at])
    part = tuple(picked)
  return value[part] if isinstance(value, np.ndarray) else value.fetch(extent.from_slice(part, value.shape))


def assign(a, idx, value):
  """a[idx] = value as a NEW array (assign.py:36-52).  value: scalar, array_like or DistArray."""
  if not isinstance(idx, extent.TileExtent):
    if np.isscalar(idx):
      idx = slice(idx, idx + 1)
    region = extent.from_slice(idx, a.shape)
  else:
    region = idx
  if isinstance(value, Expr):
    value = value.evaluate()
  return region_map(a, region, _assign_mapper, {'assign_region': region, 'value': value})


# ------------------------------------------------------------------ retile (retile.py:12-32)
def _retile_mapper(array, ex, orig_array):
  yield ex, orig_array.fetch(ex)


def retile(array, tile_hint):
  """Same values, new tiling (retile.py:16-32)."""
  if isinstance(array, Expr):
    array = array.evaluate()
  tiling_type = int(tile_hint[0] == array.shape[0])
  new_array = shuffle(ndarray(array.shape, dtype=array.dtype, tile_hint=tile_hint).evaluate(),
                      _retile_mapper, kw={'orig_array': array}, shape_hint=array.shape,
                      cost_hint={hash(array): {'%d%d' % (tiling_type, tiling_type): 0,
                                               '%d%d' % (1 - tiling_type, tiling_type): np.prod(array.shape)}})
  return new_array.optimized()
