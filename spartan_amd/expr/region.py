"""Out-of-place region operators: `region_map`, `assign`, `retile` (API of the reference's
spartan/expr/operator/region_map.py:43-66, spartan/expr/assign.py:36-52, spartan/expr/retile.py:16-32).  Resolved on
first use (spartan_amd.__getattr__); not part of the default import.

One node carries all three ideas: `PatchExpr(array, boxes, patch)` builds a NEW array cut like `array` in which the
cells under the boxes are what `patch(view, tile extent, meeting box)` says and every other cell is the source's.
A tile no box meets is handed on as it is (the new array's table points at the same blob, as the reference re-uses
the tile, region_map.py:35-40); a tile a box meets is copied once (one HBM copy), and the patch is pasted over the
meeting (a box copy or a fused fill).  `retile` is the in-place loader of write.py pointed at a fresh array.
"""
import numpy as np

from . import base
from .base import Expr, as_array
from .ndarray import ndarray
from .write import write
from .. import context
from ..array import distarray, extent, tile as tile_mod
from ..context import LocalKernelResult


def _patch_tile(ex, source, boxes, patch):
  """One tile of the new array.  Runs on every rank (fetches are collective walks); only the rank that owns the
  tile's worker launches."""
  ctx = context.get()
  meet = None
  for box in boxes:
    meet = extent.intersection(box, ex)
    if meet is not None:
      break
  here = source.fetch(ex)
  value = None if meet is None else patch.value_for(ex, meet)         # may fetch from another array: all ranks
  meta = (np.dtype(source.dtype), False)
  if not ctx.executing:
    return LocalKernelResult(result=[(ex, ctx.create(None))], meta=meta)
  be = ctx.backend
  if meet is None:
    out = here                                      # untouched: the same blob under a new tile id
  else:
    out = be.copy(here)
    where = extent.offset_slice(ex, meet)
    be.assign_box(out, where, patch.apply(out[where], ex, value))
  return LocalKernelResult(result=[(ex, ctx.create(tile_mod.from_data(out, dtype=be.dtype_of(out))))], meta=meta)


class _UserPatch(object):
  """region_map: the cells under the meeting become fn(view of them, the tile's extent, **kw)."""

  def __init__(self, fn, kw):
    self.fn, self.kw = fn, dict(kw or {})

  def value_for(self, ex, meet):
    return None

  def apply(self, view, ex, value):
    return self.fn(view, ex, **self.kw)


class _ValuePatch(object):
  """assign: the cells under the meeting become the matching cells of `value` (a scalar, a NumPy array or a
  distributed array laid over `box`)."""

  def __init__(self, box, value):
    self.box, self.value = box, value

  def value_for(self, ex, meet):
    value, box = self.value, self.box
    if np.isscalar(value):
      return value
    part = extent.offset_slice(box, meet)                 # the meeting, in the box's own coordinates
    if len(value.shape) != len(box.shape):
      # a value with fewer axes than the box (a[10, :] = v with v 1-d): its axes are matched to the box's by
      # length, left to right, each box axis used at most once (the reference's rule, assign.py:21-27)
      axes = iter(range(len(box.shape)))
      part = tuple(part[next(ax for ax in axes if box.shape[ax] == n)] for n in value.shape)
    if isinstance(value, np.ndarray):
      return value[part]
    return value.fetch(extent.from_slice(part, value.shape))

  def apply(self, view, ex, value):
    if isinstance(value, np.ndarray) and value.shape != tuple(view.shape):
      value = value.reshape(tuple(view.shape))
    elif hasattr(value, 'reshape') and not isinstance(value, np.ndarray) and tuple(value.shape) != tuple(view.shape):
      value = value.reshape(tuple(view.shape))
    return value


class PatchExpr(Expr):
  members = ('array', 'boxes', 'patch')

  def dependencies(self):
    return {'array': self.array}

  def visit(self, visitor):
    return base.expr_like(self, array=visitor.visit(self.array), boxes=self.boxes, patch=self.patch)

  def pretty_str(self):
    return 'Patch[%d](%s, %d boxes)' % (self.expr_id, self.array, len(self.boxes))

  def compute_shape(self):
    return self.array.shape

  def _evaluate(self, ctx, deps):
    source = deps['array']
    if not isinstance(source, distarray.DistArray):
      raise TypeError('region_map / assign need a distributed array, not %s' % type(source))
    return source.map_to_array(_patch_tile, kw={'source': source, 'boxes': list(self.boxes), 'patch': self.patch})


def region_map(array, region, fn, fn_kw=None):
  """A new array: `array` with fn(view, extent, **fn_kw) written over the cells under `region` (one TileExtent or a
  list of them; where boxes overlap inside one tile the first one that meets the tile counts, as in the
  reference).  fn gets a backend view of the cells and returns a scalar, a NumPy array or a backend array."""
  boxes = [region] if isinstance(region, extent.TileExtent) else list(region)
  return PatchExpr(array=as_array(array), boxes=boxes, patch=_UserPatch(fn, fn_kw))


def assign(a, idx, value):
  """A new array equal to `a` with a[idx] = value.  idx: int, slice, tuple of them, or a TileExtent; value: scalar,
  array_like, distributed array or expression."""
  a = as_array(a)
  if isinstance(idx, extent.TileExtent):
    box = idx
  else:
    box = extent.from_slice(slice(idx, idx + 1) if np.isscalar(idx) else idx, a.shape)
  if isinstance(value, Expr):
    value = value.evaluate()
  elif not np.isscalar(value) and not isinstance(value, (np.ndarray, distarray.DistArray)):
    value = np.asarray(value)
  return PatchExpr(array=a, boxes=[box], patch=_ValuePatch(box, value))


def retile(array, tile_hint):
  """The same values cut into tiles of `tile_hint`: a fresh array of that tiling, filled tile by tile from the
  source (every new tile pulls its own box: write.py)."""
  source = as_array(array).evaluate()
  everything = tuple(slice(0, n) for n in source.shape)
  fresh = ndarray(source.shape, dtype=source.dtype, tile_hint=tuple(tile_hint))
  return write(fresh, everything, base.Val(val=source), everything)
