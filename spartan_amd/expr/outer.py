"""outer: tile x tile cartesian mapper (reference spartan/expr/operator/outer.py),
used by dot when rows > cols."""
from . import base
from .base import Expr, TupleExpr
from .. import context
from ..array import distarray, extent
from ..context import LocalKernelResult


def _right_hand_slabs(right, axis, fetch_whole):
  """The pieces of the right-hand array a left tile is paired with, as (extent, data) in tile order: the whole array
  once (axis None) or each DISTINCT slab its tiles turn into when they are re-cut along `axis` -- a vector re-cut
  along its other axis yields the same whole-vector slab for every tile, and a cut finer than the new axis yields
  nothing for some (reference outer.py:33-47)."""
  if axis is None:
    everything = extent.from_shape(right.shape)
    yield everything, (fetch_whole(right, everything) if fetch_whole is not None else right.fetch(everything))
    return
  met = set()
  for tile_extent in right.tiles:
    slab = extent.change_partition_axis(tile_extent, axis)
    if slab is None or slab in met:
      continue
    met.add(slab)
    yield slab, right.fetch(slab)


def outer_mapper(ex, arrays, axes, local_user_fn, local_user_fn_kw, target):
  """One tile of the left array against every slab of the right one (kernel-function protocol of the reference's
  outer_mapper, outer.py:12-59): the user function gets (left extent, left data, right extent, right data) and what it
  yields -- (target extent, data) pairs -- is pushed into `target`."""
  left, right = arrays[0], arrays[1]
  here = extent.change_partition_axis(ex, axes[0])
  mine = left.fetch(here)
  kw = local_user_fn_kw or {}
  owned = bool(getattr(local_user_fn, 'yields_fresh_tensors', False))   # see map.join_mapper
  # a mapper may bring its own way of fetching the whole right-hand array (dot: column chunks gathered
  # asynchronously so that the gather overlaps the GEMM); default: one replicated fetch
  fetch_whole = getattr(local_user_fn, 'fetch_rhs', None)
  for there, theirs in _right_hand_slabs(right, axes[1], fetch_whole):
    for where, data in (local_user_fn(here, mine, there, theirs, **kw) or ()):
      target.update(where, data, wait=False, owned=owned)
  return LocalKernelResult(result=[])


class OuterProductExpr(Expr):
  """outer.py:62-99."""
  members = ('arrays', 'axes', 'fn', 'fn_kw', 'shape_', 'dtype', 'tile_hint', 'reducer')

  def pretty_str(self):
    return 'OuterProduct[%d](arrays=%s, axes=%s, tile_hint=%s)' % (
        self.expr_id, self.arrays.pretty_str(), self.axes, self.tile_hint)

  def dependencies(self):
    return {'arrays': self.arrays}

  def visit(self, visitor):
    return base.expr_like(self, arrays=visitor.visit(self.arrays), axes=self.axes, fn=self.fn,
                          fn_kw=self.fn_kw, shape_=self.shape_, dtype=self.dtype,
                          tile_hint=self.tile_hint, reducer=self.reducer)

  def compute_shape(self):
    return self.shape_

  def _evaluate(self, ctx, deps):
    arrays = deps['arrays']
    dtype = self.dtype
    if dtype is None:
      dtype = arrays[0].dtype
    sparse = len(arrays) > 1 and bool(getattr(arrays[0], 'sparse', False) and getattr(arrays[1], 'sparse', False))   # outer.py:94
    target = distarray.create(self.shape_, dtype, sharder=None, reducer=self.reducer,
                              tile_hint=self.tile_hint, sparse=sparse)
    arrays[0].foreach_tile(mapper_fn=outer_mapper,
                           kw=dict(arrays=arrays, axes=self.axes, local_user_fn=self.fn,
                                   local_user_fn_kw=self.fn_kw, target=target))
    return target


def outer(arrays, axes, fn, fn_kw=None, shape=None, tile_hint=None, reducer=None, dtype=None):
  """outer.py:102-120."""
  assert fn is not None
  assert shape is not None
  arrays = TupleExpr(vals=tuple(arrays))
  return OuterProductExpr(arrays=arrays, axes=tuple(axes), fn=fn, fn_kw=fn_kw, shape_=tuple(shape),
                          dtype=dtype, tile_hint=tile_hint, reducer=reducer)
