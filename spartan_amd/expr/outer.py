"""outer: tile x tile cartesian mapper (reference spartan/expr/operator/outer.py),
used by dot when rows > cols."""
from . import base
from .base import Expr, TupleExpr
from .. import context
from ..array import distarray, extent
from ..context import LocalKernelResult


def outer_mapper(ex, arrays, axes, local_user_fn, local_user_fn_kw, target):
  """outer.py:12-59."""
  first_extent = extent.change_partition_axis(ex, axes[0])
  first_tile = arrays[0].fetch(first_extent)
  if local_user_fn_kw is None:
    local_user_fn_kw = {}
  fresh = bool(getattr(local_user_fn, 'yields_fresh_tensors', False))   # see map.join_mapper
  if axes[1] is None:
    outer_extent = extent.from_shape(arrays[1].shape)
    # a mapper may bring its own way of fetching the whole right-hand array (dot: column chunks
    # gathered asynchronously so the gather overlaps the GEMM); default: one replicated fetch
    fetch_rhs = getattr(local_user_fn, 'fetch_rhs', None)
    outer_tile = fetch_rhs(arrays[1], outer_extent) if fetch_rhs is not None else arrays[1].fetch(outer_extent)
    result = local_user_fn(first_extent, first_tile, outer_extent, outer_tile, **local_user_fn_kw)
    if result is not None:
      for tex, v in result:
        target.update(tex, v, wait=False, owned=fresh)
  else:
    done_extent = {}
    for key in arrays[1].tiles.keys():
      outer_extent = extent.change_partition_axis(key, axes[1])
      if outer_extent is None or done_extent.get(outer_extent, None) is not None:
        continue
      outer_tile = arrays[1].fetch(outer_extent)
      result = local_user_fn(first_extent, first_tile, outer_extent, outer_tile, **local_user_fn_kw)
      if result is not None:
        for tex, v in result:
          target.update(tex, v, wait=False, owned=fresh)
      done_extent[outer_extent] = True
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
