"""reduce: mirror of the reference's spartan/expr/operator/reduce.py.  The
local reduction is one fused map->reduce HIP launch; the cross-tile combine is
`output.update(...)` with the accumulate fn (an RCCL reduce / reduce-scatter for
the regular patterns, see array/distarray.py UpdateBatch)."""
import collections

from . import base, broadcast
from .base import Expr, ListExpr
from .local import LocalInput, LocalReduceExpr, make_var
from .. import context
from ..array import distarray, extent
from ..context import LocalKernelResult
from ..util import Assert


def _reduce_mapper(ex, children, child_to_var, op, axis, output):
  """reduce.py:21-70."""
  ctx = context.get()
  from .map import get_local_values
  local_values = get_local_values(ex, children, child_to_var)
  local_values.update(extent=ex, axis=axis)
  dst_extent = extent.index_for_reduction(ex, axis)
  if ctx.executing:
    local_reduction = ctx.backend.evaluate_reduce(op, local_values, ex, axis)
    Assert.eq(int(local_reduction.size),
              dst_extent.size)
    local_reduction = local_reduction.reshape(dst_extent.shape)
  else:
    local_reduction = distarray.Absent(dst_extent.shape, output.dtype)
  output.update(dst_extent, local_reduction, owned=True)
  return LocalKernelResult(result=[])


def prelower(node, ctx):
  """expr/map.prelower for a ReduceExpr whose inputs exist already: the fused map -> reduce program of its local
  reduction is handed to the backend before the first evaluation (`prelower_reduce`: lowered and remembered, not
  run), one tile per distinct tile shape.  The walk is the prelude of ReduceExpr._evaluate and _reduce_mapper."""
  from .map import get_local_values
  hook = getattr(ctx.backend, 'prelower_reduce', None)
  kids = getattr(node.children, 'vals', None)
  if hook is None or ctx.world.size != 1 or not kids or not all(isinstance(k, base._Leaf) for k in kids):
    return 0
  values = [k.evaluate() for k in kids]
  if not all(isinstance(v, distarray.LocalWrapper) or
             (isinstance(v, distarray.DistArrayImpl) and not v.sparse and not v.bad_tiles) for v in values):
    return 0
  children = broadcast.broadcast(values)
  largest = distarray.largest_value(children)
  if not isinstance(largest, distarray.DistArrayImpl):
    return 0
  cut = largest.tiles.keys()
  if not all(isinstance(v, broadcast.Broadcast) or v is largest or v.tiles.keys() == cut for v in children):
    return 0
  done, shapes = 0, set()
  for ex in cut:
    if ex.shape in shapes or len(shapes) >= 4:
      continue
    shapes.add(ex.shape)
    local_values = get_local_values(ex, children, node.child_to_var)
    local_values.update(extent=ex, axis=node.axis)
    done += bool(hook(node.op, local_values, ex, node.axis))
  return done


class ReduceExpr(Expr):
  """reduce.py:73-127."""
  members = ('children', 'child_to_var', 'axis', 'dtype_fn', 'op', 'accumulate_fn', 'tile_hint')

  def dependencies(self):
    return {'children': self.children}

  def visit(self, visitor):
    return base.expr_like(self, children=visitor.visit(self.children), child_to_var=self.child_to_var,
                          axis=self.axis, dtype_fn=self.dtype_fn, op=self.op,
                          accumulate_fn=self.accumulate_fn, tile_hint=self.tile_hint)

  def compute_shape(self):
    # (per-axis maximum over the children, left-aligned as the reference does it, reduce.py:87-94)
    nd = max(len(c.shape) for c in self.children)
    input_shape = tuple(max([c.shape[i] for c in self.children if i < len(c.shape)] or [0]) for i in range(nd))
    return tuple(extent.shape_for_reduction(input_shape, self.axis))

  def pretty_str(self):
    return 'Reduce(%s, axis=%s, %s, hint=%s)' % (getattr(self.op.fn, '__name__', self.op.fn), self.axis,
                                                 self.children.pretty_str(), self.tile_hint)

  def _evaluate(self, ctx, deps):
    children = deps['children']
    children = broadcast.broadcast(list(children))
    largest = distarray.largest_value(children)
    dtype = self.dtype_fn(children[0])
    shape = extent.shape_for_reduction(children[0].shape, self.axis)
    output_array = distarray.create(shape, dtype, reducer=self.accumulate_fn, tile_hint=self.tile_hint)
    largest.foreach_tile(_reduce_mapper, kw={'children': children,
                                             'child_to_var': self.child_to_var,
                                             'op': self.op,
                                             'axis': self.axis,
                                             'output': output_array})
    return output_array


def reduce(v, axis, dtype_fn, local_reduce_fn, accumulate_fn, fn_kw=None, tile_hint=None):
  """reduce.py:130-167."""
  if fn_kw is None:
    fn_kw = {}
  varname = make_var()
  assert 'axis' not in fn_kw, '"axis" argument is reserved.'
  fn_kw['axis'] = axis
  reduce_op = LocalReduceExpr(fn=local_reduce_fn,
                              deps=[LocalInput(idx='extent'), LocalInput(idx=varname)],
                              kw=fn_kw)
  return ReduceExpr(children=ListExpr(vals=[base.as_array(v)]), child_to_var=[varname], axis=axis,
                    dtype_fn=dtype_fn, op=reduce_op, accumulate_fn=accumulate_fn, tile_hint=tile_hint)
