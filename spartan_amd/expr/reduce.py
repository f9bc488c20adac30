"""reduce: mirror of the reference's spartan/expr/operator/reduce.py.  The
local reduction is one fused map->reduce HIP launch; the cross-tile combine is
`output.update(...)` with the accumulate fn (an RCCL reduce / reduce-scatter for
the regular patterns, see array/distarray.py UpdateBatch)."""
import collections

from . import base, broadcast
from .base import Expr, ListExpr
from .local import LocalInput, LocalReduceExpr, make_var
from .map import get_local_values
from .. import context
from ..array import distarray, extent, tile
from ..context import LocalKernelResult
from ..util import Assert


def _reduce_mapper(ex, children, child_to_var, op, axis, output):
  """reduce.py:21-70."""
  ctx = context.get()
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
  if done:
    # the cut of the result array (metadata: extents per shape / hint / worker count are remembered, distarray._cuts)
    try:
      distarray.compute_extents(tuple(extent.shape_for_reduction(children[0].shape, node.axis)), node.tile_hint,
                                ctx.num_workers)
    except Exception:      # noqa: BLE001 -- whatever is wrong with the shape, _evaluate will say it
      pass
  return done


def _first(pair):
  return pair[0]


def _evaluate_aligned(node, ctx, values):
  """ReduceExpr._evaluate for one process and operands that are dense, whole, written everywhere and cut the same
  way (or scalars) -- expr/map._evaluate_aligned's case: the tiles' own tensors go to the backend's fused
  map -> reduce, tile after tile in table order, and each partial is merged into the result array by the same
  `update` the general path issues, in the same order.  One difference in ORDER of host work, none in results: the
  kernels are launched first and the result array (metadata: its cut, its empty tiles) is made while they run,
  instead of before the first launch.  Returns None, having done nothing, when the case does not apply."""
  if ctx.world.size != 1 or ctx.pending is not None or not len(values):
    return None
  DA, LW = distarray.DistArrayImpl, distarray.LocalWrapper
  lead = values[0]
  if type(lead) is not DA or lead.sparse or lead.bad_tiles or not lead.tiles or len(lead.tiles) > 256:
    return None
  for v in values[1:]:
    t = type(v)
    if t is DA:
      if v.sparse or v.bad_tiles or v.shape != lead.shape or (v.tiles is not lead.tiles and list(v.tiles) != list(lead.tiles)):
        return None
    elif t is LW:
      if v._data.ndim != 0:
        return None
    else:
      return None
  blobs = ctx._blobs
  ALL_SET, DENSE = tile.MASK_ALL_SET, tile.TYPE_DENSE
  names, axis = node.child_to_var, node.axis
  rows = []
  for ex, tid in lead.tiles.items():
    operands = {}
    for v, name in zip(values, names):
      if type(v) is LW:
        operands[name] = v._scalar if v._scalar is not None else v._data
        continue
      t = blobs.get(v.tiles[ex])
      if t is None or type(t.mask) is not int or t.mask != ALL_SET or t.type != DENSE or t.data is None or not t.shape:
        return None
      operands[name] = t.data
    operands['extent'] = ex
    operands['axis'] = axis
    rows.append((ex, tid.worker, operands))
  if len(rows) > 1:
    # the order run_kernel walks the tiles in (distarray.kernel_order): the order the partials meet in
    at = {tid: k for k, tid in enumerate(distarray.kernel_order(lead, list(lead.tiles.values()), ctx))}
    tids = list(lead.tiles.values())
    rows = [row for _, row in sorted(zip([at[t] for t in tids], rows), key=_first)]
  backend, op = ctx.backend, node.op
  outer_worker = ctx.current_worker
  partials = []
  try:
    for ex, worker, operands in rows:
      ctx.current_worker = worker
      local = backend.evaluate_reduce(op, operands, ex, axis)
      dst = extent.index_for_reduction(ex, axis)
      Assert.eq(int(local.size), dst.size)
      partials.append((worker, dst, local.reshape(dst.shape)))
    output = distarray.create(extent.shape_for_reduction(lead.shape, axis), node.dtype_fn(lead),
                              reducer=node.accumulate_fn, tile_hint=node.tile_hint)
    # (the partials join ONE batch, as the updates of a kernel do: what is whole-array and regular is then merged
    #  whole -- UpdateBatch._merge_whole_partials -- instead of piece by piece)
    batch = distarray.UpdateBatch(ctx)
    ctx.pending = batch
    try:
      for worker, dst, local in partials:
        ctx.current_worker = worker
        output.update(dst, local, owned=True)
    finally:
      ctx.pending = None
    batch.flush()
  finally:
    ctx.current_worker = outer_worker
  return output


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
    fast = _evaluate_aligned(self, ctx, deps['children'])
    if fast is not None:
      return fast
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
