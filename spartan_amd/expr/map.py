"""map / map2 / map_with_location: mirror of the reference's
spartan/expr/operator/map.py and map_with_location.py.  The per-tile body
(`op.evaluate` in the reference) is ONE fused HIP kernel launch issued through
the backend."""
import collections

from . import base
from .base import Expr, ListExpr, TupleExpr, as_array
from .broadcast import Broadcast, broadcast
from .local import FnCallExpr, LocalInput, LocalMapExpr, LocalMapLocationExpr, make_var
from .. import context, util
from ..array import distarray, extent, tile
from ..context import LocalKernelResult
from ..util import Assert


def get_local_values(ex, children, child_to_var):
  """map.py:33-45."""
  local_values = {}
  for child, childv in zip(children, child_to_var):
    if isinstance(child, Broadcast):
      local_val = child.fetch_base_tile(ex)
    else:
      local_val = child.fetch(ex)
    local_values[childv] = local_val
  return local_values


def tile_mapper(ex, children, child_to_var, op):
  """map.py:48-88: evaluate the (fused) map on one tile -> new local tile."""
  ctx = context.get()
  local_values = get_local_values(ex, children, child_to_var)
  local_values['extent'] = ex
  if not ctx.executing:
    # another rank owns this tile: only the tile id is allocated here
    return LocalKernelResult(result=[(ex, ctx.create(None))])
  result = ctx.backend.evaluate_map(op, local_values, ex)
  # (the reference's identity shortcut, map.py:76-77, is not taken: whether a
  # result aliases its input is only known on the executing rank, and tile ids
  # must advance identically on every rank)
  Assert.eq(ex.shape, tuple(result.shape), 'Bad shape -- result = %s, op = (%s)', result.shape, op)
  result_tile = tile.from_data(result, dtype=ctx.backend.dtype_of(result))
  tile_id = ctx.create(result_tile)
  return LocalKernelResult(result=[(ex, tile_id)])


class MapExpr(Expr):
  """map.py:91-169."""
  members = ('children', 'child_to_var', 'op')

  def pretty_str(self):
    return 'Map[%d](%s, %s)' % (self.expr_id, self.op.pretty_str(), self.children.pretty_str())

  def dependencies(self):
    return {'children': self.children, 'child_to_var': self.child_to_var, 'op': self.op}

  def visit(self, visitor):
    return base.expr_like(self, children=visitor.visit(self.children),
                          child_to_var=self.child_to_var, op=self.op)

  def compute_shape(self):
    """map.py:105-128: NumPy broadcasting of the children's shapes."""
    orig_shapes = [list(x.shape) for x in self.children]
    dims = [len(shape) for shape in orig_shapes]
    max_dim = max(dims)
    new_shapes = []
    for shp in orig_shapes:
      diff = max_dim - len(shp)
      new_shapes.append([1] * diff + shp)
    output_shape = collections.defaultdict(int)
    for s in new_shapes:
      for i, v in enumerate(s):
        output_shape[i] = max(output_shape[i], v)
    return tuple([output_shape[i] for i in range(len(output_shape))])

  def _evaluate(self, ctx, deps):
    children = list(deps['children'])
    child_to_var = list(deps['child_to_var'])
    children = broadcast(children)
    largest = distarray.largest_value(children)
    i = children.index(largest)
    children[0], children[i] = children[i], children[0]
    child_to_var[0], child_to_var[i] = child_to_var[i], child_to_var[0]
    return largest.map_to_array(tile_mapper, kw={'children': children,
                                                 'child_to_var': child_to_var,
                                                 'op': self.op})


def map(inputs, fn, numpy_expr=None, fn_kw=None):
  """map.py:172-205."""
  assert fn is not None
  if not util.is_iterable(inputs):
    inputs = [inputs]
  op_deps = []
  children = []
  child_to_var = []
  for v in inputs:
    v = as_array(v)
    varname = make_var()
    children.append(v)
    child_to_var.append(varname)
    op_deps.append(LocalInput(idx=varname))
  children = ListExpr(vals=children)
  op = LocalMapExpr(fn=fn, kw=fn_kw, pretty_fn=numpy_expr, deps=op_deps)
  return MapExpr(children=children, child_to_var=child_to_var, op=op)


def map_with_location(inputs, fn, numpy_expr=None, fn_kw=None):
  """map_with_location.py:22-60: the mapper also sees the tile's extent."""
  assert fn is not None
  if not util.is_iterable(inputs):
    inputs = [inputs]
  op_deps = []
  children = []
  child_to_var = []
  for v in inputs:
    v = as_array(v)
    varname = make_var()
    children.append(v)
    child_to_var.append(varname)
    op_deps.append(LocalInput(idx=varname))
  op_deps += [LocalInput(idx='extent')]
  children = ListExpr(vals=children)
  op = LocalMapLocationExpr(fn=fn, kw=fn_kw, pretty_fn=numpy_expr, deps=op_deps)
  return MapExpr(children=children, child_to_var=child_to_var, op=op)


def join_mapper(ex, arrays, axes, local_user_fn, local_user_fn_kw, target):
  """map.py:243-286: join the tile with the matching slabs of the other arrays,
  run the user fn and push its outputs into `target`."""
  ctx = context.get()
  if len(axes) == 0:
    tiles = [arrays[i].fetch(ex) for i in range(len(arrays))]
    join_extents = ex
  else:
    first_extent = extent.change_partition_axis(ex, axes[0])
    if first_extent is None:
      return LocalKernelResult(result=[])
    keys = (first_extent.ul[axes[0]], first_extent.lr[axes[0]])
    join_extents = [first_extent]
    for i in range(1, len(arrays)):
      ul = [0 for _ in range(len(arrays[i].shape))]
      lr = list(arrays[i].shape)
      ul[axes[i]] = keys[0]
      lr[axes[i]] = keys[1]
      join_extents.append(extent.create(ul, lr, arrays[i].shape))
    tiles = [arrays[i].fetch(join_extents[i]) for i in range(len(arrays))]
  if local_user_fn_kw is None:
    local_user_fn_kw = {}
  result = local_user_fn(join_extents, tiles, **local_user_fn_kw)
  if result is not None:
    # only a mapper that declares its outputs freshly produced hands them over: what an arbitrary user
    # mapper yields may be (a view of) a fetched input tile, and a target tile adopting it would alias
    # -- and later reduce into -- the source array's storage
    fresh = bool(getattr(local_user_fn, 'yields_fresh_tensors', False))
    for tex, v in result:
      target.update(tex, v, wait=False, owned=fresh)
  return LocalKernelResult(result=[])


class Map2Expr(Expr):
  """map.py:289-334."""
  members = ('arrays', 'axes', 'fn', 'fn_kw', 'shape_', 'update_region', 'tile_hint', 'dtype', 'reducer')

  def pretty_str(self):
    return 'Map2[%d](arrays=%s, axes=%s, tile_hint=%s)' % (self.expr_id, self.arrays, self.axes, self.tile_hint)

  def dependencies(self):
    return {'arrays': self.arrays}

  def visit(self, visitor):
    return base.expr_like(self, arrays=visitor.visit(self.arrays), axes=self.axes, fn=self.fn,
                          fn_kw=self.fn_kw, shape_=self.shape_, update_region=self.update_region,
                          tile_hint=self.tile_hint, dtype=self.dtype, reducer=self.reducer)

  def compute_shape(self):
    return self.shape_

  def _evaluate(self, ctx, deps):
    arrays = deps['arrays']
    dtype = self.dtype
    if dtype is None:
      dtype = arrays[0].dtype
    if self.update_region is not None:
      raise NotImplementedError('map2(update_region=...) (region_join_mapper) is not on the tile path')
    sparse = len(arrays) > 1 and bool(getattr(arrays[0], 'sparse', False) and getattr(arrays[1], 'sparse', False))   # map.py:323
    target = distarray.create(self.shape_, dtype, sharder=None, reducer=self.reducer,
                              tile_hint=self.tile_hint, sparse=sparse)
    # a mapper may know how to run its whole join as one pipeline of transfers and kernels when operands and
    # target are laid out regularly (dot: the K-split with its all-to-all and reduce-scatter, dot.ksplit_plan)
    plan = getattr(self.fn, 'collective_plan', None)
    if plan is not None and plan(arrays, self.axes, target, self.fn_kw):
      return target
    arrays[0].foreach_tile(mapper_fn=join_mapper,
                           kw=dict(arrays=arrays, axes=self.axes, local_user_fn=self.fn,
                                   local_user_fn_kw=self.fn_kw, target=target))
    return target


def map2(arrays, axes=[], fn=None, fn_kw=None, shape=None, update_region=None,
         tile_hint=None, dtype=None, reducer=None):
  """map.py:337-375."""
  if not util.is_iterable(arrays):
    arrays = [arrays]
  if not util.is_iterable(axes):
    axes = [axes]
  assert fn is not None
  assert list(axes) == [] or len(arrays) == len(axes)
  assert shape is not None
  arrays = TupleExpr(vals=tuple(base.lazify(a) if not isinstance(a, Expr) else a for a in arrays))
  axes = tuple(axes)
  return Map2Expr(arrays=arrays, axes=axes, fn=fn, fn_kw=fn_kw, shape_=tuple(shape),
                  update_region=update_region, tile_hint=tile_hint, dtype=dtype, reducer=reducer)
