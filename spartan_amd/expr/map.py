"""Element-wise maps and joins: the API of the reference's spartan/expr/operator/map.py and map_with_location.py
(`map`, `map_with_location`, `map2`, `MapExpr`, `Map2Expr`, `tile_mapper`, `join_mapper`).  The per-tile body of a
map -- the reference evaluates the operator tree node by node with NumPy -- is ONE fused HIP kernel launch issued
through the backend."""
import collections

import numpy as np

from . import base
from .base import Expr, ListExpr, TupleExpr, as_array
from .broadcast import Broadcast, broadcast, common_shape
from .local import LocalInput, LocalMapExpr, LocalMapLocationExpr, make_var
from .. import context, util
from ..array import distarray, extent, tile
from ..context import LocalKernelResult


def get_local_values(ex, children, child_to_var):
  """{variable name: the piece of that input under the tile `ex`} -- stretched inputs hand out their un-stretched
  slab (the kernel broadcasts through zero strides)."""
  values = {name: (child.fetch_base_tile(ex) if isinstance(child, Broadcast) else child.fetch(ex))
            for child, name in zip(children, child_to_var)}
  tile.reject_masked(values, 'map / reduce')
  return values


def tile_mapper(ex, children, child_to_var, op):
  """One tile of an element-wise map: the (fused) operator `op` over the inputs' pieces, ONE kernel launch,
  result kept as a new tile on the worker that ran it (reference tile_mapper, map.py:48-88)."""
  ctx = context.get()
  operands = get_local_values(ex, children, child_to_var)
  operands['extent'] = ex
  # With several ranks every rank needs the dtype of the new array, and only the rank that ran the kernel has a
  # tile to look at: the backend derives it from the operator tree and the operands' dtypes where it can -- on EVERY
  # rank, from metadata all of them hold, so all of them take the same decision -- and the array is then built
  # without asking the owner of its first tile (from_table -> Context.tile_meta, one control-plane broadcast per map)
  meta = None
  if ctx.world.size > 1:
    derive = getattr(ctx.backend, 'map_result_meta', None)
    meta = derive(op, operands, ex) if derive is not None else None
  if not ctx.executing:
    return LocalKernelResult(result=[(ex, ctx.create(None))], meta=meta)      # another rank's tile: only the id advances
  out = ctx.backend.evaluate_map(op, operands, ex)
  # (the reference hands back the INPUT tile's id when the operator returned its input unchanged, map.py:76-77;
  #  whether that happened is only known where the kernel ran, and tile ids must advance alike on all ranks)
  if tuple(out.shape) != ex.shape:
    raise AssertionError('Bad shape -- result = %s, op = (%s)' % (tuple(out.shape), op))
  if meta is not None and (np.dtype(ctx.backend.dtype_of(out)) != meta[0] or tile.is_sparse_blob(out) != meta[1]):
    raise AssertionError('derived tile type %s, kernel produced %s (op = %s)' % (meta, ctx.backend.dtype_of(out), op))
  return LocalKernelResult(result=[(ex, ctx.create(tile.from_data(out, dtype=ctx.backend.dtype_of(out))))], meta=meta)


class MapExpr(Expr):
  """Element-wise operator over broadcast-compatible inputs; the result is tiled like its largest input."""
  members = ('children', 'child_to_var', 'op')

  def pretty_str(self):
    return 'Map[%d](%s, %s)' % (self.expr_id, self.op.pretty_str(), self.children.pretty_str())

  def visit(self, visitor):
    return base.expr_like(self, children=visitor.visit(self.children), child_to_var=self.child_to_var, op=self.op)

  def compute_shape(self):
    return common_shape([tuple(c.shape) for c in self.children])

  def _evaluate(self, ctx, deps):
    fast = _evaluate_aligned(self, ctx, deps['children'], deps['child_to_var'])
    if fast is not None:
      return fast
    inputs = broadcast(list(deps['children']))
    names = list(deps['child_to_var'])
    # the largest input drives the tile walk (its tiles are read in place, the others are fetched to them)
    lead = inputs.index(distarray.largest_value(inputs))
    inputs[0], inputs[lead] = inputs[lead], inputs[0]
    names[0], names[lead] = names[lead], names[0]
    return inputs[0].map_to_array(tile_mapper, kw={'children': inputs, 'child_to_var': names, 'op': self.op})


def _evaluate_aligned(node, ctx, values, names):
  """MapExpr._evaluate for the case a driver loop hits every time: one process, every array operand dense, whole,
  written everywhere and CUT THE SAME WAY (same shape, same tile table), the other operands scalars.  Then nothing
  is stretched, fetched, gathered or combined: tile by tile the operands are the tiles' own tensors and the result is
  a new tile on the same worker -- what tile_mapper / run_kernel / from_table arrive at through their general
  machinery (broadcast views, fetch plans, update batches, shape discovery), with the same calls into the backend
  in the same order.  Returns None, having done nothing, whenever any of that does not hold."""
  if ctx.world.size != 1 or ctx.pending is not None:
    return None
  DA, LW = distarray.DistArrayImpl, distarray.LocalWrapper
  lead = None
  for v in values:
    t = type(v)
    if t is DA:
      if v.sparse or v.bad_tiles:
        return None
      if lead is None:
        lead = v
      elif v.shape != lead.shape or (v.tiles is not lead.tiles and list(v.tiles) != list(lead.tiles)):
        return None
    elif t is LW:
      if v._data.ndim != 0:
        return None
    else:
      return None
  if lead is None or not lead.tiles:
    return None
  blobs = ctx._blobs
  ALL_SET, DENSE = tile.MASK_ALL_SET, tile.TYPE_DENSE
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
    rows.append((ex, tid.worker, operands))
  if len(rows) > 1:
    # the tiles in the order run_kernel walks them (distarray.kernel_order): a map's tiles do not meet, but an operator
    # tree with a random source draws from its worker's stream tile after tile -- the same draws on either path
    at = {tid: k for k, tid in enumerate(distarray.kernel_order(lead, list(lead.tiles.values()), ctx))}
    order = sorted(range(len(rows)), key=lambda i, tids=list(lead.tiles.values()): at[tids[i]])
    rows = [rows[i] for i in order]
  backend, op = ctx.backend, node.op
  table = collections.OrderedDict()
  outer_worker = ctx.current_worker
  dtype = None
  try:
    for ex, worker, operands in rows:
      ctx.current_worker = worker
      out = backend.evaluate_map(op, operands, ex)
      if tuple(out.shape) != ex.shape:
        raise AssertionError('Bad shape -- result = %s, op = (%s)' % (tuple(out.shape), op))
      if tile.is_sparse_blob(out):
        raise AssertionError('a map over dense tiles produced a sparse tile (op = %s)' % (op,))
      dt = backend.dtype_of(out)
      if dtype is None:
        dtype = dt
      table[ex] = ctx.create(tile.from_data(out, dtype=dt), hint=worker)
  finally:
    ctx.current_worker = outer_worker
  if len(table) > 1:
    # (the result's tile table in the operands' order, whatever order the kernels ran in: arrays cut the same way
    #  compare equal tile by tile)
    table = collections.OrderedDict((ex, table[ex]) for ex in lead.tiles)
  return distarray.DistArrayImpl(shape=lead.shape, dtype=dtype, tiles=table, reducer_fn=None, sparse=False)


def prelower(node, ctx):
  """Hand the kernels of a map whose inputs all exist already to the backend BEFORE the map is evaluated
  (`prelower_map`: lowered and remembered, not run) -- one tile per distinct tile shape.  Called by the optimiser for
  a DAG it sees for the first time; the walk below is the prelude of MapExpr._evaluate and tile_mapper."""
  hook = getattr(ctx.backend, 'prelower_map', None)
  kids = getattr(node.children, 'vals', None)
  if hook is None or ctx.world.size != 1 or not kids or not all(isinstance(k, base._Leaf) for k in kids):
    return 0
  values = [k.evaluate() for k in kids]
  if not all(isinstance(v, distarray.LocalWrapper) or
             (isinstance(v, distarray.DistArrayImpl) and not v.sparse and not v.bad_tiles) for v in values):
    return 0
  inputs = broadcast(values)
  names = list(node.child_to_var)
  lead = inputs.index(distarray.largest_value(inputs))
  inputs[0], inputs[lead] = inputs[lead], inputs[0]
  names[0], names[lead] = names[lead], names[0]
  if not isinstance(inputs[0], distarray.DistArrayImpl):
    return 0
  # every other input cut like the lead (read in place) or stretched (its one slab): nothing is gathered or copied
  cut = inputs[0].tiles.keys()
  if not all(isinstance(v, Broadcast) or v.tiles.keys() == cut for v in inputs[1:]):
    return 0
  done, shapes = 0, set()
  for ex in cut:
    if ex.shape in shapes or len(shapes) >= 4:
      continue
    shapes.add(ex.shape)
    operands = get_local_values(ex, inputs, names)
    operands['extent'] = ex
    done += bool(hook(node.op, operands, ex))
  return done


def _map_node(inputs, fn, numpy_expr, fn_kw, op_type, extra_vars=()):
  if fn is None:
    raise AssertionError('map needs a function')
  arrays = [as_array(v) for v in (inputs if util.is_iterable(inputs) else [inputs])]
  names = [make_var() for _ in arrays]
  op = op_type(fn=fn, kw=fn_kw, pretty_fn=numpy_expr, deps=[LocalInput(idx=n) for n in names + list(extra_vars)])
  return MapExpr(children=ListExpr(vals=arrays), child_to_var=names, op=op)


def map(inputs, fn, numpy_expr=None, fn_kw=None):
  """fn applied element-wise to the inputs (arrays, expressions, scalars; NumPy broadcasting)."""
  return _map_node(inputs, fn, numpy_expr, fn_kw, LocalMapExpr)


def map_with_location(inputs, fn, numpy_expr=None, fn_kw=None):
  """Like map, but fn also receives the position of the tile it is applied to (its extent, last argument)."""
  return _map_node(inputs, fn, numpy_expr, fn_kw, LocalMapLocationExpr, extra_vars=('extent',))


def join_mapper(ex, arrays, axes, local_user_fn, local_user_fn_kw, target):
  """One tile of a join (reference join_mapper, map.py:243-286).  `ex` is a tile of arrays[0]; it is re-read as
  the slab that cuts axis axes[0] the same way (change_partition_axis), every other array contributes the slab
  with the same index range on ITS join axis, the user function gets ([extents], [slabs]) and what it yields,
  (target extent, data) pairs, is pushed into `target` (merged there by the target's reducer)."""
  if not axes:
    extents = ex
    slabs = [a.fetch(ex) for a in arrays]
  else:
    lead = extent.change_partition_axis(ex, axes[0])
    if lead is None:
      return LocalKernelResult(result=[])
    lo, hi = lead.ul[axes[0]], lead.lr[axes[0]]
    extents = [lead]
    for arr, axis in zip(arrays[1:], axes[1:]):
      ul, lr = [0] * len(arr.shape), list(arr.shape)
      ul[axis], lr[axis] = lo, hi
      extents.append(extent.create(ul, lr, arr.shape))
    slabs = [arr.fetch(e) for arr, e in zip(arrays, extents)]
  produced = local_user_fn(extents, slabs, **(local_user_fn_kw or {}))
  if produced is not None:
    # only a mapper that declares its outputs freshly produced hands them over: what an arbitrary user
    # mapper yields may be (a view of) a fetched input tile, and a target tile adopting it would alias
    # -- and later reduce into -- the source array's storage
    fresh = bool(getattr(local_user_fn, 'yields_fresh_tensors', False))
    for where, data in produced:
      target.update(where, data, wait=False, owned=fresh)
  return LocalKernelResult(result=[])


def region_join_mapper(ex, arrays, axes, local_user_fn, local_user_fn_kw, target, region):
  """One tile of a join that REWRITES regions of arrays[0] (reference region_join_mapper, map.py:208-241; its only
  user is the blocked Cholesky factorisation).  axes[0] is a LIST of axes: tile `ex` of arrays[0] is re-read as the
  cell with the same number of a grid over those axes.  The target gets that cell unchanged -- unless the cell meets
  one of the `region` boxes: then the user function is called with the cell and, from every further array, the slab
  whose range on ITS join axis (an int, or None: the whole array) equals the cell's range on grid axis i-1; the
  function returns ONE (extent, data) pair and `data` overwrites the cell's part under the first box it meets."""
  ctx = context.get()
  cell = extent.change_partition_axis(ex, axes[0])
  if cell is None:                                   # more tiles than grid cells: nothing to do for this one
    return LocalKernelResult(result=[])
  data = arrays[0].fetch(cell)
  for box in region:
    hit = extent.intersection(box, cell)
    if not hit:
      continue
    extents, slabs = [cell], [data]
    for i, (arr, axis) in enumerate(zip(arrays[1:], axes[1:])):
      ul, lr = [0] * len(arr.shape), list(arr.shape)
      if axis is not None:
        ul[axis], lr[axis] = cell.ul[axes[0][i]], cell.lr[axes[0][i]]
      extents.append(extent.create(ul, lr, arr.shape))
      slabs.append(arr.fetch(extents[-1]))
    _, value = local_user_fn(extents, slabs, **(local_user_fn_kw or {}))
    if not isinstance(data, distarray.Absent):       # the rank that runs this worker's kernels
      be = ctx.backend
      data = be.copy(data)                           # the fetched cell may be a view of the source tile
      be.assign_box(data, extent.offset_slice(cell, hit), value)
    break
  target.update(cell, data, wait=False)
  return LocalKernelResult(result=[])


_warned_grid = []


def _warn_if_grid_tiled(arrays):
  """spartan.dot's join re-reads tile number b of a GRID-tiled operand as a slab ONE index thick (extent.pyx:545-552,
  change_partition_axis), so only as many indices of the contraction as there are tiles take part: the result is the
  reference's, bit for bit (tests/test_dot_grid_tiles.py), and it is not the matrix product.  Say so once."""
  if _warned_grid:
    return
  for arr in arrays:
    tiles = getattr(arr, 'tiles', None)
    if not tiles or len(getattr(arr, 'shape', ())) != 2 or any(len(ex.ul) != 2 for ex in tiles):
      continue
    if len({ex.ul[0] for ex in tiles}) > 1 and len({ex.ul[1] for ex in tiles}) > 1:
      import warnings
      _warned_grid.append(True)
      warnings.warn('spartan.dot on an operand cut into a 2-D grid of tiles %s: the join takes one slab of ONE index per '
                    'tile (as the reference does), so the result is not the matrix product; use row tiles (the '
                    'default) or tile_hint=(rows, all columns)' % (sorted({ex.shape for ex in tiles})[:2],),
                    RuntimeWarning, stacklevel=4)
      return


class Map2Expr(Expr):
  """A join of arrays on chosen axes whose per-tile function writes into a new target array."""
  members = ('arrays', 'axes', 'fn', 'fn_kw', 'shape_', 'update_region', 'tile_hint', 'dtype', 'reducer')

  def pretty_str(self):
    return 'Map2[%d](arrays=%s, axes=%s, tile_hint=%s)' % (self.expr_id, self.arrays, self.axes, self.tile_hint)

  def dependencies(self):
    return {'arrays': self.arrays}

  def visit(self, visitor):
    fields = {name: getattr(self, name) for name in self.members}
    fields['arrays'] = visitor.visit(self.arrays)
    return base.expr_like(self, **fields)

  def compute_shape(self):
    return self.shape_

  def _evaluate(self, ctx, deps):
    arrays = deps['arrays']
    both_sparse = len(arrays) > 1 and all(bool(getattr(a, 'sparse', False)) for a in arrays[:2])   # map.py:323
    target = distarray.create(self.shape_, self.dtype if self.dtype is not None else arrays[0].dtype,
                              sharder=None, reducer=self.reducer, tile_hint=self.tile_hint, sparse=both_sparse)
    # a mapper may know how to run its whole join as one pipeline of transfers and kernels when operands and
    # target are laid out regularly (dot: the K-split with its all-to-all and reduce-scatter, dot.ksplit_plan)
    plan = getattr(self.fn, 'collective_plan', None)
    if getattr(self.fn, 'is_contraction', False):
      _warn_if_grid_tiled(arrays)
    if self.update_region is not None:
      arrays[0].foreach_tile(mapper_fn=region_join_mapper,
                             kw=dict(arrays=arrays, axes=self.axes, local_user_fn=self.fn,
                                     local_user_fn_kw=self.fn_kw, target=target, region=self.update_region))
    elif plan is None or not plan(arrays, self.axes, target, self.fn_kw):
      arrays[0].foreach_tile(mapper_fn=join_mapper,
                             kw=dict(arrays=arrays, axes=self.axes, local_user_fn=self.fn,
                                     local_user_fn_kw=self.fn_kw, target=target))
    return target


def map2(arrays, axes=(), fn=None, fn_kw=None, shape=None, update_region=None,
         tile_hint=None, dtype=None, reducer=None):
  """Join `arrays` on `axes` (one axis per array, or none: the tiles as they are) with `fn(extents, tiles, **fn_kw)`
  yielding (target extent, data) pairs into a new array of `shape` (reference map2, map.py:337-375)."""
  arrays = list(arrays) if util.is_iterable(arrays) else [arrays]
  axes = tuple(axes) if util.is_iterable(axes) else (axes,)
  if fn is None or shape is None or (axes and len(axes) != len(arrays)):
    raise AssertionError('map2 needs fn, shape and one axis per array (or none)')
  nodes = TupleExpr(vals=tuple(a if isinstance(a, Expr) else base.lazify(a) for a in arrays))
  if update_region is not None:
    update_region = tuple(update_region) if util.is_iterable(update_region) else (update_region,)
  return Map2Expr(arrays=nodes, axes=axes, fn=fn, fn_kw=fn_kw, shape_=tuple(shape), update_region=update_region,
                  tile_hint=tile_hint, dtype=dtype, reducer=reducer)
