"""Rewrites of the expression DAG before it is evaluated.

What the reference's spartan/expr/operator/optimize.py does for the tile path: chains of element-wise maps -- and a
reduction over maps -- become ONE LocalExpr tree, i.e. one HIP kernel launch per tile (its MapMapFusion :133-187
and ReduceMapFusion :190-227); subtrees that were already evaluated are replaced by their values
(CollapsedCachedExpressions :230-247); slices are pushed below maps so that the maps above them can still fuse
(RotateSlice :393-456); the tiling of new arrays is chosen (AutomaticTiling, here expr/tiling.py).

Structure here: a rewrite is a `Pass` whose `rules` table maps an expression type to a method that receives the
node AFTER its dependencies were rewritten and returns its replacement; `Pass.visit` is the memoised bottom-up walk
(it is also the visitor `Expr.visit` calls back into).  The fused operator of a map is assembled by `_FusedInputs`,
which keeps one entry per distinct input variable -- the same array reaching a fused tree through two sub-maps is
read once by the kernel.

The observable results are the reference's, pinned by tests/golden/fusion_golden.json: the fused operator trees
print identically and have the same inputs.
"""
import importlib
import sys
import weakref

import numpy as np

from .base import AsArray, Expr, ListExpr, NotShapeable, Val, expr_like, lazify
from .local import LocalInput, LocalMapLocationExpr, LocalReduceExpr, make_var
from . import plan
from .map import MapExpr
from .ndarray import NdArrayExpr
from .reduce import ReduceExpr
from .shuffle import ShuffleExpr

FLAGS = {
    'optimization': True,
    'opt_collapse_cached': True,
    # On by default, as in the reference (optimize.py:1094).  The solver is pinned against the reference's own
    # (tests/golden/tiling_golden.json, tests/test_tiling.py); the pass never changes values -- the reference's,
    # run under Python 3, does (see tests/golden/make_golden.py: tiling_goldens), so the program goldens were
    # recorded with the reference's pass off and are compared with ours on.
    'opt_auto_tiling': True,
    'opt_rotate_slice': False,        # off by default in the reference as well (optimize.py:1095)
    'opt_map_fusion': True,
    'opt_reduce_fusion': True,
    # not a rewrite of the reference's: sum(x * (dot(x, w) - y), axis=0) in one pass over x (expr/rowdot.py), applied
    # only where the backend has the kernel
    'opt_rowdot_fusion': True,
    # not a rewrite: DAGs with the structure of one seen before take its recorded result (expr/plan.py)
    'opt_plan_cache': True,
    # not a rewrite: the maps of a DAG optimised for the first time are lowered to kernels here (see _prelower)
    'opt_prelower': True,
}

# Results of builders whose value differs from call to call (rand, ...) must be computed exactly once: such a node
# is never inlined into a consumer's kernel.  Keyed by expression id, which a rebuilt node keeps (the reference
# keys its list by id() of the Python object, optimize.py:60-68, so the rebuilt copy of such a map -- a different
# object -- IS inlined by its consumer there, and a recycled id() can stop an unrelated map from fusing).
_keep_apart = set()


def not_idempotent(fn):
  def build(*args, **kw):
    node = fn(*args, **kw)
    if isinstance(node, Expr):
      node.needs_cache = True
      _keep_apart.add(node.expr_id)
      weakref.finalize(node, _keep_apart.discard, node.expr_id)
    return node
  return build


def disable_parakeet(fn):
  return fn


def _inlinable(node):
  """A map whose operator tree may be merged into its consumer's."""
  return isinstance(node, MapExpr) and node.expr_id not in _keep_apart


# What may sit under a map that is being fused: other maps are inlined, these are read as inputs.
_PLAIN_INPUTS = (MapExpr, ReduceExpr, ShuffleExpr, NdArrayExpr, Val, AsArray)


class Pass(object):
  """Memoised bottom-up rewrite.  `rules`: {Expr subclass name: method name}; a rule gets the node with its
  dependencies already rewritten.  Without a rule the rebuilt node is kept."""
  name = None
  rules = {}

  def __init__(self):
    self._memo = {}

  def visit(self, node):
    if not isinstance(node, Expr):
      return node
    done = self._memo.get(node.expr_id)
    if done is None:
      done = self.rewrite(node)
      self._memo[node.expr_id] = done
    return done

  def rewrite(self, node):
    rebuilt = node.visit(self)                    # same node type and id, rewritten dependencies
    rule = self.rules.get(type(node).__name__)
    return getattr(self, rule)(node, rebuilt) if rule else rebuilt


class _FusedInputs(object):
  """Inputs of a fused operator: (variable name, array expression) pairs, one per distinct variable."""

  def __init__(self):
    self.names, self.arrays = [], []

  def take(self, name, array):
    if name in self.names:
      held = self.arrays[self.names.index(name)]
      same = held.expr_id == array.expr_id if isinstance(array, Expr) and isinstance(held, Expr) else held is array
      assert same, 'variable %s names two different inputs' % name
      return
    self.names.append(name)
    self.arrays.append(array)

  def inline(self, submap, into):
    """Make `submap`'s operator tree a dependency of `into`; its inputs become ours."""
    for name, array in zip(submap.child_to_var, submap.children):
      self.take(name, array)
    into.add_dep(submap.op)

  def feed(self, array, into):
    """`array` stays a kernel input under a fresh variable."""
    name = make_var()
    self.take(name, array)
    into.add_dep(LocalInput(idx=name))


class MapMapFusion(Pass):
  """map(f, map(g, x), y)  ->  map(f(g(.), .), x, y)"""
  name = 'map_fusion'
  rules = {'MapExpr': 'fuse'}

  def fuse(self, original, node):
    if original.expr_id in _keep_apart or not all(isinstance(c, _PLAIN_INPUTS) for c in node.children):
      return node
    op = type(node.op)(fn=node.op.fn, kw=node.op.kw, pretty_fn=node.op.pretty_fn)
    inputs = _FusedInputs()
    for child in node.children:
      if _inlinable(child):
        inputs.inline(child, op)
      else:
        inputs.feed(child, op)
    if isinstance(op, LocalMapLocationExpr):
      op.add_dep(LocalInput(idx='extent'))        # the tile's position is the operator's last argument
    return expr_like(node, children=ListExpr(vals=inputs.arrays), child_to_var=inputs.names, op=op)


class ReduceMapFusion(Pass):
  """reduce(f, map(g, x))  ->  reduce(f(g(.)), x): the map runs in the reduction kernel's prologue."""
  name = 'reduce_fusion'
  rules = {'ReduceExpr': 'fuse'}

  def fuse(self, original, node):
    if not all(_inlinable(c) for c in node.children):
      return node
    op = LocalReduceExpr(fn=node.op.fn, kw=node.op.kw, deps=[node.op.deps[0]])
    inputs = _FusedInputs()
    for child in node.children:
      inputs.inline(child, op)
    # (as in the reference, reduce.py:110, dtype_fn is afterwards applied to the fused tree's FIRST input)
    return expr_like(node, children=ListExpr(vals=inputs.arrays), child_to_var=inputs.names, op=op, axis=node.axis,
                     dtype_fn=node.dtype_fn, accumulate_fn=node.accumulate_fn, tile_hint=node.tile_hint)


class RowDotColSumFusion(Pass):
  """reduce(sum, axis=0, x * (dot(x, w) - y))  ->  one pass over the rows of x (expr/rowdot.py).  Runs after
  ReduceMapFusion, on the shape that pass leaves: a column sum whose fused operator is multiply(a, subtract(b, c))
  (or multiply(a, b)) over the inputs x, dot(x, w) = the map2 join with a driver-side vector, and y."""
  name = 'rowdot_fusion'
  rules = {'ReduceExpr': 'fuse'}

  def fuse(self, original, node):
    found = self.match(node)
    if found is None:
      return node
    from .rowdot import RowDotColSumExpr
    x, w, y = found
    return RowDotColSumExpr(expr_id=node.expr_id, x=x, y=y, w=w, tile_hint=node.tile_hint)

  @staticmethod
  def match(node):
    from .. import context
    from ..array import distarray
    from . import builtins
    from .local import LocalMapExpr
    from .map import Map2Expr
    dot_mod = sys.modules.get(__package__ + '.dot') or importlib.import_module(__package__ + '.dot')   # (the package
    # attribute `dot` is the builder, not the module)
    if not context.initialized() or getattr(context.get().backend, 'rowdot_colsum', None) is None:
      return None
    if node.axis != 0 or node.accumulate_fn is not np.add or node.op.fn is not builtins._sum_local:
      return None
    data = [d for d in node.op.deps if not isinstance(d, LocalInput)]
    if len(data) != 1 or not isinstance(data[0], LocalMapExpr) or data[0].fn is not np.multiply or len(data[0].deps) != 2:
      return None
    by_var = dict(zip(node.child_to_var, node.children))

    def leaf(d):
      return by_var.get(d.idx) if isinstance(d, LocalInput) else None

    def as_x(e):
      v = e.val if isinstance(e, Val) else None
      if not isinstance(v, distarray.DistArrayImpl) or v.sparse or len(v.shape) != 2 or v.dtype != np.float32:
        return None
      cols = v.shape[1]
      if cols < 4 or cols > 4096 or cols % 4 or not all(t.ul[1] == 0 and t.lr[1] == cols for t in v.tiles):
        return None
      return e

    def as_dot(e, x):
      if not isinstance(e, Map2Expr) or e.fn is not dot_mod.dot_map2_np_mapper or tuple(e.axes) != (0,):
        return None
      arrays = list(e.arrays.vals)
      w = (e.fn_kw or {}).get('array2')
      if len(arrays) != 1 or arrays[0].expr_id != x.expr_id or e.update_region is not None:
        return None
      if not (isinstance(w, np.ndarray) or dot_mod._is_backend_tensor(w)) or np.dtype(w.dtype) != np.float32 \
          or tuple(w.shape) not in ((x.val.shape[1], 1), (x.val.shape[1],)):
        return None
      return w

    def as_y(e, x):
      v = e.val if isinstance(e, Val) else None
      if not isinstance(v, distarray.DistArrayImpl) or v.sparse or v.dtype != np.float32 or tuple(v.shape) != (x.val.shape[0], 1):
        return None
      return e

    for a, b in (data[0].deps, data[0].deps[::-1]):
      x = leaf(a) and as_x(leaf(a))
      if not x:
        continue
      if isinstance(b, LocalInput):                      # x * dot(x, w)
        w = as_dot(leaf(b), x)
        if w is not None and len(w.shape) == 2:
          return x, w, None
      elif isinstance(b, LocalMapExpr) and b.fn is np.subtract and len(b.deps) == 2:   # x * (dot(x, w) - y)
        t, yv = leaf(b.deps[0]), leaf(b.deps[1])
        if t is None or yv is None:
          continue
        w = as_dot(t, x)
        y = as_y(yv, x)
        if w is not None and len(w.shape) == 2 and y is not None:
          return x, w, y
    return None


class CollapsedCachedExpressions(Pass):
  """A subtree whose value already exists is replaced by that value."""
  name = 'collapse_cached'

  def rewrite(self, node):
    value = node.cache()
    return lazify(value) if value is not None else node.visit(self)


class RotateSlice(Pass):
  """(a + b)[idx]  ->  a'[idx] + b'[idx], with a', b' = a, b seen in the map's shape: a slice ABOVE a map is pushed
  onto the map's inputs, recursively through chains of maps, so that the slice ends up on the arrays and the maps
  above it are adjacent again and fuse into one kernel over the SLICED extent only."""
  name = 'rotate_slice'

  def rewrite(self, node):
    from .views import SliceExpr
    if not isinstance(node, SliceExpr) or not isinstance(node.src, MapExpr):
      return node.visit(self)
    return self.push(node.src, node.idx)

  def push(self, mapped, idx):
    """`mapped[idx]` as a map over sliced inputs."""
    from .views import SliceExpr
    try:
      shape = mapped.compute_shape()
    except NotShapeable:
      return SliceExpr(src=self.visit(mapped), idx=idx, broadcast_to=None)
    sliced = []
    for child in mapped.children:
      if isinstance(child, MapExpr):
        sliced.append(self.push_through_broadcast(child, idx, shape))
      else:
        sliced.append(SliceExpr(src=self.visit(child), idx=idx, broadcast_to=shape))
    # always a NEW node: it has the slice's shape, not the map's, so neither the map's id (its cached value)
    # nor its shape cache may be carried over (the reference re-uses the id for the first slice of a map)
    return MapExpr(children=ListExpr(vals=sliced), child_to_var=list(mapped.child_to_var), op=mapped.op)

  def push_through_broadcast(self, child_map, idx, shape):
    """A map under the sliced map: slice it further down when it already has the parent's shape; a map that is
    broadcast by its parent keeps its own extent and is sliced as a whole (through the broadcast)."""
    from .views import SliceExpr
    try:
      same = tuple(child_map.compute_shape()) == tuple(shape)
    except NotShapeable:
      same = False
    if same:
      return self.push(child_map, idx)
    return SliceExpr(src=self.visit(child_map), idx=idx, broadcast_to=shape)


def _passes():
  from .tiling import AutomaticTiling   # (imports this module)
  return (('opt_collapse_cached', CollapsedCachedExpressions), ('opt_auto_tiling', AutomaticTiling),
          ('opt_rotate_slice', RotateSlice), ('opt_map_fusion', MapMapFusion), ('opt_reduce_fusion', ReduceMapFusion),
          ('opt_rowdot_fusion', RowDotColSumFusion))


def _run_passes(dag):
  for flag, factory in _passes():
    if FLAGS[flag]:
      dag = factory().visit(dag)
  return dag


def optimize(dag):
  """Apply the enabled passes in the reference's order (optimize.py:1093-1099).  A DAG with the structure of one
  that was optimised before is answered from the plan table (expr/plan.py): the recorded result, instantiated over
  this DAG's leaves."""
  if not FLAGS['optimization']:
    return dag
  if not FLAGS['opt_plan_cache']:
    return _run_passes(dag)
  first = plan.stats['misses']
  out = plan.optimized(dag, tuple(FLAGS.values()), _run_passes)
  if plan.stats['misses'] != first and FLAGS['opt_prelower']:
    _prelower(out)
  return out


def _prelower(dag):
  """A DAG optimised for the first time: its maps over arrays that exist already get their kernels lowered now
  (expr/map.py prelower -> backend.prelower_map), where the reference generates the code of its fused operators
  (optimize.py:1023-1076), instead of inside the first evaluation.  Later DAGs of the same structure come from the
  plan table and find the programs in the backend's."""
  from .map import prelower as prelower_map
  from .reduce import ReduceExpr, prelower as prelower_reduce
  from .base import CollectionExpr
  from .. import context
  ctx = context.get() if context.initialized() else None
  if ctx is None or getattr(ctx.backend, 'prelower_map', None) is None:
    return
  seen = set()

  def walk(v):
    if isinstance(v, Expr):
      if id(v) in seen:
        return
      seen.add(id(v))
      if type(v) is MapExpr:
        prelower_map(v, ctx)
      elif type(v) is ReduceExpr:
        prelower_reduce(v, ctx)
      if isinstance(v, CollectionExpr):
        for x in (v.vals.values() if isinstance(v.vals, dict) else v.vals):
          walk(x)
      else:
        for name in v.members:
          walk(getattr(v, name))
    elif isinstance(v, (list, tuple)):
      for x in v:
        walk(x)
  walk(dag)
