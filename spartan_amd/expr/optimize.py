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
import weakref

from .base import AsArray, Expr, ListExpr, NotShapeable, Val, expr_like, lazify
from .local import LocalInput, LocalMapLocationExpr, LocalReduceExpr, make_var
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
          ('opt_rotate_slice', RotateSlice), ('opt_map_fusion', MapMapFusion), ('opt_reduce_fusion', ReduceMapFusion))


def optimize(dag):
  """Apply the enabled passes in the reference's order (optimize.py:1093-1099)."""
  if not FLAGS['optimization']:
    return dag
  for flag, factory in _passes():
    if FLAGS[flag]:
      dag = factory().visit(dag)
  return dag
