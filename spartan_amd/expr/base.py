"""Nodes of the lazy expression DAG, the value cache, and the containers that hold several nodes.

API of the reference's spartan/expr/operator/base.py: `Expr` with evaluate / optimized / glom / shape / cache, the
`Val` / `AsArray` leaves, `ListExpr` / `TupleExpr` / `DictExpr`, `expr_like`, and the module functions `evaluate`,
`eager`, `glom`, `lazify`, `as_array`.  `force()` is `evaluate()` (README.md:64-65 of the reference calls the act
of evaluating "forcing").

How a node works here:
  * a subclass lists its fields in `members`; the constructor takes exactly those as keywords;
  * `dependencies()` says which fields hold sub-expressions -- `evaluate()` evaluates them first, hands the results
    to `_evaluate(ctx, deps)` and remembers the value in `eval_cache` under the node's `expr_id`;
  * `visit(visitor)` rebuilds the node from visited fields with THE SAME id (`expr_like`): a rewritten DAG finds
    the values its unrewritten twin already computed;
  * a cached value that lost tiles to a failed worker does not count (`cache()` -> `load_data()`), so the node is
    evaluated again -- or re-read, for a checkpoint.
"""
import itertools

import numpy as np

from .. import context
from ..array import distarray

unique_id = itertools.count()


class NotShapeable(Exception):
  """The shape of this node is only known once it has been evaluated."""


class newaxis(object):
  pass


class EvalCache(object):
  """Values of evaluated nodes by expression id.  An id may be shared by several node objects (a node and its
  rewritten copies): the value goes away when the last of them does."""

  def __init__(self):
    self._holders = {}     # expr id -> number of live node objects carrying it
    self._values = {}

  def register(self, expr_id):
    self._holders[expr_id] = self._holders.get(expr_id, 0) + 1

  def deregister(self, expr_id):
    left = self._holders.get(expr_id, 0) - 1
    if left > 0:
      self._holders[expr_id] = left
    else:
      self._holders.pop(expr_id, None)
      self._values.pop(expr_id, None)

  def set(self, expr_id, value):
    self._values[expr_id] = value

  def get(self, expr_id):
    return self._values.get(expr_id)

  def clear(self):
    self._holders.clear()
    self._values.clear()


eval_cache = EvalCache()


def expr_like(expr, **fields):
  """A node of `expr`'s type built from `fields`, with expr's id and shape cache."""
  return type(expr)(expr_id=expr.expr_id, shape_cache=expr.shape_cache, **fields)


class Expr(object):
  members = ()
  needs_cache = True

  def __init__(self, expr_id=None, shape_cache=None, **fields):
    # (a driver loop builds and rewrites some fifty nodes per forced expression: plain dict stores, no set algebra)
    d = self.__dict__
    pop = fields.pop
    for name in self.members:
      d[name] = pop(name, None)
    if fields:
      raise TypeError('%s: unexpected fields %s' % (type(self).__name__, sorted(fields)))
    d['expr_id'] = expr_id = next(unique_id) if expr_id is None else expr_id
    d['shape_cache'] = shape_cache
    d['optimized_expr'] = None
    holders = eval_cache._holders
    holders[expr_id] = holders.get(expr_id, 0) + 1

  def __del__(self):
    try:
      eval_cache.deregister(self.expr_id)
    except Exception:      # interpreter shutdown
      pass

  # -- structure ---------------------------------------------------------------------------------------------
  def dependencies(self):
    return {name: getattr(self, name) for name in self.members}

  def visit(self, visitor):
    return expr_like(self, **{name: visitor.visit(getattr(self, name)) for name in self.members})

  def typename(self):
    return type(self).__name__

  def pretty_str(self):
    return '%s[%d]' % (self.typename(), self.expr_id)

  __repr__ = lambda self: self.pretty_str()

  def __hash__(self):
    return self.expr_id

  # -- values ------------------------------------------------------------------------------------------------
  def cache(self):
    """The node's value if it was computed and is intact; else whatever `load_data` can restore; else None."""
    value = eval_cache.get(self.expr_id)
    if value is not None and not getattr(value, 'bad_tiles', None):
      return value
    return self.load_data(value)

  def load_data(self, damaged):
    """Nothing to restore from: the node is evaluated again from its dependencies (CheckpointExpr reloads)."""
    return None

  def evaluate(self):
    ctx = context.get()
    if ctx.eval_depth == 0:
      ctx.eval_epoch += 1           # one number per top-level evaluation (what the tiles of one evaluation share)
      if ctx.heartbeat is not None and ctx.current_worker is None:
        ctx.apply_failures()        # safe point (once per top-level evaluation): workers declared silent lose their tiles
      pend = ctx.pending_destructors
      if pend and ctx.current_worker is None:
        # tiles of arrays that died since the last evaluation go back to the tile store BEFORE this one allocates
        # its results (the reference frees them when the next array is registered, distarray.py:249-268 -- here
        # that is after the new tiles exist: a loop that keeps one result alive then held three sets of tiles and
        # paid a fresh allocation of the third, 160 ms per 2 GiB tile)
        ctx.destroy_all(pend)
        del pend[:]
    ctx.eval_depth += 1
    try:
      value = self.cache()
      if value is None:
        ready = {k: (d.evaluate() if isinstance(d, Expr) else d) for k, d in self.dependencies().items()}
        value = self._evaluate(ctx, ready)
        if self.needs_cache:
          eval_cache._values[self.expr_id] = value
    finally:
      ctx.eval_depth -= 1
    return value

  def force(self):
    return self.evaluate()

  def _evaluate(self, ctx, deps):
    raise NotImplementedError

  def glom(self):
    return glom(self)

  def optimized(self):
    """The node after the DAG rewrites (fusion is opt-in, as in the reference: `evaluate` runs the DAG as built).
    Optimising an optimised node is the identity -- remembered with a flag rather than the reference's
    self-reference (base.py:489), a cycle only the cyclic collector frees, which would keep the multi-GiB tiles
    of dead results in HBM between collections."""
    if getattr(self, '_is_optimized', False):
      return self
    if self.optimized_expr is None:
      self.optimized_expr = optimized_dag(self)
      self.optimized_expr._is_optimized = True
    return self.optimized_expr

  # -- shape -------------------------------------------------------------------------------------------------
  def compute_shape(self):
    raise NotShapeable

  @property
  def shape(self):
    value = self.cache()
    if value is not None:
      return value.shape
    if self.shape_cache is None:
      try:
        self.shape_cache = tuple(self.compute_shape())
      except NotShapeable:
        self.shape_cache = evaluate(self).shape
    return self.shape_cache

  @property
  def ndim(self):
    return len(self.shape)

  @property
  def size(self):
    return int(np.prod(self.shape, dtype=np.int64))

  def __setitem__(self, k, val):
    raise Exception('Expressions are read-only.')


_map_builder = []        # expr.map.map (that module imports this one: bound on first use, not per call)


def _elementwise(fn, *operands):
  if not _map_builder:
    from .map import map
    _map_builder.append(map)
  return _map_builder[0](operands, fn)


def _install_operators():
  """a + b, -a, 2 * a ... build element-wise maps over NumPy ufuncs (the reference spells out one method each,
  base.py:331-388)."""
  binary = {'add': np.add, 'sub': np.subtract, 'mul': np.multiply, 'mod': np.mod, 'truediv': np.divide,
            'floordiv': np.floor_divide, 'pow': np.power, 'eq': np.equal, 'ne': np.not_equal, 'lt': np.less,
            'le': np.less_equal, 'gt': np.greater, 'ge': np.greater_equal, 'and': np.logical_and,
            'or': np.logical_or, 'xor': np.logical_xor}
  for name, fn in binary.items():
    setattr(Expr, '__%s__' % name, (lambda f: lambda self, other: _elementwise(f, self, other))(fn))
  for name in ('add', 'sub', 'mul', 'truediv'):
    setattr(Expr, '__r%s__' % name, (lambda f: lambda self, other: _elementwise(f, other, self))(binary[name]))
  Expr.__div__, Expr.__rdiv__ = Expr.__truediv__, Expr.__rtruediv__        # the reference's Python-2 names
  Expr.__neg__ = lambda self: _elementwise(np.negative, self)
  Expr.__hash__ = lambda self: self.expr_id                                 # (defining __eq__ would drop it)


_install_operators()


# ---- leaves ----------------------------------------------------------------------------------------------------
class _Leaf(Expr):
  members = ('val',)

  def visit(self, visitor):
    return self

  def pretty_str(self):
    return str(self.val)


class AsArray(_Leaf):
  """A driver-side operand (NumPy array or scalar) taking part in a map."""

  def compute_shape(self):
    if hasattr(self.val, 'shape'):
      return self.val.shape
    if np.isscalar(self.val):
      return ()
    raise NotShapeable

  def _evaluate(self, ctx, deps):
    return distarray.as_array(deps['val'])


class Val(_Leaf):
  """An existing value (a distributed array, usually) as a node; nothing to evaluate, nothing to cache."""
  needs_cache = False

  def dependencies(self):
    return {}

  def compute_shape(self):
    return self.val.shape

  def evaluate(self):
    # (inside an evaluation a leaf is its value: no cache entry to look for, no dependencies, nothing to record; at
    #  top level the general entry runs, for its safe-point duties)
    if context.get().eval_depth:
      return self.val
    return Expr.evaluate(self)

  def _evaluate(self, ctx, deps):
    return self.val

  def evaluate(self):
    # nothing to compute and nothing to remember; as the root of an evaluation it is still a safe point
    if context.get().eval_depth == 0:
      return Expr.evaluate(self)
    return self.val


# ---- containers ------------------------------------------------------------------------------------------------
class CollectionExpr(Expr):
  """Several nodes as one dependency.  Sequence containers share everything but their Python type."""
  members = ('vals',)
  needs_cache = False
  _ctor = None
  _brackets = ('', '')

  def __getitem__(self, idx):
    return self.vals[idx]

  def __iter__(self):
    return iter(self.vals)

  def __len__(self):
    return len(self.vals)

  def dependencies(self):
    return {'v%d' % i: v for i, v in enumerate(self.vals)}

  def _evaluate(self, ctx, deps):
    return self._ctor(deps['v%d' % i] for i in range(len(self.vals)))

  def evaluate(self):
    # (a container inside a DAG: its members' values, without the bookkeeping of a node that has a value of its own)
    if type(self.vals) is dict or context.get().eval_depth == 0:
      return Expr.evaluate(self)
    return self._ctor([v.evaluate() if isinstance(v, Expr) else v for v in self.vals])

  def visit(self, visitor):
    return type(self)(vals=self._ctor(visitor.visit(v) for v in self.vals))

  def pretty_str(self):
    inner = ','.join(v.pretty_str() if isinstance(v, Expr) else str(v) for v in self.vals)
    return self._brackets[0] + inner + self._brackets[1]


class ListExpr(CollectionExpr):
  _ctor = list
  _brackets = ('[\n', '\n]')


class TupleExpr(CollectionExpr):
  _ctor = tuple
  _brackets = ('( ', ' )')


class DictExpr(CollectionExpr):
  def dependencies(self):
    return self.vals

  def items(self):
    return self.vals.items()

  def _evaluate(self, ctx, deps):
    return deps

  def visit(self, visitor):
    return DictExpr(vals={k: visitor.visit(v) for k, v in self.vals.items()})

  def pretty_str(self):
    return '{ %s } ' % ',\n'.join('%s : %r' % kv for kv in self.vals.items())


_optimize_fn = []        # expr.optimize.optimize, bound on first use (that module imports this one)


# ---- module functions ----------------------------------------------------------------------------------------
def optimized_dag(node):
  if not isinstance(node, Expr):
    raise TypeError('optimized_dag of %r' % type(node))
  if not _optimize_fn:
    from .optimize import optimize
    _optimize_fn.append(optimize)
  return _optimize_fn[0](node)


def evaluate(node):
  if isinstance(node, Expr):
    return node.evaluate()
  if not isinstance(node, (np.ndarray, distarray.DistArray)):
    raise AssertionError('%r is neither an expression nor an array' % (node,))
  return node


def eager(node):
  return Val(val=evaluate(node))


def glom(value):
  """The value of an expression (or array) as one host array."""
  value = evaluate(value) if isinstance(value, Expr) else value
  return value if isinstance(value, np.ndarray) else value.glom()


_CONTAINER_OF = ((dict, DictExpr), (list, ListExpr), (tuple, TupleExpr))


def lazify(value):
  """Anything as a node: expressions as they are, Python containers as container nodes, the rest as Val."""
  if isinstance(value, Expr):
    return value
  for python_type, node_type in _CONTAINER_OF:
    if isinstance(value, python_type):
      return node_type(vals=value)
  return Val(val=value)


def as_array(value):
  return value if isinstance(value, Expr) else AsArray(val=value)
