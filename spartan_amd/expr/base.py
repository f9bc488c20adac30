"""Expr: the lazy-DAG node base class, the evaluation cache and the collection
nodes.  Mirror of the reference's spartan/expr/operator/base.py (same names and
evaluate/cache/optimized/glom semantics); `force()` is the alias of `evaluate()`
that README.md:64-65 promises.
"""
import collections
import itertools

import numpy as np

from .. import context
from ..array import distarray
from ..util import Assert

unique_id = itertools.count()


class NotShapeable(Exception):
  """base.py:30-34."""


class newaxis(object):
  pass


class EvalCache(object):
  """base.py:73-114: results keyed by expression id, manually refcounted."""

  def __init__(self):
    self.refs = collections.defaultdict(int)
    self.cache = {}

  def set(self, exprid, value):
    self.cache[exprid] = value

  def get(self, exprid):
    return self.cache.get(exprid, None)

  def register(self, exprid):
    self.refs[exprid] += 1

  def deregister(self, expr_id):
    self.refs[expr_id] -= 1
    if self.refs[expr_id] == 0:
      if expr_id in self.cache:
        del self.cache[expr_id]
      del self.refs[expr_id]

  def clear(self):
    self.refs.clear()
    self.cache.clear()


eval_cache = EvalCache()


def expr_like(expr, **kw):
  """base.py:51-69: same expression id (so cache entries carry over)."""
  kw['expr_id'] = expr.expr_id
  kw['shape_cache'] = expr.shape_cache
  return expr.__class__(**kw)


class Expr(object):
  """base.py:163-505.  Subclasses list their dependency fields in `members`."""
  members = ()
  needs_cache = True

  def __init__(self, expr_id=None, shape_cache=None, **kw):
    for k in self.members:
      setattr(self, k, kw.pop(k, None))
    if kw:
      raise TypeError('%s: unexpected fields %s' % (type(self).__name__, list(kw)))
    self.expr_id = next(unique_id) if expr_id is None else expr_id
    self.shape_cache = shape_cache
    self.optimized_expr = None
    eval_cache.register(self.expr_id)

  def __del__(self):
    try:
      eval_cache.deregister(self.expr_id)
    except Exception:
      pass

  @property
  def ndim(self):
    return len(self.shape)

  def cache(self):
    """base.py:193-203."""
    result = eval_cache.get(self.expr_id)
    if result is not None and len(getattr(result, 'bad_tiles', ())) == 0:
      return result
    return self.load_data(result)

  def load_data(self, cached_result):
    """base.py:189-191: nothing to reload -- an expression whose cached value lost tiles is evaluated again from
    its dependencies (CheckpointExpr overrides this with a reload from disk)."""
    return None

  def dependencies(self):
    return dict([(k, getattr(self, k)) for k in self.members])

  def compute_shape(self):
    raise NotShapeable

  def visit(self, visitor):
    deps = {}
    for k in self.members:
      deps[k] = visitor.visit(getattr(self, k))
    return expr_like(self, **deps)

  def __repr__(self):
    return self.pretty_str()

  def pretty_str(self):
    return '%s[%d]' % (type(self).__name__, self.expr_id)

  def typename(self):
    return self.__class__.__name__

  def evaluate(self):
    """base.py:272-313: dependencies first, then `_evaluate`, then cache."""
    ctx = context.get()
    if ctx.heartbeat is not None and ctx.current_worker is None:
      ctx.apply_failures()          # safe point: workers the heartbeat declared silent lose their tiles here
    cache = self.cache()
    if cache is not None:
      return cache
    deps = {}
    for k, vs in self.dependencies().items():
      if isinstance(vs, Expr):
        deps[k] = vs.evaluate()
      else:
        deps[k] = vs
    value = self._evaluate(ctx, deps)
    if self.needs_cache:
      eval_cache.set(self.expr_id, value)
    return value

  def force(self):
    """README.md:64-65: expressions are "forced" -- alias of evaluate()."""
    return self.evaluate()

  def _evaluate(self, ctx, deps):
    raise NotImplementedError

  def __hash__(self):
    return self.expr_id

  # -- operators (base.py:331-388) -> map(np.ufunc) -------------------------------
  def __add__(self, other): return _map(self, other, fn=np.add)
  def __sub__(self, other): return _map(self, other, fn=np.subtract)
  def __mul__(self, other): return _map(self, other, fn=np.multiply)
  def __mod__(self, other): return _map(self, other, fn=np.mod)
  def __truediv__(self, other): return _map(self, other, fn=np.divide)
  __div__ = __truediv__
  def __floordiv__(self, other): return _map(self, other, fn=np.floor_divide)
  def __eq__(self, other): return _map(self, other, fn=np.equal)
  def __ne__(self, other): return _map(self, other, fn=np.not_equal)
  def __lt__(self, other): return _map(self, other, fn=np.less)
  def __le__(self, other): return _map(self, other, fn=np.less_equal)
  def __gt__(self, other): return _map(self, other, fn=np.greater)
  def __ge__(self, other): return _map(self, other, fn=np.greater_equal)
  def __and__(self, other): return _map(self, other, fn=np.logical_and)
  def __or__(self, other): return _map(self, other, fn=np.logical_or)
  def __xor__(self, other): return _map(self, other, fn=np.logical_xor)
  def __pow__(self, other): return _map(self, other, fn=np.power)
  def __neg__(self): return _map(self, fn=np.negative)
  def __rsub__(self, other): return _map(other, self, fn=np.subtract)
  def __radd__(self, other): return _map(other, self, fn=np.add)
  def __rmul__(self, other): return _map(other, self, fn=np.multiply)
  def __rtruediv__(self, other): return _map(other, self, fn=np.divide)
  __rdiv__ = __rtruediv__

  def __setitem__(self, k, val):
    raise Exception('Expressions are read-only.')

  @property
  def shape(self):
    """base.py:452-471."""
    cache = self.cache()
    if cache is not None:
      return cache.shape
    if self.shape_cache is None:
      try:
        self.shape_cache = tuple(self.compute_shape())
      except NotShapeable:
        self.shape_cache = evaluate(self).shape
    return self.shape_cache

  @property
  def size(self):
    return int(np.prod(self.shape, dtype=np.int64))

  def optimized(self):
    """base.py:477-492 (fusion is opt-in, as in the reference)."""
    # The reference makes the optimised node point at itself (base.py:489), a
    # reference cycle that only the cyclic GC can free -- which would keep multi-GiB
    # HBM tiles of dead results alive between collections.  A flag has the same
    # effect (optimising an optimised node is the identity) without the cycle.
    if getattr(self, '_is_optimized', False):
      return self
    if self.optimized_expr is None:
      self.optimized_expr = optimized_dag(self)
      self.optimized_expr._is_optimized = True
    return self.optimized_expr

  def glom(self):
    return glom(self)


def _map(*args, **kw):
  """base.py:39-48."""
  fn = kw['fn']
  from .map import map
  return map(args, fn)


class AsArray(Expr):
  """base.py:508-533."""
  members = ('val',)

  def visit(self, visitor):
    return self

  def dependencies(self):
    return {'val': self.val}

  def compute_shape(self):
    if hasattr(self.val, 'shape'):
      return self.val.shape
    if np.isscalar(self.val):
      return np.asarray(self.val).shape
    raise NotShapeable

  def _evaluate(self, ctx, deps):
    return distarray.as_array(deps['val'])

  def pretty_str(self):
    return str(self.val)


class Val(Expr):
  """base.py:536-557."""
  members = ('val',)
  needs_cache = False

  def visit(self, visitor):
    return self

  def dependencies(self):
    return {}

  def compute_shape(self):
    return self.val.shape

  def _evaluate(self, ctx, deps):
    return self.val

  def pretty_str(self):
    return str(self.val)


class CollectionExpr(Expr):
  """base.py:560-577."""
  members = ('vals',)
  needs_cache = False

  def __getitem__(self, idx):
    return self.vals[idx]

  def __iter__(self):
    return iter(self.vals)

  def __len__(self):
    return len(self.vals)

  def compute_shape(self):
    raise NotShapeable


class DictExpr(CollectionExpr):
  def dependencies(self):
    return self.vals

  def _evaluate(self, ctx, deps):
    return deps

  def visit(self, visitor):
    return DictExpr(vals=dict([(k, visitor.visit(v)) for (k, v) in self.vals.items()]))

  def items(self):
    return self.vals.items()

  def pretty_str(self):
    return '{ %s } ' % ',\n'.join(['%s : %s' % (k, repr(v)) for k, v in self.vals.items()])


class ListExpr(CollectionExpr):
  def dependencies(self):
    return dict(('v%d' % i, self.vals[i]) for i in range(len(self.vals)))

  def pretty_str(self):
    return '[\n%s\n]' % ','.join([v.pretty_str() if isinstance(v, Expr) else str(v) for v in self.vals])

  def _evaluate(self, ctx, deps):
    return [deps['v%d' % i] for i in range(len(self.vals))]

  def visit(self, visitor):
    return ListExpr(vals=[visitor.visit(v) for v in self.vals])


class TupleExpr(CollectionExpr):
  def dependencies(self):
    return dict(('v%d' % i, self.vals[i]) for i in range(len(self.vals)))

  def pretty_str(self):
    return '( %s )' % ','.join([v.pretty_str() if isinstance(v, Expr) else str(v) for v in self.vals])

  def _evaluate(self, ctx, deps):
    return tuple(deps['v%d' % i] for i in range(len(self.vals)))

  def visit(self, visitor):
    return TupleExpr(vals=tuple([visitor.visit(v) for v in self.vals]))


def glom(value):
  """base.py:652-662."""
  if isinstance(value, Expr):
    value = evaluate(value)
  if isinstance(value, np.ndarray):
    return value
  return value.glom()


def optimized_dag(node):
  if not isinstance(node, Expr):
    raise TypeError
  from .optimize import optimize
  return optimize(node)


def evaluate(node):
  """base.py:678-688."""
  if isinstance(node, Expr):
    return node.evaluate()
  Assert.isinstance(node, (np.ndarray, distarray.DistArray))
  return node


def eager(node):
  return Val(val=evaluate(node))


def lazify(val):
  """base.py:701-722."""
  if isinstance(val, Expr):
    return val
  if isinstance(val, dict):
    return DictExpr(vals=val)
  if isinstance(val, list):
    return ListExpr(vals=val)
  if isinstance(val, tuple):
    return TupleExpr(vals=val)
  return Val(val=val)


def as_array(v):
  """base.py:725-734."""
  if isinstance(v, Expr):
    return v
  return AsArray(val=v)
