"""Plans for repeated DAGs: an iterative driver builds the SAME expression over and over -- other arrays, other
driver-side operands, the same structure -- and there is no reason to derive the same optimised DAG each time.

The reference caches optimised expressions for that reason (spartan/expr/operator/optimize.py:70-76, a weak
dictionary per pass keyed by node; spartan/expr/operator/base.py:73-114 keeps evaluated values by expression id);
a driver loop that rebuilds its expression misses both.  Here the key is the STRUCTURE of the DAG as built:

    signature(dag) -> (key, leaves, ids)

`key` names everything the rewrites may look at -- node types, operator trees (functions by identity, keyword values,
variable names by order of appearance), shapes / dtypes / tile layouts of the arrays at the leaves, dtype and shape of
driver-side operands, Python scalars by value, the pass flags and the number of workers; `leaves` are the things that
may differ between two DAGs with one key (leaf nodes, and arrays that sit inside keyword dictionaries); `ids` the
expression ids of the inner nodes in walk order.  A DAG whose key was seen before is answered by INSTANTIATING the
optimised DAG recorded then: its nodes are rebuilt with the new leaves in the old one's places, inner nodes that
kept their id through the rewrites (`expr_like`) get the id of the node at the same position of the new DAG, so
values computed for one twin are found by the other exactly as without the cache.  Operator trees are shared
between instances (nothing mutates them once the rewrites are done), which is what lets the backend key ITS cache
of lowered programs by the operator object.

Anything the walk cannot describe makes the DAG unplannable and it is optimised the long way: a node that already
has a value (the collapse pass would cut the DAG there), results of non-idempotent builders (kept apart by id),
leaf values other than tiled arrays / NumPy arrays / scalars, objects inside fields it does not know.
"""
import collections

import numpy as np

from . import base
from .base import AsArray, CollectionExpr, DictExpr, Expr, ListExpr, Val, eval_cache
from .local import FnCallExpr, LocalExpr, LocalInput
from .. import context
from ..array import distarray, extent

MAX_PLANS = 256
_plans = collections.OrderedDict()        # key -> Plan
stats = {'hits': 0, 'misses': 0, 'unplannable': 0}

_SIMPLE = (type(None), bool, int, float, str, bytes, complex)
_keep_apart_ref = []
_MapExpr = [None]                          # expr.map.MapExpr (map imports this module's users: set on first use)
_layouts = {}                              # tile table -> small int


class Unplannable(Exception):
  pass


def clear():
  _plans.clear()
  _layouts.clear()


def layout_token(array):
  """A small int that two DistArrayImpl share iff they are cut into the same tiles on the same workers."""
  tok = getattr(array, '_layout_token', None)
  if tok is None:
    table = tuple((ex.ul, ex.lr, tid.worker) for ex, tid in array.tiles.items())
    tok = _layouts.setdefault(table, len(_layouts))
    array._layout_token = tok
  return tok


def _scalar_key(v):
  """A Python / NumPy scalar by VALUE as the kernels see it: floats by their bits (0.0 and -0.0 are two operands, and
  a NaN equals itself here)."""
  if isinstance(v, float):
    return v.hex()
  if isinstance(v, complex):
    return (v.real.hex(), v.imag.hex())
  return v


def _describe_value(v, pins):
  """Key of a leaf VALUE (what the rewrites may depend on, not the data)."""
  if isinstance(v, distarray.DistArrayImpl):
    if v.bad_tiles:
      raise Unplannable('array with bad tiles')
    red = v.reducer_fn
    if red is not None:
      pins.append(red)
    return ('D', v.shape, v.dtype.str, bool(v.sparse), layout_token(v), None if red is None else id(red),
            v.written is None)
  if isinstance(v, np.ndarray):
    return ('N', v.shape, v.dtype.str)
  if hasattr(v, 'data_ptr') and hasattr(v, 'strides'):       # a backend tensor used as a driver-side operand
    return ('B', tuple(v.shape), np.dtype(v.dtype).str)
  if isinstance(v, np.generic):
    return ('G', v.dtype.str, _scalar_key(v.item()))
  if isinstance(v, _SIMPLE):
    return ('S', type(v).__name__, _scalar_key(v))
  raise Unplannable('leaf value of type %s' % type(v).__name__)


class _Walk(object):
  def __init__(self):
    self.leaves = []          # leaf nodes / leaf values, in walk order
    self.ids = []             # expr ids of inner nodes, in walk order
    self.pins = []            # objects named by identity in the key (kept alive with the plan)
    self.seen = {}            # id(node) -> position (a node reached twice is one node)
    self.values = {}          # id(array at a leaf) -> order of appearance (the same array behind two leaves)
    self.names = {}           # variable name -> order of appearance
    self.in_local = 0         # > 0 while inside an operator tree (those are SHARED between the instances of a plan)

  def alias(self, value):
    if not isinstance(value, (np.ndarray, distarray.DistArray)):
      return None
    return self.values.setdefault(id(value), len(self.values))

  def var(self, name):
    n = self.names.get(name)
    if n is None:
      n = self.names[name] = len(self.names)
    return n

  def node(self, e):
    seen = self.seen
    pos = seen.get(id(e))
    if pos is not None:
      return ('@', pos)
    seen[id(e)] = len(seen)
    t = type(e)
    if t is _MapExpr[0]:
      return self.map_node(e)
    if t is Val or t is AsArray or isinstance(e, (Val, AsArray)):
      self.leaves.append(e)
      return ('V' if isinstance(e, Val) else 'A', _describe_value(e.val, self.pins), self.alias(e.val))
    if e.expr_id in eval_cache._values:
      raise Unplannable('a node already has a value')
    self.ids.append(e.expr_id)
    field = self.field
    if isinstance(e, CollectionExpr):
      if isinstance(e, DictExpr):
        return ('DictExpr', tuple([(k, field(v)) for k, v in sorted(e.vals.items(), key=_by_name)]))
      return (t, tuple([field(v) for v in e.vals]))
    d = e.__dict__
    return (t, tuple([field(d[name]) for name in e.members]), bool(e.needs_cache))

  def map_node(self, e):
    """node() for a MapExpr (already entered into `seen`): the nodes a driver loop's operators build -- a list of
    children, one name per child, an operator call over those names without keywords -- described in ONE pass, with
    the bookkeeping of the general walk in the general walk's order (children list and children first, then the
    names, then the operator's variables); anything else about the node goes the general way."""
    values = eval_cache._values
    if e.expr_id in values:
      raise Unplannable('a node already has a value')
    self.ids.append(e.expr_id)
    kids, names, op = e.children, e.child_to_var, e.op
    seen = self.seen
    if (type(kids) is ListExpr and type(names) is list and isinstance(op, FnCallExpr) and not op.kw
            and id(kids) not in seen and kids.expr_id not in values):
      deps = op.deps
      plain = True
      for d in deps:
        if type(d) is not LocalInput:
          plain = False
          break
      for n in names:
        if type(n) is not str:
          plain = False
          break
      if plain:
        seen[id(kids)] = len(seen)
        self.ids.append(kids.expr_id)
        node, var = self.node, self.var
        ck = tuple([node(c) for c in kids.vals])
        nk = tuple([var(n) if n.startswith('key_') else n for n in names])
        self.pins.append(op.fn)
        dk = tuple([var(d.idx) if d.idx.startswith('key_') else d.idx for d in deps])
        return ('M!', type(op), id(op.fn), ck, nk, dk, bool(e.needs_cache))
    field, d = self.field, e.__dict__
    return (type(e), tuple([field(d[name]) for name in e.members]), bool(e.needs_cache))

  def local(self, op):
    if type(op) is LocalInput:
      idx = op.idx
      return ('i', self.var(idx) if idx.startswith('key_') else idx)
    if isinstance(op, FnCallExpr):
      self.pins.append(op.fn)
      local = self.local
      # An operator tree is shared by every instance of the plan (the backend keys its lowered programs by the
      # operator object), so nothing inside it may differ between two DAGs with one key: scalars are in the key
      # by value; an ARRAY among the keyword values (fn_kw={'w': array}) has no place in the key and cannot be
      # swapped per instance -- such a DAG is optimised the long way every time.
      self.in_local += 1
      try:
        kw = self.field(op.kw) if op.kw else None
      finally:
        self.in_local -= 1
      return (type(op), id(op.fn), kw, tuple([local(d) for d in op.deps]))
    if isinstance(op, LocalInput):
      return ('i', self.var(op.idx) if op.idx.startswith('key_') else op.idx)
    raise Unplannable('local expression %s' % type(op).__name__)

  def field(self, v):
    t = type(v)
    if t in _SIMPLE_SET:
      if t is str and v.startswith('key_'):
        return ('v', self.var(v))
      if t is float or t is complex:
        return (t, _scalar_key(v))
      return (t, v)
    if isinstance(v, Expr):
      return self.node(v)
    if isinstance(v, LocalExpr):
      return self.local(v)
    if t is tuple or t is list:
      field = self.field
      return (t, tuple([field(x) for x in v]))
    if t is dict:
      return ('dict', tuple([(k, self.field(x)) for k, x in sorted(v.items(), key=_by_name)]))
    if isinstance(v, (np.ndarray, distarray.DistArrayImpl)) or (hasattr(v, 'data_ptr') and hasattr(v, 'strides')):
      if self.in_local:
        raise Unplannable('an array inside the keywords of an operator tree')
      first = self.seen.get(id(v))
      if first is not None:
        return ('@', first)                 # the same object again: the recipe maps it to one slot
      self.seen[id(v)] = len(self.seen)
      self.leaves.append(v)
      return ('L', _describe_value(v, self.pins), self.alias(v))
    if isinstance(v, _SIMPLE):
      return (t, _scalar_key(v))
    if isinstance(v, (tuple, list)):
      return (t, tuple([self.field(x) for x in v]))
    if isinstance(v, dict):
      return ('dict', tuple([(k, self.field(x)) for k, x in sorted(v.items(), key=_by_name)]))
    if isinstance(v, extent.TileExtent):
      return ('ex', v.ul, v.lr, v.array_shape)
    if isinstance(v, slice):
      return ('sl', v.start, v.stop, v.step)
    if isinstance(v, np.generic):
      return ('G', v.dtype.str, _scalar_key(v.item()))
    if isinstance(v, (np.dtype, type)):
      return ('T', str(v))
    if callable(v):
      self.pins.append(v)
      return ('f', id(v))
    raise Unplannable('field of type %s' % type(v).__name__)


def _by_name(kv):
  return str(kv[0])


_SIMPLE_SET = frozenset(_SIMPLE)


def signature(dag, flags):
  """(key, leaves, ids, pins) of the DAG as built, or None when it cannot be planned."""
  if _MapExpr[0] is None:
    from .map import MapExpr
    from .optimize import _keep_apart as keep
    _MapExpr[0] = MapExpr
    _keep_apart_ref.append(keep)         # (the set object itself: optimize.py only ever mutates it)
  _keep_apart = _keep_apart_ref[0]
  w = _Walk()
  try:
    body = w.node(dag)
  except Unplannable:
    return None
  if _keep_apart and any(i in _keep_apart for i in w.ids):
    return None
  ctx = context.get() if context.initialized() else None
  env = (flags, None if ctx is None else (ctx.num_workers, ctx.world.size, type(ctx.backend)))
  return (env, body), w.leaves, w.ids, w.pins


class _Slot(object):
  __slots__ = ('k',)

  def __init__(self, k):
    self.k = k


class _Node(object):
  """One inner node of a recorded DAG: type, fields (recipes), the position in the walk of the as-built DAG whose
  id it carries (None: a node the rewrites made), shape cache, needs_cache."""
  __slots__ = ('type', 'fields', 'at', 'shape_cache', 'needs_cache', 'index')


class Plan(object):
  """The optimised DAG recorded for one key, as a recipe that holds no array: leaves are slots (filled from the
  leaves of the DAG being answered), inner nodes are rebuilt, everything else (operator trees, functions, scalars,
  extents) is shared."""

  def __init__(self, optimized, leaves, ids, pins):
    slot = {id(v): k for k, v in enumerate(leaves)}
    position = {i: k for k, i in enumerate(ids)}
    nodes = {}

    def compile_(v):
      k = slot.get(id(v))
      if k is not None:
        return _Slot(k)
      if isinstance(v, Expr):
        if isinstance(v, (Val, AsArray)):
          # a leaf the rewrites made themselves: nothing of the recorded DAG's data may stay behind it
          if isinstance(v.val, (np.ndarray, distarray.DistArray)):
            raise Unplannable('the optimised DAG holds data of its own')
          return v
        n = nodes.get(id(v))
        if n is None:
          n = nodes[id(v)] = _Node()
          n.index = len(nodes) - 1
          n.type, n.at = type(v), position.get(v.expr_id)
          n.shape_cache, n.needs_cache = v.shape_cache, v.needs_cache
          d = v.__dict__
          n.fields = tuple((name, compile_(d[name])) for name in v.members)
        return n
      if isinstance(v, (np.ndarray, distarray.DistArray)) or hasattr(v, 'data_ptr'):
        raise Unplannable('the optimised DAG holds data of its own')
      if isinstance(v, (list, tuple)):
        return type(v)(compile_(x) for x in v) if type(v) in (list, tuple) else v
      if isinstance(v, dict):
        return {k: compile_(x) for k, x in v.items()}
      return v
    self.recipe = compile_(optimized)
    self.n_nodes = len(nodes)
    self.pins = pins                    # objects the key names by identity stay alive (their ids stay theirs)

  def instantiate(self, leaves, ids):
    return self._build(self.recipe, leaves, ids, [None] * self.n_nodes)

  def _build(self, r, leaves, ids, made):
    # (a method, not a closure that calls itself: such a function is a reference cycle with its own cell, and the
    # `made` / `leaves` lists it closes over -- hence the new DAG, and through its id the multi-GiB value computed
    # for it -- would live until the cyclic collector next runs)
    t = type(r)
    if t is _Slot:
      return leaves[r.k]
    build = self._build
    if t is _Node:
      done = made[r.index]
      if done is None:
        fields = {name: build(x, leaves, ids, made) for name, x in r.fields}
        done = r.type(expr_id=None if r.at is None else ids[r.at], shape_cache=r.shape_cache, **fields)
        if done.needs_cache != r.needs_cache:
          done.needs_cache = r.needs_cache
        made[r.index] = done
      return done
    if t is list:
      return [build(x, leaves, ids, made) for x in r]
    if t is tuple:
      return tuple(build(x, leaves, ids, made) for x in r)
    if t is dict:
      return {k: build(x, leaves, ids, made) for k, x in r.items()}
    return r


def optimized(dag, flags, optimize_fn):
  """optimize_fn(dag) through the plan table."""
  sig = signature(dag, flags)
  if sig is None:
    stats['unplannable'] += 1
    return optimize_fn(dag)
  key, leaves, ids, pins = sig
  plan = _plans.get(key)
  if plan is not None:
    _plans.move_to_end(key)
    stats['hits'] += 1
    return plan.instantiate(leaves, ids)
  stats['misses'] += 1
  out = optimize_fn(dag)
  try:
    _plans[key] = Plan(out, leaves, ids, pins)
  except Unplannable:
    stats['unplannable'] += 1
    return out
  while len(_plans) > MAX_PLANS:
    _plans.popitem(last=False)
  return out
