"""Automatic tiling: choose row / column / block tiling (or replication) for the arrays an expression creates, so
that the bytes moved between GPUs are minimal.

Mirror of the reference's AutomaticTiling pass (spartan/expr/operator/optimize.py:459-1054) re-targeted at xGMI:
the pass walks the DAG and builds a graph whose nodes are (expression, tiling) alternatives -- tiling 0 = split
along dimension 0 (rows), 1 = along dimension 1 (columns), 2 = sqrt(p) x sqrt(p) blocks, 3 = replicated -- groups
the alternatives of one expression (exactly one is chosen), and connects producers to consumers with the cost of
feeding the consumer's tiling from the producer's.  The reference uses unit costs times the element count
(its `cost_model` tables, optimize.py:487-505); here an edge costs BYTES OVER xGMI LINKS: units x nbytes x (p-1)/p
(one redistribution moves all but 1/p of every GPU's share), replication p-1 copies.  The solver is C++ behind the
C-ABI (sp_tiling_solve, csrc/tiling.hip -- the reference's tiling.cc is a CPython-2 extension); the chosen tiling is
written into the `tile_hint` of NdArrayExpr / ReduceExpr / Map2Expr / OuterProductExpr nodes exactly as
AutomaticTiling.tile_expr does (optimize.py:922-935).

On by default, like the reference's (`optimize.FLAGS['opt_auto_tiling']`, optimize.py:1094).  The solver is pinned
against the reference's own tiling.cc on the cost graphs its pass builds for the shared test programs
(tests/golden/tiling_golden.json <- make_golden.py --tiling; tests/test_tiling.py): the same optimum as its exhaustive
best_tiling on every graph, never worse than its default mincost heuristic.
"""
import ctypes as C
import math

import numpy as np

from .base import AsArray, CollectionExpr, DictExpr, Expr, Val
from .map import Map2Expr
from .ndarray import NdArrayExpr
from .outer import OuterProductExpr
from .reduce import ReduceExpr
from .. import context
from ..array import distarray

ROW, COL, BLOCK, REPLICATED = 0, 1, 2, 3
_INFEASIBLE = 1e30
NUM_ALTERNATIVES = 4

# optimize.py:487-505, in redistributions: (producer tiling, consumer tiling) -> how many times the producer's data
# has to be re-split to feed an element-wise consumer (0: aligned; a replicated producer feeds anything)
_MAP_UNITS = {(0, 0): 0, (0, 1): 1, (0, 2): 1, (0, 3): 1,
              (1, 0): 1, (1, 1): 0, (1, 2): 1, (1, 3): 1,
              (2, 0): 1, (2, 1): 1, (2, 2): 0, (2, 3): 2,
              (3, 0): 0, (3, 1): 0, (3, 2): 1, (3, 3): 0}
# join of a producer with tiling t on axis a (map2): aligned when the producer is split along a or replicated
_MAP2_UNITS = {(0, 0): 0, (0, 1): 1, (0, 2): 1, (0, -1): 1,
               (1, 0): 1, (1, 1): 0, (1, 2): 1, (1, -1): 1,
               (2, 0): 1, (2, 1): 1, (2, 2): 0, (2, -1): 1,
               (3, 0): 0, (3, 1): 0, (3, 2): 1, (3, -1): 1}


def _dtype_of(x):
  try:
    return np.dtype(x.dtype)
  except Exception:
    return np.dtype(np.float32)


def _nbytes(x):
  shape = x.shape
  n = 1
  for s in shape:
    n *= int(s)
  try:
    item = np.dtype(x.dtype).itemsize
  except Exception:
    item = 4
  return float(n) * item


class _Node(object):
  __slots__ = ('exprs', 'tiling')

  def __init__(self, exprs, tiling):
    self.exprs = list(exprs)
    self.tiling = tiling


def solve(n_nodes, edges, groups):
  """sp_tiling_solve through ctypes: edges [(u, v, cost)], groups [[node, ...]] -> ([choice per group], total)."""
  from .. import _hip
  lib = _hip.lib()
  eu = (C.c_int32 * max(1, len(edges)))(*[e[0] for e in edges])
  ev = (C.c_int32 * max(1, len(edges)))(*[e[1] for e in edges])
  ec = (C.c_double * max(1, len(edges)))(*[float(e[2]) for e in edges])
  ptr, flat = [0], []
  for g in groups:
    flat.extend(g)
    ptr.append(len(flat))
  gp = (C.c_int32 * len(ptr))(*ptr)
  gn = (C.c_int32 * max(1, len(flat)))(*flat)
  choice = (C.c_int32 * max(1, len(groups)))()
  total = C.c_double(0.0)
  _hip.check(lib.sp_tiling_solve(n_nodes, len(edges), eu, ev, ec, len(groups), gp, gn, choice, C.byref(total)))
  return [int(choice[i]) for i in range(len(groups))], float(total.value)


class AutomaticTiling(object):
  """optimize.py:459-1054.  `visit(dag)` annotates the DAG in place and returns it; `self.report` keeps what was
  decided (tests, and the curious)."""
  name = 'auto_tiling'

  def __init__(self, num_workers=None):
    self.p = int(num_workers if num_workers is not None else context.get().num_workers)
    self.nodes = []
    self.edges = {}
    self.groups = []
    self.group_of = {}
    self.expr_nodes = {}
    self.report = {}

  # -- graph construction ----------------------------------------------------------
  def _link_bytes(self, units, x):
    """Bytes that cross xGMI links when `x` is redistributed `units` times."""
    return units * _nbytes(x) * (self.p - 1) / max(self.p, 1)

  def _new_node(self, exprs, tiling):
    self.nodes.append(_Node(exprs, tiling))
    return len(self.nodes) - 1

  def _add_edge(self, u, v, cost):
    self.edges[(u, v)] = self.edges.get((u, v), 0.0) + float(cost)

  def _add_group(self, ids):
    g = len(self.groups)
    self.groups.append(list(ids))
    for i in ids:
      self.group_of[i] = g

  def _alternatives(self, expr, tilings):
    ids = [self._new_node([expr], t) for t in tilings]
    if len(ids) > 1:
      self._add_group(ids)
    return ids

  def _tied(self, expr, kids, tilings):
    """Alternatives of `expr` that follow those of a child one to one (child alternative i <=> mine i): the
    reference leaves the mismatched pairs without an edge, which its solver prices as infinite (tiling.cc:103)."""
    ids = self._alternatives(expr, tilings)
    for i, mine in enumerate(ids):
      for j, k in enumerate(kids):
        self._add_edge(k, mine, 0 if i == j else _INFEASIBLE)
    return ids

  def _attach(self, ids, expr):
    for i in ids:
      self.nodes[i].exprs.append(expr)
    return ids

  def _children(self, children, skip=None):
    out = []
    for c in children:
      if isinstance(c, (Expr, distarray.DistArray)) and c is not skip:
        out.extend(self._visit(c))
    return out

  def _visit(self, expr):
    key = id(expr)
    if key in self.expr_nodes:
      return self.expr_nodes[key]
    if isinstance(expr, distarray.DistArray) or (isinstance(expr, (Val, AsArray)) and isinstance(expr.val, distarray.DistArray)):
      ids = self._visit_array(expr)
    elif isinstance(expr, (Val, AsArray)):
      ids = []                                   # a scalar / host array: replicated for free
    elif isinstance(expr, CollectionExpr):
      vals = expr.vals.values() if isinstance(expr, DictExpr) else expr.vals
      kids = self._children(vals)
      n = self._new_node([expr], -1)
      for k in kids:
        self._add_edge(k, n, 0)
      ids = [n]
    else:
      fn = getattr(self, '_visit_%s' % expr.typename(), None)
      ids = fn(expr) if fn is not None else self._visit_passthrough(expr)
    self.expr_nodes[key] = ids
    return ids

  def _visit_array(self, expr):
    """An already partitioned array: its tiling is what it is (optimize.py:1013-1030)."""
    array = expr if isinstance(expr, distarray.DistArray) else expr.val
    if isinstance(array, distarray.LocalWrapper) or not hasattr(array, 'tile_shape'):
      return []
    tile_shape = array.tile_shape()
    tiling = BLOCK
    for i in range(len(tile_shape)):
      if tile_shape[i] == array.shape[i]:
        tiling = 1 - i if i < 2 else ROW
        break
    return [self._new_node([expr], tiling)]

  def _visit_passthrough(self, expr):
    """Views and wrappers keep their source's alternatives (Slice / Filter / Checkpoint / TileOp,
    optimize.py:895-910); unknown nodes contribute nothing."""
    for attr in ('src', 'array'):
      src = getattr(expr, attr, None)
      if isinstance(src, (Expr, distarray.DistArray)):
        return self._attach(self._visit(src), expr)
    return []

  def _visit_NdArrayExpr(self, expr):
    """A new array can be created in any tiling; replicating it costs p - 1 copies (optimize.py:541-577)."""
    if len(expr.shape) > 1 and expr.shape[1] > 1:
      ids = self._alternatives(expr, range(NUM_ALTERNATIVES))
      self._add_edge(ids[REPLICATED], ids[REPLICATED], _nbytes(expr) * (self.p - 1))
      return ids
    return [self._new_node([expr], ROW)]

  def _visit_MapExpr(self, expr):
    """optimize.py:579-618: the map runs in the tiling of its LARGEST input; every other input is re-split to it."""
    vals = list(expr.children.vals)
    largest = max(vals, key=_nbytes)
    big = self._children([largest])
    others = self._children(vals, skip=largest)
    if not others or not big:
      return self._attach(big, expr) if big else self._attach(others, expr)
    tilings = [self.nodes[i].tiling for i in big]
    ids = self._tied(expr, big, tilings)
    for t, mine in zip(tilings, ids):
      for o in others:
        units = _MAP_UNITS.get((self.nodes[o].tiling, t), 1)
        self._add_edge(o, mine, self._link_bytes(units, self.nodes[o].exprs[0]))
    return ids

  def _visit_ReduceExpr(self, expr):
    """optimize.py:620-629: reducing along the split axis leaves partials to combine (the output's bytes);
    along the other axis, or of a replicated input, nothing moves."""
    kids = self._children(expr.children.vals)
    n = self._new_node([expr], ROW)
    for k in kids:
      t = self.nodes[k].tiling
      free = expr.axis is None or t == REPLICATED or (1 - expr.axis) == t
      self._add_edge(k, n, 0 if free else self._link_bytes(1, expr))
    return [n]

  def _join(self, expr, aligned):
    """Common part of map2 / outer (optimize.py:631-698): one copy node per joined array, then the output in any
    tiling; producing the output replicated costs twice its bytes."""
    groups = [self._children([a]) for a in expr.arrays.vals]
    copies = []
    for axis, kids in zip(expr.axes, groups):
      if isinstance(axis, tuple):
        axis = BLOCK
      if axis is None:
        axis = -1
      n = self._new_node([expr], axis)
      for k in kids:
        self._add_edge(k, n, aligned(self.nodes[k].tiling, axis, self.nodes[k].exprs[0]))
      if copies:
        self._add_edge(n, copies[0], 0)
      copies.append(n)
    out_bytes = self._link_bytes(1, expr)
    if len(expr.shape) > 1 and expr.shape[1] > 1:
      ids = self._alternatives(expr, range(NUM_ALTERNATIVES))
      for t, i in zip(range(NUM_ALTERNATIVES), ids):
        self._add_edge(copies[0], i, (t // 3 + 1) * out_bytes)
      return ids
    n = self._new_node([expr], ROW)
    self._add_edge(copies[0], n, out_bytes)
    return [n]

  def _visit_Map2Expr(self, expr):
    return self._join(expr, lambda t, axis, x: self._link_bytes(_MAP2_UNITS.get((t, axis), 1), x))

  def _visit_OuterProductExpr(self, expr):
    def aligned(t, axis, x):
      free = axis is not None and axis != -1 and (axis == t or t == REPLICATED)
      return 0 if free else self._link_bytes(1, x)
    return self._join(expr, aligned)

  def _visit_ShuffleExpr(self, expr):
    """optimize.py:700-770: the shuffle runs in its source's tiling; every array in its keyword arguments, and its
    target, is reached from there.  What that costs is the user's `cost_hint` when there is one -- ELEMENTS of the
    array that move for a (tiling of the array, tiling of the shuffle) pair, keys '00' '01' '10' '11' with 0 = rows
    and 1 = columns, as the reference defines it (shuffle.py:99-135, optimize.py:701-707) -- converted to bytes
    over links; without a hint the array is fetched whole once."""
    ids = self._visit(expr.array)
    extra = []
    kw = expr.fn_kw
    if isinstance(kw, DictExpr):
      extra = self._children(kw.vals.values())
    if expr.target is not None:
      extra += self._children([expr.target])
    if not extra:
      return self._attach(ids, expr)
    out = self._tied(expr, ids, [self.nodes[i].tiling for i in ids]) if ids else [self._new_node([expr], ROW)]
    hints = expr.cost_hint or {}
    for mine in out:
      for o in extra:
        array = self.nodes[o].exprs[0]
        hint = hints.get(hash(array))
        key = '%d%d' % (self.nodes[o].tiling, self.nodes[mine].tiling)
        if hint is not None and self.nodes[o].tiling in (ROW, COL) and self.nodes[mine].tiling in (ROW, COL) \
            and key in hint:
          moved = float(hint[key]) * np.dtype(_dtype_of(array)).itemsize * (self.p - 1) / max(self.p, 1)
        else:
          moved = self._link_bytes(1, array)
        self._add_edge(o, mine, moved)
    return out

  def _aligned_view(self, expr, swap):
    """Transpose / Reshape swap rows and columns of their source's tiling (optimize.py:856-893)."""
    kids = self._visit(expr.array)
    if not kids:
      return []
    tilings = []
    for k in kids:
      t = self.nodes[k].tiling
      tilings.append((1 - t) if (swap and t in (ROW, COL)) else t)
    return self._tied(expr, kids, tilings)

  def _visit_TransposeExpr(self, expr):
    return self._aligned_view(expr, True)

  def _visit_ReshapeExpr(self, expr):
    return self._aligned_view(expr, True)

  # -- solution ----------------------------------------------------------------------
  def tile_expr(self, expr, tiling):
    """optimize.py:922-935."""
    if isinstance(expr, (NdArrayExpr, ReduceExpr, Map2Expr, OuterProductExpr)) and len(expr.shape) > 0:
      hint = [int(s) for s in expr.shape]
      if tiling >= REPLICATED:
        return                                   # one copy everywhere: no tiling to ask for
      if tiling == BLOCK and len(hint) > 1:
        side = math.sqrt(self.p)
        hint[0] = int(math.ceil(hint[0] / side))
        hint[1] = int(math.ceil(hint[1] / side))
      elif len(hint) > tiling:
        hint[tiling] = int(math.ceil(hint[tiling] / float(self.p)))
      expr.tile_hint = tuple(max(1, h) for h in hint)

  def visit(self, dag):
    if not isinstance(dag, Expr) or self.p <= 1:
      return dag
    top = self._visit(dag)
    if not self.nodes:
      return dag
    sink = self._new_node([dag], -1)
    for t in top:
      self._add_edge(t, sink, 0)
    edges = [(u, v, c) for (u, v), c in self.edges.items()]
    try:
      choice, total = solve(len(self.nodes), edges, self.groups)
    except ImportError as e:       # _hip.HipLibraryMissing: the solver is native code of libspartan_hip.so
      # without the built library (a CPU-only checkout driving the test backend) the pass is skipped -- every
      # expression keeps the tiling it would have had with opt_auto_tiling off
      self.report = {'skipped': str(e)}
      return dag
    chosen = set(range(len(self.nodes)))
    for g, members in enumerate(self.groups):
      for s, n in enumerate(members):
        if s != choice[g]:
          chosen.discard(n)
    decided = {}
    for n in sorted(chosen):
      node = self.nodes[n]
      if node.tiling is None or node.tiling < 0:
        continue
      for e in node.exprs:
        if isinstance(e, Expr):
          self.tile_expr(e, node.tiling)
          decided[e.expr_id] = node.tiling
    self.report = {'link_bytes': total, 'tilings': decided, 'groups': len(self.groups), 'nodes': len(self.nodes)}
    return dag
