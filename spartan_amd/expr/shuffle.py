"""shuffle: generic tile -> [(extent, data)] scatter (reference
spartan/expr/operator/shuffle.py).  The user function receives the source
DistArray and the tile extent and works on backend (HBM) tensors."""
from . import base
from .base import Expr, NotShapeable, lazify
from .. import context
from ..array import tile
from ..context import LocalKernelResult
from ..util import Assert, is_iterable


def shuffle(v, fn, cost_hint=None, shape_hint=None, target=None, kw=None):
  """shuffle.py:12-38."""
  if kw is None:
    kw = {}
  if cost_hint is None:
    cost_hint = {}
  kw = lazify(kw)
  v = lazify(v)
  if target is not None:
    target = lazify(target)
  assert not is_iterable(v)
  return ShuffleExpr(array=v, map_fn=fn, cost_hint=cost_hint, shape_hint=shape_hint, target=target,
                     fn_kw=kw)


def _produced(map_fn, source, ex, fn_kw):
  """What the user function yields for tile `ex` of `source`: (extent, data) pairs (None: nothing)."""
  return list(map_fn(source, ex, **(fn_kw or {})) or ())


def target_mapper(ex, map_fn=None, source=None, target=None, fn_kw=None):
  """Scatter with the target's reducer (protocol of the reference's target_mapper, shuffle.py:41-66)."""
  for where, data in _produced(map_fn, source, ex, fn_kw):
    target.update(where, data, wait=False)
  return LocalKernelResult(result=[])


def notarget_mapper(ex, array=None, map_fn=None, source=None, fn_kw=None):
  """Every piece the function yields becomes a tile of a new array (shuffle.py:69-96).  Every rank walks the same
  pieces so that tile ids advance alike; only the rank that runs this worker's kernels holds data."""
  ctx = context.get()
  table = []
  for where, data in _produced(map_fn, source, ex, fn_kw):
    piece = None
    if ctx.executing:
      Assert.eq(where.shape, tuple(data.shape), 'Bad shape from %s' % map_fn)
      piece = tile.from_data(data, dtype=ctx.backend.dtype_of(data))
    table.append((where, ctx.create(piece)))
  return LocalKernelResult(result=table, futures=None)


class ShuffleExpr(Expr):
  """shuffle.py:99-135."""
  members = ('array', 'map_fn', 'target', 'cost_hint', 'shape_hint', 'fn_kw')

  def dependencies(self):
    return {'array': self.array, 'target': self.target, 'fn_kw': self.fn_kw}

  def visit(self, visitor):
    return base.expr_like(self, array=visitor.visit(self.array), map_fn=self.map_fn,
                          target=visitor.visit(self.target) if self.target is not None else None,
                          cost_hint=self.cost_hint, shape_hint=self.shape_hint,
                          fn_kw=visitor.visit(self.fn_kw))

  def _evaluate(self, ctx, deps):
    v = deps['array']
    fn_kw = deps['fn_kw']
    target = deps['target']
    if target is not None:
      v.foreach_tile(mapper_fn=target_mapper,
                     kw=dict(map_fn=self.map_fn, source=v, target=target, fn_kw=fn_kw))
      return target
    return v.map_to_array(mapper_fn=notarget_mapper, kw=dict(source=v, map_fn=self.map_fn, fn_kw=fn_kw))

  def compute_shape(self):
    if self.target is not None:
      return self.target.shape
    if self.shape_hint is not None:
      return self.shape_hint
    raise NotShapeable
