"""tile_operation: run a function over every tile and hand its results straight back to the driver
(reference spartan/expr/operator/tile_operation.py).  Every rank walks the tiles; the executing rank calls `fn` on
the fetched data, and the values are made available on every rank (the reference returns them to the master)."""
from . import base
from .base import Expr, NotShapeable, lazify
from .. import context
from ..context import LocalKernelResult
from ..util import is_iterable


class _Value(object):
  __slots__ = ('v',)

  def __init__(self, v):
    self.v = v


def tile_op_mapper(ex, map_fn=None, source=None, fn_kw=None):
  """tile_operation.py:31-53."""
  ctx = context.get()
  result = map_fn(source, ex, **fn_kw)
  values = [v for (k, v) in list(result)] if result is not None else []
  return LocalKernelResult(result=[(ex, _Value(values if ctx.executing else None))])


class TileOpExpr(Expr):
  """tile_operation.py:56-77."""
  members = ('array', 'map_fn', 'fn_kw')

  def dependencies(self):
    return {'array': self.array, 'fn_kw': self.fn_kw}

  def visit(self, visitor):
    return base.expr_like(self, array=visitor.visit(self.array), map_fn=self.map_fn, fn_kw=visitor.visit(self.fn_kw))

  def pretty_str(self):
    return 'tile_operation[%d](%s, %s)' % (self.expr_id, self.map_fn, self.array)

  def compute_shape(self):
    raise NotShapeable

  def _evaluate(self, ctx, deps):
    v = deps['array']
    res = v.foreach_tile(mapper_fn=tile_op_mapper, kw=dict(map_fn=self.map_fn, source=v, fn_kw=deps['fn_kw']))
    out = {}
    for tile_id, items in res.items():
      for ex, val in items:
        out[tile_id] = val.v
    if ctx.world.distributed:
      merged = {}
      for part in ctx.world.all_gather_object({(k.worker, k.id): v for k, v in out.items() if v is not None}):
        merged.update(part)
      out = {k: merged.get((k.worker, k.id)) for k in out}
    return out


def tile_operation(v, fn, kw=None):
  """tile_operation.py:9-29: {tile_id: [values yielded by fn(array, extent, **kw)]}."""
  if kw is None:
    kw = {}
  kw = lazify(kw)
  v = lazify(v)
  assert not is_iterable(v)
  return TileOpExpr(array=v, map_fn=fn, fn_kw=kw)
