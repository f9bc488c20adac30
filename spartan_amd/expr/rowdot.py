"""`sum(x * (dot(x, w) - y), axis=0)` as one pass over the rows of x.

The expression API states the gradient of a least-squares fit (reference: spartan/examples/linear_regression.py:10-24,
driven by tests/benchmark_lreg.py -- BASELINE configs[4]) as a matrix.vector product followed by a fused map ->
column reduce: two launches per tile, x streamed from HBM twice.  `RowDotColSumFusion` (expr/optimize.py) recognises
that DAG -- after the reference's own map / reduce fusions -- and replaces it by this node when the backend has a
one-pass kernel for it (HIP: csrc/rowdot.hip, sp_rowdot_colsum_f32): per row tile the rows stay in registers between
the two uses, the (d,) partial of each tile joins the target with np.add exactly like a reduction's.  The rewrite
changes the order of the floating-point sums, not their operands (results agree to rounding); on a backend without
the kernel (the NumPy oracle) it is never applied.
"""
import numpy as np

from . import base
from .base import Expr
from .broadcast import broadcast
from .map import get_local_values
from .. import context
from ..array import distarray, extent
from ..context import LocalKernelResult


def _rowdot_mapper(ex, inputs, names, w, output):
  ctx = context.get()
  values = get_local_values(ex, inputs, names)
  d = ex.array_shape[1]
  dst = extent.create((0,), (d,), (d,))
  if ctx.executing:
    partial = ctx.backend.rowdot_colsum(values['x'], w, values.get('y'))
  else:
    partial = distarray.Absent((d,), output.dtype)
  output.update(dst, partial, owned=True)
  return LocalKernelResult(result=[])


class RowDotColSumExpr(Expr):
  """g[c] = sum_i x[i, c] * (x[i, :] . w - y[i]) for a row-tiled 2-D fp32 x, a driver-side vector w and an optional
  (N, 1) array y."""
  members = ('x', 'y', 'w', 'tile_hint')

  def dependencies(self):
    return {'x': self.x, 'y': self.y} if self.y is not None else {'x': self.x}

  def visit(self, visitor):
    return base.expr_like(self, x=visitor.visit(self.x), y=visitor.visit(self.y) if self.y is not None else None,
                          w=self.w, tile_hint=self.tile_hint)

  def compute_shape(self):
    return (self.x.shape[1],)

  def pretty_str(self):
    return 'RowDotColSum[%d](%s, w%s, %s)' % (self.expr_id, self.x, tuple(self.w.shape), self.y)

  def _evaluate(self, ctx, deps):
    x, y = deps['x'], deps.get('y')
    inputs = broadcast([x, y]) if y is not None else [x]
    names = ['x', 'y'][:len(inputs)]
    output = distarray.create((x.shape[1],), np.float32, reducer=np.add, tile_hint=self.tile_hint)
    inputs[0].foreach_tile(_rowdot_mapper, kw={'inputs': inputs, 'names': names, 'w': self.w, 'output': output})
    return output
