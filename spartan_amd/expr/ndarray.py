"""NdArrayExpr: lazily allocate an empty DistArray (reference
spartan/expr/operator/ndarray.py)."""
import numpy as np

from .base import Expr, expr_like
from ..array import distarray


class NdArrayExpr(Expr):
  members = ('_shape', 'sparse', 'dtype', 'tile_hint', 'reduce_fn')

  def pretty_str(self):
    return 'DistArray[%d](%s, %s, hint=%s)' % (self.expr_id, self.shape, np.dtype(self.dtype).name,
                                               self.tile_hint)

  def visit(self, visitor):
    return expr_like(self, _shape=self._shape, dtype=self.dtype, tile_hint=self.tile_hint,
                     sparse=self.sparse, reduce_fn=self.reduce_fn)

  def dependencies(self):
    return {}

  def compute_shape(self):
    return self._shape

  def _evaluate(self, ctx, deps):
    return distarray.create(self._shape, self.dtype, reducer=self.reduce_fn,
                            tile_hint=self.tile_hint, sparse=bool(self.sparse))


def ndarray(shape, dtype=float, tile_hint=None, reduce_fn=None, sparse=False):
  """ndarray.py:43-58."""
  return NdArrayExpr(_shape=tuple(shape), dtype=dtype, tile_hint=tile_hint, reduce_fn=reduce_fn,
                     sparse=sparse)
