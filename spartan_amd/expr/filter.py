"""Indexing an expression with an ARRAY (`x[idx]`): mirror of the reference's spartan/expr/operator/filter.py.

Integer indices (1-D): row i of the result is row idx[i] of the source (filter.py:50-75: one `src.select` RPC per
index there).  Here every tile of the result is produced by its owner with ONE row gather per source tile it touches
(sp_gather_rows) after the source rows it needs have been fetched as whole tiles' row ranges.

Boolean indices: the reference returns the source array again with NumPy MaskedArray tiles (filter.py:78-97);
masked arrays do not exist on the device, so this is refused loudly (use `where`-style maps or a reduction over a
comparison instead).
"""
import numpy as np

from . import base
from .base import Expr, NotShapeable, lazify
from .. import context
from ..array import distarray, extent, tile as tile_mod
from ..context import LocalKernelResult
from ..util import Assert


def _int_index_mapper(ex, src=None, idx=None, dst=None):
  """filter.py:50-75."""
  ctx = context.get()
  lo, hi = ex.ul[0], ex.lr[0]
  rows = np.asarray(idx[lo:hi], dtype=np.int64)
  rows = np.where(rows < 0, rows + src.shape[0], rows)
  out_shape = (hi - lo,) + tuple(src.shape[1:])
  out_ex = extent.create((lo,) + (0,) * (len(src.shape) - 1), (hi,) + tuple(src.shape[1:]), dst.shape)
  # the smallest row range of the source that holds every row this tile needs
  r0, r1 = (int(rows.min()), int(rows.max()) + 1) if rows.size else (0, 1)
  block = src.fetch(extent.create((r0,) + (0,) * (len(src.shape) - 1), (r1,) + tuple(src.shape[1:]), src.shape))
  if ctx.executing:
    data = ctx.backend.gather_rows(block, rows - r0).reshape(out_shape)
    t = tile_mod.from_data(data, dtype=ctx.backend.dtype_of(data))
  else:
    t = None
  return LocalKernelResult(result=[(out_ex, ctx.create(t))])


def eval_index(ctx, src, idx):
  """filter.py:100-128."""
  Assert.isinstance(idx, (np.ndarray, distarray.DistArray))
  if np.dtype(idx.dtype) == np.bool_:
    raise NotImplementedError('boolean-array indexing yields MaskedArray tiles in the reference (filter.py:78-97); '
                              'masked arrays are not supported on the GPU backend')
  Assert.eq(len(idx.shape), 1)
  host_idx = np.asarray(idx.glom() if isinstance(idx, distarray.DistArray) else idx).astype(np.int64)
  dst = distarray.create((int(host_idx.shape[0]),) + tuple(src.shape[1:]), dtype=src.dtype)
  return dst.map_to_array(_int_index_mapper, kw={'src': src, 'idx': host_idx, 'dst': dst})


class FilterExpr(Expr):
  """filter.py:16-47."""
  members = ('src', 'idx')

  def dependencies(self):
    return {'src': self.src, 'idx': self.idx}

  def visit(self, visitor):
    return base.expr_like(self, src=visitor.visit(self.src), idx=visitor.visit(self.idx))

  def pretty_str(self):
    return 'Filter[%d](%s, %s)' % (self.expr_id, self.src, self.idx)

  def compute_shape(self):
    idx = self.idx
    if isinstance(idx, Expr):
      idx_shape = idx.shape
    else:
      idx_shape = np.asarray(idx).shape
    if len(idx_shape) != 1:
      raise NotShapeable
    return (int(idx_shape[0]),) + tuple(self.src.shape[1:])

  def _evaluate(self, ctx, deps):
    idx = deps['idx']
    if isinstance(idx, distarray.LocalWrapper):
      idx = np.asarray(idx.glom())
    return eval_index(ctx, deps['src'], idx)


def filter_expr(src, idx):
  return FilterExpr(src=src, idx=lazify(idx) if not isinstance(idx, np.ndarray) else base.Val(val=idx))
