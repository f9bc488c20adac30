"""In-place region writes: `write(array, src_slices, data, dst_slices)` -- array[src_slices] = data[dst_slices]
(the loader of SURVEY 8f.1; API of the reference's spartan/expr/operator/write_array.py:84-94).

The node is evaluated for its side effect and hands back the array it wrote into.  Host data is uploaded once and
pasted through `DistArray.update`; a distributed source is walked tile by tile of the TARGET: each tile that meets
the written box pulls exactly its own share of the source (a box copy on the owning GPU, or a grouped transfer when
the share lies on another rank) and pastes it into itself.
"""
import numpy as np

from . import base
from .base import Expr
from .views import Slice
from ..array import distarray, extent
from ..context import LocalKernelResult


def _pull_share(ex, target, box, source):
  """Tile `ex` of the target: if it meets `box`, the matching part of `source` (whose origin is the box's upper-left
  corner) is fetched and written over the meeting."""
  meet = extent.intersection(ex, box)
  if meet is not None:
    share = extent.offset_from(box, meet)              # the same cells, in the source's coordinates
    where = extent.create(share.ul, share.lr, source.shape)
    target.update(meet, source.fetch(where), wait=False)
  return LocalKernelResult(result=None)


class WriteArrayExpr(Expr):
  members = ('array', 'src_slices', 'data', 'dst_slices')

  def dependencies(self):
    return {'array': self.array, 'data': self.data}

  def visit(self, visitor):
    seen = lambda v: visitor.visit(v) if isinstance(v, Expr) else v     # noqa: E731
    return base.expr_like(self, array=seen(self.array), src_slices=self.src_slices, data=seen(self.data),
                          dst_slices=self.dst_slices)

  def pretty_str(self):
    return 'WriteArrayExpr[%d] %s %s' % (self.expr_id, self.array, self.data)

  def compute_shape(self):
    return self.array.shape

  def _evaluate(self, ctx, deps):
    target, data = deps['array'], deps['data']
    box = extent.from_slice(self.src_slices, target.shape)
    if isinstance(data, np.ndarray):
      host = data if data.shape == box.shape else data[self.dst_slices]
      if host.shape != box.shape:
        raise AssertionError('write: %s values for a box of shape %s' % (host.shape, box.shape))
      be = ctx.backend
      target.update(box, be.astype(be.from_numpy(np.ascontiguousarray(host)), target.dtype))
    elif isinstance(data, distarray.DistArray):
      source = Slice(data, self.dst_slices)
      if tuple(source.shape) != tuple(box.shape):
        raise AssertionError('write: source box %s, target box %s' % (source.shape, box.shape))
      target.foreach_tile(mapper_fn=_pull_share, kw={'target': target, 'box': box, 'source': source})
    else:
      raise TypeError('write: data must be a NumPy array or a distributed array, not %s' % type(data))
    return target


def write(array, src_slices, data, dst_slices):
  return WriteArrayExpr(array=array, src_slices=src_slices, data=data, dst_slices=dst_slices)
