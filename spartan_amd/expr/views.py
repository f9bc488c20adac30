"""Zero-copy views: Slice, Transpose, Reshape (SURVEY 8f.1 -- the loaders/views
either side of the tile path).  Mirrors of the reference's
spartan/expr/operator/slice.py, transpose.py and reshape.py: a view re-maps
extents and delegates `fetch` to its base array; kernels then read the fetched
(possibly strided) slab.
"""
import itertools

import numpy as np

from . import base as base_mod
from .base import Expr, lazify
from .broadcast import Broadcast
from .. import context
from ..array import distarray, extent, tile
from ..context import LocalKernelResult
from ..util import Assert


def _permute_all(t):
  """ndarray.transpose() (reverse all axes) for a backend tensor / placeholder."""
  if isinstance(t, distarray.Absent):
    return distarray.Absent(t.shape[::-1], t.dtype)
  if tile.is_sparse_blob(t):
    return context.get().backend.sparse_transpose(t)
  if hasattr(t, 'permute'):
    return t.transpose()
  if isinstance(t, np.ndarray):
    return t.transpose()
  return t


# ------------------------------------------------------------------ views in general
class _View(distarray.DistArray):
  """A zero-copy re-indexing of `base`: regions are translated, the data is read where it lies.  A subclass says
  how a tile of the base looks from the view (`from_base`, None = not visible) and how a region of the view reads
  from the base (`fetch`).  Tile walks are chained through the base's own walk, so views of views compose."""

  def _adopt(self, base, shape):
    if not isinstance(base, distarray.DistArray):
      raise AssertionError('a view needs a distributed array, got %r' % type(base))
    self.base = base
    self.shape = tuple(shape)
    self.dtype, self.sparse, self.tiles = base.dtype, base.sparse, base.tiles
    self.bad_tiles = []

  def foreach_tile(self, mapper_fn, kw=None):
    return self.base.foreach_tile(mapper_fn=_view_tile, kw={'_view': self, '_fn': mapper_fn, '_fn_kw': kw})


def _view_tile(ex, _view, _fn, _fn_kw):
  seen = _view.from_base(ex)
  if seen is None:
    return LocalKernelResult(result=[])
  return _fn(seen, **(_fn_kw or {}))


# ----------------------------------------------------------------------- Slice
class Slice(_View):
  """base[idx] (reference spartan/expr/operator/slice.py:42-85)."""

  def __init__(self, darray, idx):
    self.slice = idx if isinstance(idx, extent.TileExtent) else extent.from_slice(idx, darray.shape)
    self._adopt(darray, self.slice.shape)
    self._tile_shape = distarray.good_tile_shape(self.shape, context.get().num_workers)

  def tile_shape(self):
    return self._tile_shape

  def from_base(self, base_ex):
    """The part of a base tile inside the slice, in the slice's own coordinates."""
    inside = extent.intersection(self.slice, base_ex)
    if inside is None:
      return None
    local = extent.offset_from(self.slice, inside)
    local.array_shape = self.slice.shape
    return local

  def extent_for_blob(self, id):
    return extent.intersection(self.slice, self.base.extent_for_blob(id))     # (base coordinates, slice.py:79-81)

  def fetch(self, idx):
    return self.base.fetch(extent.compute_slice(self.slice, idx.to_slice()))


class SliceExpr(Expr):
  members = ('src', 'idx', 'broadcast_to')

  def dependencies(self):
    return {'src': self.src}

  def visit(self, visitor):
    return base_mod.expr_like(self, src=visitor.visit(self.src), idx=self.idx, broadcast_to=self.broadcast_to)

  def compute_shape(self):
    if not isinstance(self.idx, (int, slice, tuple)):
      raise base_mod.NotShapeable
    return extent.compute_slice(extent.from_shape(self.src.shape), self.idx).shape

  def pretty_str(self):
    return 'Slice[%d](%s, %s)' % (self.expr_id, self.src, self.idx)

  def _evaluate(self, ctx, deps):
    src = deps['src']
    # (the slice-rotation pass slices operands in the shape of the map they fed: optimize.RotateSlice)
    if self.broadcast_to is not None and src.shape != self.broadcast_to:
      src = Broadcast(src, self.broadcast_to)
    return Slice(src, self.idx)


# ------------------------------------------------------------------- Transpose
def _mirror(ex, shape):
  return extent.create(ex.ul[::-1], ex.lr[::-1], shape)


class Transpose(_View):
  """base with its axes reversed (reference spartan/expr/operator/transpose.py:27-67)."""

  def __init__(self, base):
    self._adopt(base, base.shape[::-1])

  def tile_shape(self):
    return self.base.tile_shape()[::-1]

  def from_base(self, base_ex):
    return _mirror(base_ex, self.shape)

  view_extent = from_base

  def extent_for_blob(self, id):
    return _mirror(self.base.extent_for_blob(id), self.shape)

  def fetch(self, ex):
    return _permute_all(self.base.fetch(_mirror(ex, self.base.shape)))


class TransposeExpr(Expr):
  members = ('array', 'tile_hint')

  def dependencies(self):
    return {'array': self.array}

  def visit(self, visitor):
    return base_mod.expr_like(self, array=visitor.visit(self.array), tile_hint=self.tile_hint)

  def pretty_str(self):
    return 'Transpose[%d] %s' % (self.expr_id, self.array)

  def compute_shape(self):
    return self.array.shape[::-1]

  def _evaluate(self, ctx, deps):
    return Transpose(deps['array'])


def transpose(array, tile_hint=None):
  return TransposeExpr(array=lazify(array), tile_hint=tile_hint)


# --------------------------------------------------------------------- Reshape
def _ravelled_ex(ul, lr, shape):
  """reshape.py:20-23."""
  return extent.ravelled_pos(ul, shape), extent.ravelled_pos([l - 1 for l in lr], shape)


def _unravelled_ex(ravelled_ul, ravelled_lr, shape):
  """reshape.py:26-29."""
  return extent.unravelled_pos(ravelled_ul, shape), extent.unravelled_pos(ravelled_lr, shape)


def _reshape_invoke(self, tile_id, blob, mapper_fn, kw):
  """reshape.py:32-44."""
  if self.shape_array is None:
    ex = self.base.extent_for_blob(tile_id)
    r_ul, r_lr = _ravelled_ex(ex.ul, ex.lr, self.base.shape)
    u_ul, u_lr = _unravelled_ex(r_ul, r_lr, self.shape)
    ex = extent.create(u_ul, [v + 1 for v in u_lr], self.shape)
  else:
    ex = self.shape_array.extent_for_blob(tile_id)
  return mapper_fn(ex, **kw)


class Reshape(distarray.DistArray):
  """reshape.py:47-193 (dense)."""

  def __init__(self, base, shape, tile_hint=None):
    Assert.isinstance(base, distarray.DistArray)
    self.base = base
    self.shape = tuple(int(s) for s in shape)
    self.dtype = base.dtype
    self.sparse = self.base.sparse
    self.tiles = self.base.tiles
    self.bad_tiles = []
    self._tile_shape = distarray.good_tile_shape(self.shape, context.get().num_workers)
    self.shape_array = None
    # One extra axis of length 1 is the cheap case (the kmeans 'broadcast' program, argmin's reshape): a region of
    # the view is the same region of the base with that axis dropped -- no index arithmetic, no copy.
    self.new_dimension_idx = self._inserted_unit_axis()
    self.is_add_dimension = self.new_dimension_idx is not None
    self._same_tiles = self._tiles_line_up()

  def _inserted_unit_axis(self):
    """k if shape == base.shape with a 1 inserted at position k, else None."""
    base_shape = tuple(self.base.shape)
    if len(self.shape) != len(base_shape) + 1:
      return None
    for k, n in enumerate(self.shape):
      if n == 1 and self.shape[:k] + self.shape[k + 1:] == base_shape:
        return k
    return None

  def _tiles_line_up(self):
    """Can the tiles of the base, re-read in the new shape, serve as the tiles of the view?  Yes when axes were
    only appended (leading axes unchanged); otherwise every default tile of the new shape is tested against the
    base with the reference's predicate (reshape.py:107-118; its `or ul` makes it false for any array with
    dimensions, so such reshapes get an array of their own shape to walk over -- kept, the goldens pin it)."""
    if len(self.shape) > len(self.base.shape) and tuple(self.shape[:len(self.base.shape)]) == tuple(self.base.shape):
      return True
    for box in itertools.product(*distarray.compute_splits(self.shape, self._tile_shape)):
      ul, lr = zip(*box)
      flat_ul, flat_lr = _ravelled_ex(ul, lr, self.shape)
      rect_ul, rect_lr = extent.find_rect(flat_ul, flat_lr, self.base.shape)
      if rect_ul or ul or rect_lr != lr:
        return False
    return True

  def tile_shape(self):
    return self._tile_shape

  def view_extent(self, ex):
    r_ul, r_lr = _ravelled_ex(ex.ul, ex.lr, ex.array_shape)
    u_ul, u_lr = _unravelled_ex(r_ul, r_lr, self.shape)
    return extent.create(u_ul, [v + 1 for v in u_lr], self.shape)

  _invoke_mapper = _reshape_invoke

  def foreach_tile(self, mapper_fn, kw=None):
    if kw is None:
      kw = {}
    if self._same_tiles:
      tiles = list(self.base.tiles.values())
    else:
      if self.shape_array is None:
        self.shape_array = distarray.create(self.shape, self.base.dtype, tile_hint=self._tile_shape)
      tiles = list(self.shape_array.tiles.values())
    return distarray.run_kernel(self, tiles, mapper_fn, kw)

  def extent_for_blob(self, id):
    base_ex = self.base.extent_for_blob(id)
    r_ul, r_lr = _ravelled_ex(base_ex.ul, base_ex.lr, self.base.shape)
    u_ul, u_lr = _unravelled_ex(r_ul, r_lr, self.shape)
    return extent.create(u_ul, [v + 1 for v in u_lr], self.shape)

  def fetch(self, ex):
    """reshape.py:155-180."""
    if self.is_add_dimension:
      k = self.new_dimension_idx
      ul = ex.ul[0:k] + ex.ul[k + 1:]
      lr = ex.lr[0:k] + ex.lr[k + 1:]
      base_ex = extent.create(ul, lr, self.base.shape)
      return self.base.fetch(base_ex).reshape(ex.shape)
    ravelled_ul, ravelled_lr = _ravelled_ex(ex.ul, ex.lr, self.shape)
    b_ul, b_lr = extent.find_rect(ravelled_ul, ravelled_lr, self.base.shape)
    base_ul, base_lr = _unravelled_ex(b_ul, b_lr, self.base.shape)
    base_ex = extent.create(base_ul, [v + 1 for v in base_lr], self.base.shape)
    t = self.base.fetch(base_ex)
    if isinstance(t, distarray.Absent):
      return distarray.Absent(ex.shape, self.dtype)
    if tile.is_sparse_blob(t):
      Assert.eq(len(ex.shape), 2, 'sparse arrays are two-dimensional')
      return context.get().backend.sparse_reshape(t, ravelled_ul - b_ul, ex.shape)
    flat = context.get().backend.contiguous(t).reshape(-1)
    flat = flat[(ravelled_ul - b_ul):(ravelled_lr - b_ul) + 1]
    assert int(np.prod(flat.shape)) == int(np.prod(ex.shape)), (flat.shape, ex.shape)
    return flat.reshape(ex.shape)


class ReshapeExpr(Expr):
  """reshape.py:196-210."""
  members = ('array', 'new_shape', 'tile_hint')

  def dependencies(self):
    return {'array': self.array}

  def visit(self, visitor):
    return base_mod.expr_like(self, array=visitor.visit(self.array), new_shape=self.new_shape,
                              tile_hint=self.tile_hint)

  def pretty_str(self):
    return 'Reshape[%d] %s to %s' % (self.expr_id, self.array, self.new_shape)

  def _evaluate(self, ctx, deps):
    return Reshape(deps['array'], self.new_shape, self.tile_hint)

  def compute_shape(self):
    return tuple(self.new_shape)


def reshape(array, *args, **kargs):
  """reshape.py:213-239."""
  if len(args) == 1 and isinstance(args[0], (tuple, list)):
    new_shape = tuple(args[0])
  else:
    new_shape = tuple(args)
  tile_hint = kargs.get('tile_hint')
  Assert.isinstance(new_shape, tuple)
  return ReshapeExpr(array=lazify(array), new_shape=new_shape, tile_hint=tile_hint)


def ravel(v):
  """manipulation.py: flatten to 1-D."""
  return reshape(v, (int(np.prod(v.shape, dtype=np.int64)),))


# ------------------------------------------------------ Expr.__getitem__ & co.
def _getitem(self, idx):
  """base.py:401-447."""
  if isinstance(idx, (int, np.integer, tuple, slice)):
    is_del_dim = False
    del_dim = []
    if isinstance(idx, tuple):
      for x in range(len(idx)):
        if isinstance(idx[x], (int, np.integer)):
          is_del_dim = True
          del_dim.append(x)
    has_newaxis = isinstance(idx, tuple) and any(x is base_mod.newaxis for x in idx)
    if isinstance(idx, (int, np.integer)) or is_del_dim or has_newaxis:
      if isinstance(idx, tuple):
        new_idx = tuple([slice(x, None, None) if (isinstance(x, (int, np.integer)) and x == -1) else x
                         for x in idx if x is not base_mod.newaxis])
      else:
        new_idx = idx
      ret = SliceExpr(src=self, idx=new_idx, broadcast_to=None)
      new_shape = []
      if isinstance(idx, tuple):
        shape_ptr = idx_ptr = 0
        while shape_ptr < len(ret.shape) or idx_ptr < len(idx):
          if idx_ptr < len(idx) and idx[idx_ptr] is base_mod.newaxis:
            new_shape.append(1)
          else:
            new_shape.append(ret.shape[shape_ptr])
            shape_ptr += 1
          idx_ptr += 1
      else:
        new_shape = list(ret.shape)
        del_dim.append(0)
      # NB (base.py:437-440): only integer entries of a TUPLE index drop their axis; a bare
      # integer index keeps a length-1 axis (a[3] has shape (1, n)), as in the reference
      if is_del_dim:
        for i in sorted(del_dim, reverse=True):
          new_shape.pop(i)
      return ReshapeExpr(array=ret, new_shape=tuple(new_shape), tile_hint=None)
    return SliceExpr(src=self, idx=idx, broadcast_to=None)
  from .filter import filter_expr      # base.py:445-447: anything else is an index ARRAY
  return filter_expr(self, idx)


Expr.__getitem__ = _getitem
Expr.reshape = reshape
Expr.transpose = transpose
Expr.T = property(transpose)
Expr.ravel = ravel
Expr.flatten = ravel
