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
    return t.permute(*reversed(range(t.dim())))
  if isinstance(t, np.ndarray):
    return t.transpose()
  return t


# ----------------------------------------------------------------------- Slice
def _slice_mapper(ex, **kw):
  """slice.py:9-39: run the mapper on the part of the tile that lies inside the
  slice, expressed in the slice's own coordinates.  Chained through the base
  array's foreach_tile, so views of views compose like in the reference."""
  mapper_fn = kw['_slice_fn']
  slice_extent = kw['_slice_extent']
  fn_kw = kw['fn_kw']
  if fn_kw is None:
    fn_kw = {}
  intersection = extent.intersection(slice_extent, ex)
  if intersection is None:
    return LocalKernelResult(result=[])
  offset = extent.offset_from(slice_extent, intersection)
  offset.array_shape = slice_extent.shape
  return mapper_fn(offset, **fn_kw)


class Slice(distarray.DistArray):
  """slice.py:42-85."""

  def __init__(self, darray, idx):
    if not isinstance(idx, extent.TileExtent):
      idx = extent.from_slice(idx, darray.shape)
    Assert.isinstance(darray, distarray.DistArray)
    self.base = darray
    self.slice = idx
    self.shape = self.slice.shape
    self.tiles = self.base.tiles
    self.dtype = darray.dtype
    self.sparse = self.base.sparse
    self.bad_tiles = []
    self._tile_shape = distarray.good_tile_shape(self.shape, context.get().num_workers)

  def tile_shape(self):
    return self._tile_shape

  def foreach_tile(self, mapper_fn, kw=None):
    """slice.py:73-77."""
    return self.base.foreach_tile(mapper_fn=_slice_mapper,
                                  kw={'fn_kw': kw, '_slice_extent': self.slice, '_slice_fn': mapper_fn})

  def extent_for_blob(self, id):
    base_ex = self.base.extent_for_blob(id)
    return extent.intersection(self.slice, base_ex)

  def fetch(self, idx):
    offset = extent.compute_slice(self.slice, idx.to_slice())
    return self.base.fetch(offset)


class SliceExpr(Expr):
  """slice.py:88-137."""
  members = ('src', 'idx', 'broadcast_to')

  def dependencies(self):
    return {'src': self.src}

  def visit(self, visitor):
    return base_mod.expr_like(self, src=visitor.visit(self.src), idx=self.idx, broadcast_to=self.broadcast_to)

  def compute_shape(self):
    if isinstance(self.idx, (int, slice, tuple)):
      src_shape = self.src.shape
      ex = extent.from_shape(src_shape)
      slice_ex = extent.compute_slice(ex, self.idx)
      return slice_ex.shape
    raise base_mod.NotShapeable

  def pretty_str(self):
    return 'Slice[%d](%s, %s)' % (self.expr_id, self.src, self.idx)

  def _evaluate(self, ctx, deps):
    src = deps['src']
    idx = self.idx
    if self.broadcast_to is not None and src.shape != self.broadcast_to:
      src = Broadcast(src, self.broadcast_to)
    return Slice(src, idx)


# ------------------------------------------------------------------- Transpose
def _transpose_mapper(ex, **kw):
  """transpose.py:19-24."""
  user_fn = kw['_fn']
  fn_kw = kw['_fn_kw']
  view = kw['_base']
  if fn_kw is None:
    fn_kw = {}
  view_ex = extent.create(ex.ul[::-1], ex.lr[::-1], view.shape)
  return user_fn(view_ex, **fn_kw)


class Transpose(distarray.DistArray):
  """transpose.py:27-67."""

  def __init__(self, base):
    Assert.isinstance(base, distarray.DistArray)
    self.base = base
    self.shape = self.base.shape[::-1]
    self.dtype = base.dtype
    self.sparse = self.base.sparse
    self.tiles = base.tiles
    self.bad_tiles = []

  def tile_shape(self):
    return self.base.tile_shape()[::-1]

  def view_extent(self, ex):
    return extent.create(ex.ul[::-1], ex.lr[::-1], self.shape)

  def foreach_tile(self, mapper_fn, kw=None):
    """transpose.py:54-58."""
    return self.base.foreach_tile(mapper_fn=_transpose_mapper,
                                  kw={'_fn_kw': kw, '_base': self, '_fn': mapper_fn})

  def extent_for_blob(self, id):
    base_ex = self.base.extent_for_blob(id)
    return extent.create(base_ex.ul[::-1], base_ex.lr[::-1], self.shape)

  def fetch(self, ex):
    base_ex = extent.create(ex.ul[::-1], ex.lr[::-1], self.base.shape)
    return _permute_all(self.base.fetch(base_ex))


class TransposeExpr(Expr):
  """transpose.py:70-84."""
  members = ('array', 'tile_hint')

  def dependencies(self):
    return {'array': self.array}

  def visit(self, visitor):
    return base_mod.expr_like(self, array=visitor.visit(self.array), tile_hint=self.tile_hint)

  def pretty_str(self):
    return 'Transpose[%d] %s' % (self.expr_id, self.array)

  def _evaluate(self, ctx, deps):
    return Transpose(deps['array'])

  def compute_shape(self):
    return self.array.shape[::-1]


def transpose(array, tile_hint=None):
  """transpose.py:87-100."""
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
    only appended (leading axes unchanged), or when every default tile of the new shape is one contiguous
    rectangle of the base that starts where the tile does (reference reshape.py:91-118)."""
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

  def foreach_tile(self, mapper_fn, kw=None):
    """slice.py:73-77."""
    return self.base.foreach_tile(mapper_fn=_slice_mapper,
                                  kw={'fn_kw': kw, '_slice_extent': self.slice, '_slice_fn': mapper_fn})

  def extent_for_blob(self, id):
    base_ex = self.base.extent_for_blob(id)
    return extent.intersection(self.slice, base_ex)

  def fetch(self, idx):
    offset = extent.compute_slice(self.slice, idx.to_slice())
    return self.base.fetch(offset)


class SliceExpr(Expr):
  """slice.py:88-137."""
  members = ('src', 'idx', 'broadcast_to')

  def dependencies(self):
    return {'src': self.src}

  def visit(self, visitor):
    return base_mod.expr_like(self, src=visitor.visit(self.src), idx=self.idx, broadcast_to=self.broadcast_to)

  def compute_shape(self):
    if isinstance(self.idx, (int, slice, tuple)):
      src_shape = self.src.shape
      ex = extent.from_shape(src_shape)
      slice_ex = extent.compute_slice(ex, self.idx)
      return slice_ex.shape
    raise base_mod.NotShapeable

  def pretty_str(self):
    return 'Slice[%d](%s, %s)' % (self.expr_id, self.src, self.idx)

  def _evaluate(self, ctx, deps):
    src = deps['src']
    idx = self.idx
    if self.broadcast_to is not None and src.shape != self.broadcast_to:
      src = Broadcast(src, self.broadcast_to)
    return Slice(src, idx)


# ------------------------------------------------------------------- Transpose
def _transpose_mapper(ex, **kw):
  """transpose.py:19-24."""
  user_fn = kw['_fn']
  fn_kw = kw['_fn_kw']
  view = kw['_base']
  if fn_kw is None:
    fn_kw = {}
  view_ex = extent.create(ex.ul[::-1], ex.lr[::-1], view.shape)
  return user_fn(view_ex, **fn_kw)


class Transpose(distarray.DistArray):
  """transpose.py:27-67."""

  def __init__(self, base):
    Assert.isinstance(base, distarray.DistArray)
    self.base = base
    self.shape = self.base.shape[::-1]
    self.dtype = base.dtype
    self.sparse = self.base.sparse
    self.tiles = base.tiles
    self.bad_tiles = []

  def tile_shape(self):
    return self.base.tile_shape()[::-1]

  def view_extent(self, ex):
    return extent.create(ex.ul[::-1], ex.lr[::-1], self.shape)

  def foreach_tile(self, mapper_fn, kw=None):
    """transpose.py:54-58."""
    return self.base.foreach_tile(mapper_fn=_transpose_mapper,
                                  kw={'_fn_kw': kw, '_base': self, '_fn': mapper_fn})

  def extent_for_blob(self, id):
    base_ex = self.base.extent_for_blob(id)
    return extent.create(base_ex.ul[::-1], base_ex.lr[::-1], self.shape)

  def fetch(self, ex):
    base_ex = extent.create(ex.ul[::-1], ex.lr[::-1], self.base.shape)
    return _permute_all(self.base.fetch(base_ex))


class TransposeExpr(Expr):
  """transpose.py:70-84."""
  members = ('array', 'tile_hint')

  def dependencies(self):
    return {'array': self.array}

  def visit(self, visitor):
    return base_mod.expr_like(self, array=visitor.visit(self.array), tile_hint=self.tile_hint)

  def pretty_str(self):
    return 'Transpose[%d] %s' % (self.expr_id, self.array)

  def _evaluate(self, ctx, deps):
    return Transpose(deps['array'])

  def compute_shape(self):
    return self.array.shape[::-1]


def transpose(array, tile_hint=None):
  """transpose.py:87-100."""
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
    # adding one size-1 dimension is the cheap case (reshape.py:73-87)
    self.is_add_dimension = False
    if len(self.shape) == len(self.base.shape) + 1:
      self.is_add_dimension = True
      extra = 0
      for i in range(len(self.base.shape)):
        if self.shape[i + extra] != self.base.shape[i]:
          if extra == 0 and self.shape[i] == 1:
            self.new_dimension_idx = i
            extra = 1
          else:
            self.is_add_dimension = False
            break
      if extra == 0:
        self.new_dimension_idx = len(self.shape) - 1
    self._check_extents()

  def _check_extents(self):
    """reshape.py:91-118."""
    self._same_tiles = True
    if len(self.shape) > len(self.base.shape):
      for i in range(len(self.base.shape)):
        if self.base.shape[i] != self.shape[i]:
          self._same_tiles = False
          break
      if self._same_tiles:
        return
    splits = distarray.compute_splits(self.shape, self._tile_shape)
    for slc in itertools.product(*splits):
      ul, lr = zip(*slc)
      ravelled_ul, ravelled_lr = _ravelled_ex(ul, lr, self.shape)
      rect_ul, rect_lr = extent.find_rect(ravelled_ul, ravelled_lr, self.base.shape)
      if rect_ul or ul or rect_lr != lr:
        self._same_tiles = False
        break

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
