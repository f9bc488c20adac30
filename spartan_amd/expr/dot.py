"""spartan.dot: dispatch to map2 / outer plus the per-tile GEMM mappers.
Mirror of the reference's spartan/expr/dot.py:172-299; `tiles[0].dot(tiles[1])`
is the fp32 MFMA GEMM (sp_gemm_f32) for matrix.matrix and a fused
multiply-reduce launch for matrix.vector."""
import numpy as np

from . import map as map_mod
from . import outer as outer_mod
from .base import Expr
from .. import context
from ..array import distarray, extent


def _dot(a, b):
  """`a.dot(b)` on backend tensors (Absent on non-executing ranks)."""
  ctx = context.get()
  if isinstance(b, distarray.ChunkedWhole):
    if isinstance(a, distarray.Absent):
      return distarray.Absent((a.shape[0], b.shape[1]), a.dtype)
    return ctx.backend.dot_chunked(a, b)
  if isinstance(a, distarray.Absent) or isinstance(b, distarray.Absent):
    ash, bsh = tuple(a.shape), tuple(b.shape)
    if len(ash) == 1 and len(bsh) == 1:
      shp = ()
    elif len(bsh) == 1:
      shp = ash[:-1]
    elif len(ash) == 1:
      shp = bsh[1:]
    else:
      shp = (ash[0], bsh[1])
    return distarray.Absent(shp, a.dtype)
  return ctx.backend.dot(a, b)


def dot_map2_np_mapper(extents, tiles, array2):
  """dot.py:172-187: rhs is a driver-side NumPy array (replicated)."""
  ctx = context.get()
  ex = extents[0]
  if len(ex.ul) == 1:
    target_ex = extent.create((0,), (1,), (1,))
    rhs = ctx.backend.cached_numpy(array2, (slice(ex.ul[0], ex.lr[0]),))
    target_tile = _dot(tiles[0], rhs).reshape(1,)
  elif len(array2.shape) == 1:
    target_ex = extent.create((ex.ul[0],), (ex.lr[0],), (ex.array_shape[0],))
    rhs = ctx.backend.cached_numpy(array2, (slice(ex.ul[1], ex.lr[1]),))
    target_tile = _dot(tiles[0], rhs)
  else:
    target_ex = extent.create((ex.ul[0], 0), (ex.lr[0], array2.shape[1]),
                              (ex.array_shape[0], array2.shape[1]))
    rhs = ctx.backend.cached_numpy(array2, (slice(ex.ul[1], ex.lr[1]),))
    target_tile = _dot(tiles[0], rhs)
  yield target_ex, target_tile


def dot_map2_vec_mapper(extents, tiles):
  """dot.py:189-191."""
  target_ex = extent.create((0,), (1,), (1,))
  yield target_ex, _dot(tiles[0], tiles[1]).reshape(1,)


def dot_map2_mapper(extents, tiles, is_vec=None):
  """dot.py:195-217: the K-slab partial product of the map2 join."""
  if is_vec:
    ul = (0,)
    lr = (extents[1].lr[1],)
    shape = (extents[1].shape[1],)
    tiles[0] = tiles[0].reshape(extents[0].shape[1],)
  elif len(tiles[1].shape) == 1:
    ul = (0,)
    lr = (extents[0].lr[0],)
    shape = (extents[0].shape[0],)
  else:
    ul = (0, 0)
    lr = (extents[0].lr[0], extents[1].lr[1])
    shape = (extents[0].shape[0], extents[1].shape[1])
  target_ex = extent.create(ul, lr, shape)
  yield target_ex, _dot(tiles[0], tiles[1])


def dot_outer_mapper(ex_a, tile_a, ex_b, tile_b):
  """dot.py:222-238: row block of A times all of B -> disjoint row block of C."""
  if len(tile_b.shape) == 1:
    ul = (ex_a.ul[0],)
    lr = (ex_a.lr[0],)
    shape = (ex_a.array_shape[0],)
  else:
    ul = (ex_a.ul[0], ex_b.ul[1])
    lr = (ex_a.lr[0], ex_b.lr[1])
    shape = (ex_a.array_shape[0], ex_b.array_shape[1])
  target_ex = extent.create(ul, lr, shape)
  yield target_ex, _dot(tile_a, tile_b)


def _fetch_whole_rhs(array, whole_extent):
  """dot_outer_mapper's B: every worker needs ALL of it (outer.py:21-29).  Across GPUs the
  gather is issued as a few asynchronous column-chunk all-gathers and the GEMM runs chunk by
  chunk behind them (SPARTAN_RHS_CHUNK_COLS columns per chunk, 0 = one blocking gather)."""
  import os
  chunk_cols = int(os.environ.get('SPARTAN_RHS_CHUNK_COLS', '2048'))
  if chunk_cols > 0 and len(array.shape) == 2 and hasattr(array, 'fetch_whole_chunked') \
      and hasattr(context.get().backend, 'dot_chunked'):
    try:
      whole = array.fetch_whole_chunked(chunk_cols)
    except NotImplementedError:   # a transport without the asynchronous collective: same on every rank
      whole = None
    if whole is not None:
      return whole
  return array.fetch(whole_extent)


dot_outer_mapper.fetch_rhs = _fetch_whole_rhs
# every tensor these mappers yield is the output of a kernel launched for it (never an input tile),
# so the target may adopt it on a first full-tile write instead of copying
for _m in (dot_map2_np_mapper, dot_map2_vec_mapper, dot_map2_mapper, dot_outer_mapper):
  _m.yields_fresh_tensors = True


def dot(a, b, tile_hint=None):
  """dot.py:243-299."""
  if isinstance(b, np.ndarray):
    if len(a.shape) == 1 and len(b.shape) == 1:
      shape = (1,)
    elif len(a.shape) > 1 and len(b.shape) == 1:
      shape = (a.shape[0],)
    else:
      shape = (a.shape[0], b.shape[1])
    return map_mod.map2(a, axes=[0], fn=dot_map2_np_mapper, fn_kw={'array2': b}, shape=shape,
                        reducer=np.add)
  if len(a.shape) == 1 and len(b.shape) == 1:
    if a.shape[0] != b.shape[0]:
      raise ValueError('objects are not aligned %d %d' % (a.shape[0], b.shape[0]))
    return map_mod.map2((a, b), (0, 0), fn=dot_map2_vec_mapper, shape=(1,), reducer=np.add)
  elif len(a.shape) == 1 and len(b.shape) > 1:
    if a.shape[0] != b.shape[0]:
      raise ValueError('objects are not aligned %d %d' % (a.shape[0], b.shape[0]))
    shape = (b.shape[1],)
  elif len(a.shape) > 1 and len(b.shape) == 1:
    if a.shape[1] != b.shape[0]:
      raise ValueError('objects are not aligned %d %d' % (a.shape[1], b.shape[0]))
    shape = (a.shape[0],)
  elif len(a.shape) > 1 and len(b.shape) > 1:
    if tile_hint is None:
      tile_hint = (a.shape[0], b.shape[1])
    if a.shape[1] != b.shape[0]:
      raise ValueError('objects are not aligned %d %d' % (a.shape[1], b.shape[0]))
    shape = (a.shape[0], b.shape[1])
  else:
    raise ValueError

  if len(a.shape) > 1 and a.shape[0] > a.shape[1]:
    # rows > cols: row-partitioned outer product (dot.py:281-285)
    return outer_mod.outer((a, b), (0, None), dot_outer_mapper, shape=shape, tile_hint=tile_hint,
                           reducer=np.add)
  elif len(a.shape) > 1:
    # rows <= cols: K-split map2 join (dot.py:286-290)
    return map_mod.map2((a, b), (1, 0), dot_map2_mapper, shape=shape, tile_hint=tile_hint,
                        reducer=np.add)
  else:
    raise NotImplementedError('vector . matrix needs reshape (SURVEY 8f.1)')
