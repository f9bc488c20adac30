"""spartan.dot: which join carries a product, the per-tile bodies, and the collective plan of the K-split.

Role of the reference's spartan/expr/dot.py:172-299.  `tiles[0].dot(tiles[1])` is the fp32 / fp64 MFMA GEMM
(sp_gemm_f32 / sp_gemm_f64) for matrix.matrix and one fused multiply-reduce launch for matrix.vector.

Forms (a is m x k or a k-vector, b is k x n or a k-vector):

  b is a driver-side NumPy array   map2 on a's rows; b is replicated and sliced per tile        (dot.py:172-187)
  vector . vector                   map2 (0, 0): per-tile inner products added into a 1-cell target (:189-191)
  vector . matrix                   the vector as a 1 x k row, then the K-split below             (:296-299)
  m > k  (tall a)                   outer: a's row tile times ALL of b -> disjoint rows of c       (:222-238)
  m <= k (square / wide a)          K-split map2 (1, 0): worker i multiplies the column slab a[:, k_i] by its
                                    own rows b[k_i, :] and the m x n partials are added           (:195-217)

Across GPUs the K-split has one exchange before the GEMMs (the slab a[:, k_i] is spread over every worker's row
tile: an all-to-all of (m/p x k/p) blocks) and one after (the partials are summed into the target).  When both
operands and the target are row-tiled one tile per GPU -- dot(a, b, tile_hint=(m/p, n)) -- `ksplit_plan` runs the
whole join as a pipeline instead of tile by tile:

  * the all-to-all of a's blocks is issued asynchronously; the products that only need LOCAL data run behind it;
  * the blocks are received straight into the row blocks of ONE slab buffer a[:, k_me] (m x k/p), so after the
    first column chunk -- whose rows are multiplied in up to three pieces, my own block first, behind the
    transfer -- every chunk of the partial is ONE large GEMM slab . b[k_me, chunk];
  * the partial is produced in column chunks (m x n_c, double-buffered) and each chunk is reduce-scattered by its
    own asynchronous collective while the next chunk is multiplied: only the last chunk's reduce-scatter is
    exposed, and the 4 GiB partial of the 32768^2 north-star shape never exists (2 x 512 MiB instead).
"""
import os

import numpy as np

from . import map as map_mod
from . import outer as outer_mod
from . import views
from .. import context
from ..array import distarray, extent, tile


# ---- per-tile products -------------------------------------------------------------------------------------
def _product_shape(ash, bsh):
  if len(ash) == 1 and len(bsh) == 1:
    return ()
  if len(bsh) == 1:
    return tuple(ash[:-1])
  if len(ash) == 1:
    return tuple(bsh[1:])
  return (ash[0], bsh[1])


def _dot(a, b):
  """`a.dot(b)` on backend tensors; a shape/dtype placeholder on ranks that do not execute the tile."""
  be = context.get().backend
  tile.reject_masked((a, b), 'dot')
  if isinstance(b, distarray.ChunkedWhole):
    if isinstance(a, distarray.Absent):
      return distarray.Absent((a.shape[0], b.shape[1]), a.dtype)
    return be.dot_chunked(a, b)
  if isinstance(a, distarray.Absent) or isinstance(b, distarray.Absent):
    return distarray.Absent(_product_shape(tuple(a.shape), tuple(b.shape)), a.dtype)
  return be.dot(a, b)


def _is_backend_tensor(v):
  """A driver-side operand the driver keeps where the tiles are (examples/lreg.py: the weights between two steps)."""
  return hasattr(v, 'data_ptr') and hasattr(v, 'strides')


def dot_map2_np_mapper(extents, tiles, array2):
  """Row tile of `a` times the matching rows of the driver array (its rows follow a's LAST axis)."""
  be = context.get().backend
  ex = extents[0]
  k_axis = len(ex.ul) - 1
  rhs = be.cached_numpy(array2, (slice(ex.ul[k_axis], ex.lr[k_axis]),))
  out = _dot(tiles[0], rhs)
  if len(ex.ul) == 1:                       # vector . host vector: a partial inner product
    yield extent.create((0,), (1,), (1,)), out.reshape(1,)
  elif array2.ndim == 1:                    # matrix . host vector: this tile's rows of the result
    yield extent.create((ex.ul[0],), (ex.lr[0],), (ex.array_shape[0],)), out
  else:
    n = array2.shape[1]
    yield extent.create((ex.ul[0], 0), (ex.lr[0], n), (ex.array_shape[0], n)), out


def dot_map2_vec_mapper(extents, tiles):
  yield extent.create((0,), (1,), (1,)), _dot(tiles[0], tiles[1]).reshape(1,)


def dot_map2_mapper(extents, tiles, is_vec=None):
  """Partial product of the K-split: a[:, k_i] . b[k_i, :] covers the WHOLE result."""
  slab, rows = tiles
  if is_vec:
    slab = slab.reshape(extents[0].shape[1],)
    shape = (extents[1].shape[1],)
  elif len(rows.shape) == 1:
    shape = (extents[0].shape[0],)
  else:
    shape = (extents[0].shape[0], extents[1].shape[1])
  yield extent.from_shape(shape), _dot(slab, rows)


dot_map2_mapper.is_contraction = True       # (expr/map.py warns once when such a join meets grid-tiled operands)


def dot_outer_mapper(ex_a, tile_a, ex_b, tile_b):
  """Row block of a times all of b: its own rows of the result, no reduction."""
  if len(tile_b.shape) == 1:
    target = extent.create((ex_a.ul[0],), (ex_a.lr[0],), (ex_a.array_shape[0],))
  else:
    target = extent.create((ex_a.ul[0], ex_b.ul[1]), (ex_a.lr[0], ex_b.lr[1]),
                           (ex_a.array_shape[0], ex_b.array_shape[1]))
  yield target, _dot(tile_a, tile_b)


def _fetch_whole_rhs(array, whole_extent):
  """dot_outer_mapper's b: every worker needs ALL of it (outer.py:21-29).  Across GPUs the gather is issued as a
  few asynchronous column-chunk all-gathers and the GEMM runs chunk by chunk behind them
  (SPARTAN_RHS_CHUNK_COLS columns per chunk, 0 = one blocking gather)."""
  chunk_cols = int(os.environ.get('SPARTAN_RHS_CHUNK_COLS', '2048'))
  if chunk_cols > 0 and len(array.shape) == 2 and hasattr(array, 'fetch_whole_chunked') \
      and hasattr(context.get().backend, 'dot_chunked'):
    whole = array.fetch_whole_chunked(chunk_cols)
    if whole is not None:
      return whole
  return array.fetch(whole_extent)


# ---- the K-split across GPUs as one pipeline ------------------------------------------------------------------
def _row_tiles_in_rank_order(array, ctx):
  """[(extent, tile id)] if `array` is a plain dense 2-D array cut into world.size equal row tiles, tile r on
  rank r; else None."""
  p = ctx.world.size
  if not isinstance(array, distarray.DistArrayImpl) or array.sparse or len(array.shape) != 2:
    return None
  rows, cols = array.shape
  if len(array.tiles) != p or rows % p:
    return None
  step = rows // p
  tiles = sorted(array.tiles.items(), key=lambda kv: kv[0].ul)
  for r, (ex, tid) in enumerate(tiles):
    if ctx.rank_of(tid.worker) != r or ex.ul != (r * step, 0) or ex.lr != ((r + 1) * step, cols):
      return None
  return tiles


def _chunk_columns(n):
  """Columns of the partial per reduce-scatter (SPARTAN_DOT_CHUNK_COLS, default 4096): whole chunks only."""
  want = int(os.environ.get('SPARTAN_DOT_CHUNK_COLS', '4096'))
  if want <= 0 or want >= n:
    return n
  while n % want:
    want -= 1
  return want


def ksplit_plan(arrays, axes, target, fn_kw):
  """Run map2((a, b), (1, 0), dot_map2_mapper) into `target` as the pipeline described in the module docstring.
  Returns False (nothing done) unless the pattern is the regular one.  The decision reads array metadata, which
  every rank holds identically, plus ONE thing only the owner knows -- whether its operand tiles hold plain, fully
  written data (a never-written tile is an EmptyBlob, a partly written one a MaskedBlob) -- and that is agreed
  over the control plane before any transfer is issued, so every rank takes the same branch."""
  ctx = context.get()
  be, world = ctx.backend, ctx.world
  if not world.distributed or ctx.num_workers != world.size or fn_kw or tuple(axes) != (1, 0):
    return False
  if len(arrays) != 2 or not hasattr(be, 'gemm_into'):
    return False
  a, b = arrays
  ta, tb, tt = (_row_tiles_in_rank_order(x, ctx) for x in (a, b, target))
  if ta is None or tb is None or tt is None or getattr(target, '_touched', False):
    return False
  dt = np.dtype(target.dtype)
  if dt not in (np.dtype(np.float32), np.dtype(np.float64)) or np.dtype(a.dtype) != dt or np.dtype(b.dtype) != dt:
    return False
  if be.reducer_name(target.reducer_fn) != 'ADD':
    return False
  p, me = world.size, world.rank
  m, k = a.shape
  n = b.shape[1]
  if k % p:
    return False
  mb, kb = m // p, k // p
  my_a = ctx.tile(ta[me][1]).get(be, None)
  my_b = ctx.tile(tb[me][1]).get(be, None)
  if not (getattr(a, '_all_plain', False) and getattr(b, '_all_plain', False)):
    # (agreed once per array: written cells never become unwritten, so the answer is kept as array metadata)
    plain = [not isinstance(t, (tile.EmptyBlob, tile.MaskedBlob)) for t in (my_a, my_b)]
    votes = world.all_gather_object(plain)
    a._all_plain = all(v[0] for v in votes)
    b._all_plain = all(v[1] for v in votes)
    if not (a._all_plain and b._all_plain):
      return False    # some rank's tile is unwritten or masked: every rank takes the generic tile-by-tile join

  result = ksplit_pipeline(be, p, me, my_a, my_b, dt, world.exchange_async,
                           lambda out, part: world.reduce_scatter_async(out, part, 'ADD'), _chunk_columns(n))
  target._touched = True
  target.mark_written()
  ctx.tile(tt[me][1]).update(be, None, result, target.reducer_fn, owned=True)
  return True


def ksplit_pipeline(be, p, me, my_a, my_b, dt, exchange_async, reduce_scatter_async, nc):
  """One rank's share of the K-split dot: my row tile of a (m/p x k), my rows of b (k/p x n) -> my row tile of the
  result (m/p x n).  `exchange_async(sends, recvs)` / `reduce_scatter_async(out, part)` are the transport's
  (World's on a real job; bench.py --emulate-rank substitutes same-size device copies on a side stream to time a
  rank's share on ONE GPU).

    1. the slab a[:, k_me] (m x k/p) is ONE buffer whose row block j is the receive buffer of rank j's block, so
       no block is copied after it lands; the all-to-all is asynchronous, my own block is a local copy;
    2. the partial is produced in column chunks of nc columns, two chunk buffers: in chunk 0 the product of my
       OWN block runs first (behind the transfer), the rows above and below it after the blocks have landed; every
       later chunk is ONE GEMM slab . b[:, chunk] -- (m x k/p) . (k/p x nc), thousands of macro-tiles per launch
       instead of p launches of one partly filled round each;
    3. each finished chunk is reduce-scattered asynchronously while the next one is multiplied; the reduced
       pieces are pasted into the result as they land.  Only the last chunk's reduce-scatter is exposed."""
  mb, k = my_a.shape
  kb = k // p
  n = my_b.shape[1]
  m = mb * p
  slab = be.empty((m, kb), dt)
  rows = lambda j0, j1: slice(j0 * mb, j1 * mb)           # noqa: E731
  sends = [(j, be.copy(my_a[:, j * kb:(j + 1) * kb])) for j in range(p) if j != me]
  recvs = [(j, slab[rows(j, j + 1), :]) for j in range(p) if j != me]
  be.paste(slab, (rows(me, me + 1), slice(0, kb)), my_a[:, me * kb:(me + 1) * kb])
  arriving = exchange_async(sends, recvs)

  bufs = [be.empty((m, nc), dt), be.empty((m, nc), dt) if nc < n else None]
  result = be.empty((mb, n), dt)
  in_flight = None                                          # (c0, reduced piece, handle) of the previous chunk

  def land(item):
    c0, piece, handle = item
    if handle is not None:
      handle.wait()
    be.paste(result, (slice(0, mb), slice(c0, c0 + nc)), piece)

  for ci, c0 in enumerate(range(0, n, nc)):
    part = bufs[ci % 2]
    b_chunk = my_b[:, c0:c0 + nc]
    if ci == 0:
      be.gemm_into(slab[rows(me, me + 1), :], b_chunk, part[rows(me, me + 1), :])     # needs no transfer
      if arriving is not None:
        arriving.wait()
      for j0, j1 in ((0, me), (me + 1, p)):
        if j1 > j0:
          be.gemm_into(slab[rows(j0, j1), :], b_chunk, part[rows(j0, j1), :])
    else:
      be.gemm_into(slab, b_chunk, part)
    piece = be.empty((mb, nc), dt)
    handle = reduce_scatter_async(piece, part)
    if in_flight is not None:
      land(in_flight)                                       # its buffer is the one the NEXT chunk overwrites
    in_flight = (c0, piece, handle)
  land(in_flight)
  del sends
  return result


dot_map2_mapper.collective_plan = ksplit_plan
dot_outer_mapper.fetch_rhs = _fetch_whole_rhs
# every tensor these mappers yield is the output of a kernel launched for it (never an input tile), so the
# target may adopt it on a first full-tile write instead of copying
for _m in (dot_map2_np_mapper, dot_map2_vec_mapper, dot_map2_mapper, dot_outer_mapper):
  _m.yields_fresh_tensors = True


# ---- dispatch --------------------------------------------------------------------------------------------------
def dot(a, b, tile_hint=None):
  """The product of two arrays (1-D or 2-D each); `b` may be a driver-side NumPy array.  tile_hint tiles the
  result (default: ONE tile, as the reference, dot.py:277-278; (m/p, n) gives the reduce-scatter target)."""
  ra, rb = len(a.shape), len(b.shape)
  if ra not in (1, 2) or rb not in (1, 2):
    raise ValueError('dot of %d-d and %d-d arrays' % (ra, rb))
  if a.shape[ra - 1] != b.shape[0]:
    raise ValueError('objects are not aligned %d %d' % (a.shape[ra - 1], b.shape[0]))
  shape = _product_shape(tuple(a.shape), tuple(b.shape)) or (1,)
  if isinstance(b, np.ndarray) or _is_backend_tensor(b):
    return map_mod.map2(a, axes=[0], fn=dot_map2_np_mapper, fn_kw={'array2': b}, shape=shape, reducer=np.add)
  if ra == 1 and rb == 1:
    return map_mod.map2((a, b), (0, 0), fn=dot_map2_vec_mapper, shape=shape, reducer=np.add)
  if ra == 2 and rb == 2 and tile_hint is None:
    tile_hint = shape
  if ra == 1:
    row = views.reshape(a, (1, a.shape[0]))
    return map_mod.map2((row, b), (1, 0), dot_map2_mapper, fn_kw={'is_vec': True}, shape=shape,
                        tile_hint=tile_hint, reducer=np.add)
  if a.shape[0] > a.shape[1]:
    return outer_mod.outer((a, b), (0, None), dot_outer_mapper, shape=shape, tile_hint=tile_hint, reducer=np.add)
  return map_mod.map2((a, b), (1, 0), dot_map2_mapper, shape=shape, tile_hint=tile_hint, reducer=np.add)
