"""Distributed arrays: tiling, fetch (gather + stitch) and update (split +
scatter-with-reduce).

Host-side mirror of the reference's spartan/array/distarray.py.  Tile payloads
live in HBM as backend tensors; the reference's per-tile `get`/`update` RPCs
become grouped RCCL point-to-point transfers or, for the regular patterns of
the hot path, one collective (see `UpdateBatch.flush` and `DistArrayImpl.glom`).
"""
import collections
import itertools
import math

import numpy as np

from . import extent, placement, tile
from .. import context
from ..context import LocalKernelResult, TileId
from ..util import Assert

# number of elements per tile (distarray.py:19)
DEFAULT_TILE_SIZE = 100000


def take_first(a, b):
  return a


def good_tile_shape(shape, num_shards=-1):
  """Default tile shape: about prod(shape) / num_shards elements per tile (DEFAULT_TILE_SIZE without a shard
  count), spent on the innermost axes first -- so a 2-D array is cut into bands of whole rows.  Integer results
  are the reference's (distarray.py:26-48, with its Python-2 floor divisions)."""
  budget = DEFAULT_TILE_SIZE if num_shards == -1 else int(np.prod(shape, dtype=np.int64)) // num_shards
  tile_shape = [1] * len(shape)
  for axis in reversed(range(len(shape))):
    if budget <= 1:
      break
    tile_shape[axis] = min(shape[axis], budget)
    budget //= shape[axis]
  return tile_shape


def compute_splits(shape, tile_hint):
  """Per axis, the [start, end) pieces of length tile_hint[axis] (the last one shorter)."""
  return [[(lo, min(lo + step, n)) for lo in range(0, n, step)] for n, step in zip(shape, tile_hint)]


_cuts = {}          # (shape, tile hint, shards) -> ((extent, shard index), ...): extents never change, cuts repeat


def compute_extents(shape, tile_hint=None, num_shards=-1):
  """{tile extent: shard index} for an array of `shape`, tiles in row-major order of their position, dealt to
  the shards round-robin (reference distarray.py:73-110)."""
  if len(shape) == 0:
    return {extent.create([], [], ()): 0}
  try:
    key = (tuple(shape), None if tile_hint is None else tuple(tile_hint), num_shards)
    known = _cuts.get(key)
  except TypeError:
    key = known = None
  if known is not None:
    return collections.OrderedDict(known)
  if tile_hint is None:
    tile_hint = good_tile_shape(shape, num_shards)
  elif len(tile_hint) != len(shape):
    raise AssertionError('#dimensions in tile hint does not match shape %s vs %s' % (tile_hint, shape))
  tiles = collections.OrderedDict()
  for position, box in enumerate(itertools.product(*compute_splits(shape, tile_hint))):
    lows, highs = zip(*box)
    tiles[extent.create(lows, highs, shape)] = position if num_shards == -1 else position % num_shards
  if key is not None and len(tiles) <= 4096:
    if len(_cuts) > 512:
      _cuts.clear()
    _cuts[key] = tuple(tiles.items())
  return tiles


def _tile_mapper(tile_id, blob, array=None, user_fn=None, **kw):
  """distarray.py:113-116."""
  ex = array.extent_for_blob(tile_id)
  return user_fn(ex, **kw)


class Absent(object):
  """Stands for tile data that lives on another rank (this rank is not the one executing the current mapper).
  It carries shape and dtype only, and answers the ndarray calls a user mapper makes on its tiles (`.T`,
  `.reshape`, `.sum(axis, keepdims=...)`, `.dot`, arithmetic, NumPy ufuncs, slicing) with another placeholder of
  the right shape and dtype -- so the SAME mapper runs on every rank, yields the same extents everywhere, and only
  the executing rank touches data."""
  __slots__ = ('shape', 'dtype')
  __array_priority__ = 2000.0

  def __init__(self, shape, dtype):
    self.shape = tuple(int(s) for s in shape)
    self.dtype = np.dtype(dtype)

  ndim = property(lambda self: len(self.shape))
  size = property(lambda self: int(np.prod(self.shape, dtype=np.int64)))

  def _like(self, fn, *others, **kw):
    """Shape / dtype of fn(self, *others) as NumPy would compute it, found on zero-stride dummies (no data)."""
    def dummy(x):
      if isinstance(x, Absent) or (hasattr(x, 'shape') and hasattr(x, 'dtype') and not isinstance(x, (np.ndarray, np.generic))):
        return np.broadcast_to(np.zeros((), np.dtype(x.dtype)), tuple(x.shape))
      return x
    with np.errstate(all='ignore'):
      res = fn(*[dummy(a) for a in (self,) + others], **kw)
    res = np.asarray(res)
    return Absent(res.shape, res.dtype)

  def reshape(self, *shape):
    if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
      shape = tuple(shape[0])
    return self._like(lambda a: a.reshape(shape))

  def transpose(self, *axes):
    return self._like(lambda a: a.transpose(*axes))

  T = property(lambda self: self.transpose())

  def astype(self, dtype, **kw):
    return Absent(self.shape, dtype)

  def copy(self):
    return self

  def dot(self, other):
    return _absent_dot(self, other)

  def __getitem__(self, idx):
    return self._like(lambda a: a[idx])

  def __array_ufunc__(self, ufunc, method, *inputs, **kw):
    kw.pop('out', None)
    first = inputs[0] if isinstance(inputs[0], Absent) else Absent(np.shape(inputs[0]), np.asarray(inputs[0]).dtype if not hasattr(inputs[0], 'dtype') else inputs[0].dtype)
    return first._like(lambda *a: getattr(ufunc, method)(*a, **kw), *inputs[1:])

  def __array_function__(self, func, types, args, kwargs):
    def dummy(x):
      if isinstance(x, (list, tuple)):
        return type(x)(dummy(v) for v in x)
      if hasattr(x, 'shape') and hasattr(x, 'dtype') and not isinstance(x, (np.ndarray, np.generic)):
        return np.broadcast_to(np.zeros((), np.dtype(x.dtype)), tuple(x.shape))
      return x
    with np.errstate(all='ignore'):
      res = np.asarray(func(*dummy(list(args)), **kwargs))
    return Absent(res.shape, res.dtype)


def _absent_dot(a, b):
  ash, bsh = tuple(a.shape), tuple(b.shape) if hasattr(b, 'shape') else tuple(np.shape(b))
  dt = np.result_type(a.dtype, b.dtype if hasattr(b, 'dtype') else np.asarray(b).dtype)
  if len(ash) == 1 and len(bsh) == 1:
    return Absent((), dt)
  if len(bsh) == 1:
    return Absent(ash[:-1], dt)
  if len(ash) == 1:
    return Absent(bsh[1:], dt)
  return Absent(ash[:-1] + bsh[1:], dt)


def _absent_method(name):
  def method(self, *args, **kw):
    return self._like(lambda a: getattr(a, name)(*args, **kw))
  return method


def _absent_binary(ufunc, swap=False):
  def op(self, other):
    return self._like((lambda a, b: ufunc(b, a)) if swap else ufunc, other)
  return op


for _n in ('sum', 'prod', 'max', 'min', 'mean', 'all', 'any', 'argmax', 'argmin', 'ravel', 'flatten', 'squeeze', 'swapaxes'):
  setattr(Absent, _n, _absent_method(_n))
for _n, _u in dict(add=np.add, sub=np.subtract, mul=np.multiply, truediv=np.true_divide, floordiv=np.floor_divide,
                   mod=np.remainder, pow=np.power).items():
  setattr(Absent, '__%s__' % _n, _absent_binary(_u))
  setattr(Absent, '__r%s__' % _n, _absent_binary(_u, swap=True))
for _n, _u in dict(lt=np.less, le=np.less_equal, gt=np.greater, ge=np.greater_equal).items():
  setattr(Absent, '__%s__' % _n, _absent_binary(_u))
Absent.__neg__ = lambda self: self
Absent.__abs__ = lambda self: self
Absent.__matmul__ = lambda self, other: _absent_dot(self, other)


def _slices_shape(slices, base_shape):
  shp = []
  for slc, n in zip(slices, base_shape):
    start, stop, _ = slc.indices(n)
    shp.append(max(stop - start, 0))
  return tuple(shp)


def _abstract(name):
  def missing(self, *args, **kw):
    raise NotImplementedError('%s.%s' % (type(self).__name__, name))
  return missing


class DistArray(object):
  """What every array-like of the tile path offers (reference distarray.py:119-215): `fetch(extent)` a region,
  `update(extent, data)` a region through the reducer, `foreach_tile(mapper_fn, kw)` the SPMD tile walk and
  `extent_for_blob(tile_id)`; the rest is derived."""
  fetch = _abstract('fetch')
  update = _abstract('update')
  foreach_tile = _abstract('foreach_tile')
  extent_for_blob = _abstract('extent_for_blob')

  ndim = property(lambda self: len(self.shape))

  def real_size(self):
    return math.prod(self.shape)

  def __len__(self):
    return self.shape[0]

  def __hash__(self):
    return id(self)

  def __repr__(self):
    return '%s(id=%s, shape=%s, dtype=%s)' % (self.__class__.__name__, id(self), self.shape, self.dtype)

  def select(self, idx):
    """array[idx] for an extent, a scalar index or (a tuple of) slices."""
    if isinstance(idx, extent.TileExtent):
      return self.fetch(idx)
    if np.isscalar(idx):
      return self.select(slice(idx, idx + 1))[0]
    return self.fetch(extent.from_slice(idx, self.shape))

  __getitem__ = select

  def glom(self):
    """The whole array as one host array, on every rank (a scipy matrix for a sparse array, a numpy.ma array if
    cells were never written -- like the reference's glom)."""
    ctx = context.get()
    whole = self.select(np.index_exp[:])
    if tile.is_sparse_blob(whole):
      return ctx.backend.sparse_to_host(whole)
    if isinstance(whole, tile.MaskedBlob):
      return whole.to_host(ctx.backend)
    return ctx.backend.to_numpy(whole)

  def map_to_array(self, mapper_fn, kw=None):
    """foreach_tile whose mapper returns the tiles of a NEW array: [(extent, tile id)] per input tile."""
    made = self.foreach_tile(mapper_fn=mapper_fn, kw=kw)
    table = collections.OrderedDict()
    for produced in made.values():
      for ex, tile_id in produced:
        table[ex] = tile_id
    return from_table(table, meta=getattr(made, 'meta', None))


class ChunkedWhole(object):
  """A whole 2-D array replicated as column chunks that may still be in flight:
  chunks = [(col0, col1, tensor (rows, col1-col0), work handle or None, keep-alive)]."""

  def __init__(self, shape, dtype, chunks):
    self.shape = tuple(shape)
    self.dtype = np.dtype(dtype)
    self.chunks = chunks

  def ready(self, i):
    """Make the current stream wait for chunk i and return (col0, col1, tensor)."""
    c0, c1, t, work, _ = self.chunks[i]
    if work is not None:
      work.wait()
      self.chunks[i] = (c0, c1, t, None, None)
    return c0, c1, t


class UpdateBatch(object):
  """Updates issued while one kernel (foreach_tile) runs, joined at its end --
  the reference collects `target.update(..., wait=False)` futures and joins them
  after the tile loop (worker.py:266, rpc FutureGroup).  Batching lets the
  regular patterns of the hot path be carried by ONE RCCL collective instead of
  per-tile messages:

    every worker contributes a partial covering the WHOLE target, reducer is
    np.add / maximum / minimum and
      - the target is one tile             -> reduce to its owner
      - the target is evenly tiled, one
        contiguous chunk per rank in order -> reduce-scatter
    anything else                          -> grouped send/recv + merge at owner,
                                              in the deterministic tile order.
  """

  def __init__(self, ctx):
    self.ctx = ctx
    self.items = []  # (array, exec_worker, region, data, owned)

  def add(self, array, region, data, owned):
    self.items.append((array, self.ctx.current_worker, region, data, owned))

  # -- helpers
  def _collective_plan(self, array, items):
    """Return ('reduce', dst_rank) / ('reduce_scatter',) / None."""
    ctx = self.ctx
    world = ctx.world
    if not world.distributed:
      return None
    red = ctx.backend.reducer_name(array.reducer_fn)
    if red not in ('ADD', 'MAX', 'MIN', 'MUL'):
      return None
    if np.dtype(array.dtype).kind not in 'fiu':
      return None
    if ctx.num_workers != world.size:
      return None
    # exactly one whole-array contribution per rank, from that rank's own worker
    if len(items) != world.size:
      return None
    seen = set()
    for (_, worker, region, data, _) in items:
      if worker is None or region.shape != tuple(array.shape) or region.ul != (0,) * len(array.shape):
        return None
      seen.add(ctx.rank_of(worker))
    if len(seen) != world.size:
      return None
    # untouched target only: the first write replaces (tile.pyx:263-268), so
    # "reduce of the partials" is exactly what arrival-order merging produces
    # (`_touched` is array metadata kept identically on every rank)
    if getattr(array, '_touched', False):
      return None
    tiles = list(array.tiles.items())
    if len(tiles) == 1:
      return ('reduce', ctx.rank_of(tiles[0][1].worker))
    if len(tiles) == world.size and len(array.shape) >= 1:
      # contiguous equal chunks in rank order <=> split along dim 0 only
      n0 = array.shape[0]
      if n0 % world.size != 0:
        return None
      step = n0 // world.size
      for ex, tid in tiles:
        r = ctx.rank_of(tid.worker)
        if ex.ul[0] != r * step or ex.lr[0] != (r + 1) * step:
          return None
        if ex.ul[1:] != (0,) * (len(array.shape) - 1) or ex.lr[1:] != tuple(array.shape[1:]):
          return None
      return ('reduce_scatter',)
    return None

  def _merge_whole_partials(self, array, items):
    """ONE process, several logical workers: every item is a dense partial covering the WHOLE of a freshly created
    target that is cut by rows (sum(axis=0) over p row tiles: p partials of the full length into p target tiles).
    Piece by piece that is p x p small merges (64 launches of ~5 us for 8 workers); here the partials are combined
    whole, in issue order -- the first replaces, the others go through the reducer: element for element the order
    the piecewise merges apply -- and the target tiles adopt their row ranges of the result as views.  What RCCL's
    reduce-scatter does for one worker per GPU (_collective_plan).  Returns False, having done nothing, otherwise."""
    ctx = self.ctx
    be = ctx.backend
    if ctx.world.distributed or len(items) < 2 or len(array.tiles) < 2 or getattr(array, '_touched', False):
      return False
    if array.written is None or array.written or array.reducer_fn is None:
      return False
    if be.reducer_name(array.reducer_fn) not in ('ADD', 'MAX', 'MIN', 'MUL') or np.dtype(array.dtype).kind not in 'fiu':
      return False
    shape, nd = tuple(array.shape), len(array.shape)
    if nd < 1:
      return False
    for ex in array.tiles:
      if ex.ul[1:] != (0,) * (nd - 1) or ex.lr[1:] != shape[1:]:
        return False                   # (not cut by rows alone: a tile's part of the result would not be contiguous)
    on_device = getattr(be, 'name', '') == 'hip'
    for (_, worker, region, data, _) in items:
      if region.shape != shape or region.ul != (0,) * nd or isinstance(data, (Absent, np.generic)) \
              or tile.is_sparse_blob(data) or isinstance(data, (tile.MaskedBlob, tile.EmptyBlob)) \
              or tuple(getattr(data, 'shape', ())) != shape or (on_device and not hasattr(data, 'data_ptr')):
        return False                   # (placeholders, scalars, masked / sparse blocks, a host array on the device backend)
    first = items[0]
    acc = be.astype(first[3], array.dtype)
    if not first[4] and be.same_memory(acc, first[3]):
      acc = be.copy(acc)               # never alias a caller's tensor
    acc = be.contiguous(acc)
    for (_, _, _, data, _) in items[1:]:
      be.update_box(acc, [0] * nd, shape, data, array.reducer_fn, tile.MASK_ALL_SET, None)
    array._touched = True
    array.mark_written()
    for ex, tid in array.tiles.items():
      ctx.tile(tid).update(be, None, acc[ex.ul[0]:ex.lr[0]], array.reducer_fn, owned=True)
    return True

  def flush(self):
    ctx = self.ctx
    be = ctx.backend
    world = ctx.world
    by_array = collections.OrderedDict()
    for it in self.items:
      by_array.setdefault(id(it[0]), []).append(it)
    self.items = []
    for items in by_array.values():
      array = items[0][0]
      if array.sparse:
        self._flush_sparse(array, items)
        continue
      if self._merge_whole_partials(array, items):
        continue
      plan = self._collective_plan(array, items)
      if plan is not None:
        mine = [it for it in items if ctx.is_local_worker(it[1])][0]
        data = mine[3]
        red = be.reducer_name(array.reducer_fn)
        array._touched = True
        array.mark_written()
        data = be.astype(data, array.dtype)
        if plan[0] == 'reduce':
          buf = data if (mine[4] or data is not mine[3]) else be.copy(data)
          world.reduce(buf, plan[1], red)
          if world.rank == plan[1]:
            (ex, tid), = array.tiles.items()
            ctx.tile(tid).update(be, None, buf, array.reducer_fn, owned=True)
        else:
          my_tid = [tid for tid in array.tiles.values() if ctx.is_local(tid)][0]
          t = ctx.tile(my_tid)
          out = be.empty(t.shape, t.dtype)
          world.reduce_scatter(out, be.contiguous(data), red)
          t.update(be, None, out, array.reducer_fn, owned=True)
        continue
      # generic path: explicit transfers, merged in issue order
      array._touched = True
      sends, recvs, merges = [], [], []
      for (_, worker, region, data, owned) in items:
        exec_rank = world.rank if worker is None else ctx.rank_of(worker)
        for tile_id, src_slice, dst_slice in array._update_splits(region):
          owner = ctx.rank_of(tile_id.worker)
          whole = _slices_shape(src_slice, region.shape) == tuple(region.shape)
          tile_ex = array.blob_to_ex[tile_id]
          if array.written is not None and _slices_shape(dst_slice, tile_ex.shape) == tuple(tile_ex.shape):
            array.mark_written(tile_ex)
          if exec_rank == world.rank and owner == world.rank:
            piece = data if whole else data[src_slice]
            merges.append((tile_id, dst_slice, piece, owned and whole))
          elif worker is None:
            pass   # driver-level update: the data is replicated, every owner merges its own part
          elif exec_rank == world.rank:
            piece = data if whole else data[src_slice]
            sends.append((owner, be.contiguous(be.astype(piece, array.dtype))))
          elif owner == world.rank:
            buf = be.empty(_slices_shape(src_slice, region.shape), array.dtype)
            recvs.append((exec_rank, buf))
            merges.append((tile_id, dst_slice, buf, True))
      if sends or recvs:
        world.exchange(sends, recvs)
      for tile_id, dst_slice, piece, owned in merges:
        t = ctx.tile(tile_id)
        full = tuple(piece.shape) == t.shape
        t.update(be, None if full else dst_slice, piece, array.reducer_fn, owned=owned and full)


def _ship_sparse(ctx, outgoing):
  """Sparse blocks between ranks, device to device.  `outgoing`: {key: (destination ranks, block)} for the blocks
  this rank sends; every rank calls this at the same point.  Returns {key: block} of what arrived here.
  The three arrays of each block (row pointers, columns, values) travel in ONE grouped exchange; only the shapes
  and entry counts go through the control plane first.  (The reference pickles the scipy objects into its RPCs,
  core.py / rpc/zeromq.py; round 1 all-gathered host objects here.)"""
  be, world = ctx.backend, ctx.world
  parts = {k: be.sparse_parts(b) for k, (_, b) in outgoing.items()}
  mine = {k: (tuple(b.shape), np.dtype(be.dtype_of(b)).str, int(parts[k][1].size), sorted(dsts))
          for k, (dsts, b) in outgoing.items()}
  sends, recvs, arriving = [], [], {}
  for src, listing in enumerate(world.all_gather_object(mine)):
    for k in sorted(listing):
      shape, dstr, nnz, dsts = listing[k]
      for dst in dsts:
        if dst == src:
          continue
        if src == world.rank and nnz:
          sends.extend((dst, t) for t in parts[k])
          world.stats['sparse_blocks'] += 1
        if dst == world.rank:
          bufs = be.sparse_parts_empty(shape, np.dtype(dstr), nnz) if nnz else None
          arriving[k] = (shape, np.dtype(dstr), bufs)
          if nnz:
            recvs.extend((src, t) for t in bufs)
  world.exchange(sends, recvs)
  return {k: (be.sparse_from_parts(shape, dt, bufs) if bufs is not None else be.sparse_empty(shape, dt))
          for k, (shape, dt, bufs) in arriving.items()}


def _flush_sparse(self, array, items):
  """Updates of a SPARSE target: blocks whose tile lives on the executing rank merge on the device;
  blocks for other ranks travel device to device in one grouped exchange per kernel (_ship_sparse)."""
  ctx = self.ctx
  be = ctx.backend
  world = ctx.world
  array._touched = True
  plan = []       # (seq, tile_id, dst_slice, exec_rank, owner, src_slice, whole)
  for seq, (_, worker, region, data, owned) in enumerate(items):
    exec_rank = world.rank if worker is None else ctx.rank_of(worker)
    for tile_id, src_slice, dst_slice in array._update_splits(region):
      owner = ctx.rank_of(tile_id.worker)
      whole = _slices_shape(src_slice, region.shape) == tuple(region.shape)
      plan.append((seq, tile_id, dst_slice, exec_rank, owner, src_slice, whole, worker is None))
  crossing = world.distributed and any((not drv) and ex_r != ow for (_, _, _, ex_r, ow, _, _, drv) in plan)
  outbox = {}
  local = {}
  for k, (seq, tile_id, dst_slice, exec_rank, owner, src_slice, whole, drv) in enumerate(plan):
    if exec_rank != world.rank:
      continue
    data = items[seq][3]
    if drv and owner != world.rank:
      continue      # driver-level update: the data is replicated, every owner merges its own part
    piece = data if whole else be.sparse_slice(data, src_slice)
    if owner == world.rank:
      local[k] = piece
    else:
      outbox[k] = ([owner], piece)
  inbox = _ship_sparse(ctx, outbox) if crossing else {}
  for k, (seq, tile_id, dst_slice, exec_rank, owner, src_slice, whole, drv) in enumerate(plan):
    if owner != world.rank:
      continue
    if k in local:
      piece = local[k]
    elif k in inbox:
      piece = be.sparse_blob(inbox[k], array.dtype)
    else:
      continue
    t = ctx.tile(tile_id)
    full = tuple(piece.shape) == t.shape
    t.update(be, None if full else dst_slice, piece, array.reducer_fn)


UpdateBatch._flush_sparse = _flush_sparse


class DistArrayImpl(DistArray):
  """distarray.py:223-422."""
  _ids = itertools.count()

  def __init__(self, shape, dtype, tiles, reducer_fn, sparse=False):
    self.shape = tuple(int(s) for s in shape)
    self.dtype = np.dtype(dtype)
    self.reducer_fn = reducer_fn
    self.sparse = sparse
    self.bad_tiles = []
    self.ctx = context.get()
    Assert.not_null(dtype)
    self.blob_to_ex = blob_to_ex = {}
    for k, v in tiles.items():
      if type(k) is not extent.TileExtent or type(v) is not TileId:
        Assert.isinstance(k, extent.TileExtent)
        Assert.isinstance(v, TileId)
      blob_to_ex[v] = k
    self.tiles = tiles
    # which tiles are known to be written everywhere: None = all of them (arrays made of produced tiles); a set of
    # tile extents for an array that was created empty (`create`) and is filled by updates.  Array METADATA: every
    # rank sees every update's region, so every rank keeps the same set -- a fetch that crosses ranks uses it to
    # decide, identically everywhere, whether written-cells masks have to travel with the data.
    self.written = None
    self.id = next(DistArrayImpl._ids)
    self.ctx.register_array(self)
    pend = self.ctx.pending_destructors
    if pend:
      self.ctx.destroy_all(pend)
      del pend[:]

  def __del__(self):
    # destruction is deferred to the next safe point (distarray.py:256-268)
    try:
      self.ctx.pending_destructors.extend(self.tiles.values())
    except Exception:
      pass

  def extent_for_blob(self, id):
    return self.blob_to_ex[id]

  def mark_written(self, tile_extent=None):
    """The whole array (None) or one of its tiles has been written completely."""
    if tile_extent is None or self.written is None:
      self.written = None
      return
    self.written.add(tile_extent)
    if len(self.written) == len(self.tiles):
      self.written = None

  def _maybe_unwritten(self, splits):
    """Could a read of these tiles meet written AND never-written cells?  (An array nothing was written to yet is
    all placeholders on every rank -- creation mappers read its tiles for their shape only -- and needs no masks.)"""
    return (self.written is not None and getattr(self, '_touched', False) and
            any(ex not in self.written for ex, _ in splits))

  def tile_shape(self):
    """distarray.py:276-281: most common tile shape."""
    scounts = collections.defaultdict(int)
    for ex in self.tiles.keys():
      scounts[ex.shape] += 1
    return sorted(scounts.items(), key=lambda kv: (kv[1], kv[0]))[-1][0]

  # -- kernel dispatch ---------------------------------------------------------
  def foreach_tile(self, mapper_fn, kw=None):
    """distarray.py:283-292 + blob_ctx.map (blob_ctx.py:256-275) +
    Worker._run_kernel (worker.py:232-315).  The tile loop runs on every rank
    in the same order; see context.Context."""
    if kw is None:
      kw = {}
    return run_kernel(self, list(self.tiles.values()), mapper_fn, kw)

  # -- data plane --------------------------------------------------------------
  def _tile_piece(self, tile_id, ex, intersection):
    """Tile.get(offset_slice) on the owning rank."""
    ctx = self.ctx
    t = ctx.tile(tile_id)
    return t.get(ctx.backend, extent.offset_slice(ex, intersection))

  def _fetch_with_masks(self, region, splits, replicated, dst_rank, want):
    """fetch of a region that may hold never-written cells, across ranks (distarray.py:355-365 + tile.pyx:100-113:
    the reference ships whatever Tile.get returned, MaskedArrays included).  Every piece travels as a (values,
    written-cells) pair -- only its owner knows which of the three forms it has (plain, masked, never written),
    and the ranks must issue the same transfers -- and the destination gives back plain data if every cell turns out
    to be written, the masked form otherwise."""
    ctx = self.ctx
    be, world = ctx.backend, ctx.world
    pieces = []
    for ex, inter in splits:
      tid = self.tiles[ex]
      owner = ctx.rank_of(tid.worker)
      shape = _slices_shape(extent.offset_slice(ex, inter), ex.shape)
      if owner == world.rank:
        p = self._tile_piece(tid, ex, inter)
        if isinstance(p, tile.EmptyBlob):
          data, valid = be.zeros(shape, self.dtype), be.zeros(shape, np.uint8)
        elif isinstance(p, tile.MaskedBlob):
          data, valid = be.copy(p.data), be.copy(p.valid)
        else:
          data, valid = be.copy(p), be.zeros(shape, np.uint8)
          be.assign_box(valid, tuple(slice(0, n) for n in shape), 1)
      elif replicated or want:
        data, valid = be.empty(shape, self.dtype), be.empty(shape, np.uint8)
      else:
        data = valid = None
      if replicated:
        world.broadcast(data, owner)
        world.broadcast(valid, owner)
      elif owner != dst_rank:
        if owner == world.rank:
          world.exchange([(dst_rank, data), (dst_rank, valid)], [])
        elif want:
          world.exchange([], [(owner, data), (owner, valid)])
        else:
          world.exchange([], [])
      if want:
        pieces.append(tile.MaskedBlob(data.reshape(inter.shape), valid.reshape(inter.shape)))
    if not want:
      return Absent(region.shape, self.dtype)
    whole = pieces[0] if len(splits) == 1 else self._stitch(region, splits, pieces)
    full = tuple(slice(0, n) for n in whole.shape)
    if be.mask_all_set(whole.valid, full):
      return whole.data
    return whole

  def fetch(self, region):
    """distarray.py:294-367.  Inside a kernel the data is delivered to the rank
    owning the executing worker (other ranks get `Absent` and just serve their
    pieces); at driver level every rank receives it (replicated fetch)."""
    if type(region) is not extent.TileExtent:
      Assert.isinstance(region, extent.TileExtent)
    if region.array_shape != self.shape:
      Assert.eq(region.array_shape, self.shape)
    for l, n in zip(region.lr, self.shape):
      assert l <= n, 'Requested region is out of bounds: %s > %s' % (region, self.shape)
    ctx = self.ctx
    world = ctx.world
    # the common case first: exactly one tile, plain dense data written everywhere, held by the rank that asks
    # (any rank of a one-process world; the executing worker's own rank otherwise) -- the tile's tensor itself
    tid = self.tiles.get(region)
    if tid is not None:
      t = ctx._blobs.get(tid)
      if (t is not None and type(t.mask) is int and t.mask == tile.MASK_ALL_SET and t.type == tile.TYPE_DENSE and t.data is not None
              and t.shape and (world.size == 1 or (ctx.current_worker is not None and
                                                    ctx.current_worker % world.size == world.rank))):
        return t.data
    be = ctx.backend
    replicated = ctx.current_worker is None
    dst_rank = None if replicated else ctx.rank_of(ctx.current_worker)
    want = replicated or dst_rank == world.rank

    if region in self.tiles:  # exact tile (distarray.py:310-315)
      splits = [(region, region)]
    else:
      splits = list(extent.find_overlapping(self.tiles.keys(), region))

    if self.sparse:
      return self._fetch_sparse(region, splits, replicated, dst_rank, want)

    if not world.distributed:
      pieces = [self._tile_piece(self.tiles[ex], ex, inter) for ex, inter in splits]
      if len(splits) == 1:
        return pieces[0]
      return self._stitch(region, splits, pieces)

    if self._maybe_unwritten(splits):
      return self._fetch_with_masks(region, splits, replicated, dst_rank, want)

    # distributed: owners serve, the destination(s) assemble
    if len(splits) == 1:
      ex, inter = splits[0]
      tid = self.tiles[ex]
      owner = ctx.rank_of(tid.worker)
      shape = inter.shape if inter is not None else region.shape
      if replicated:
        pshape = _slices_shape(extent.offset_slice(ex, inter), ex.shape)
        if owner == world.rank:
          piece = self._tile_piece(tid, ex, inter)
          if isinstance(piece, tile.EmptyBlob):
            piece = be.zeros(pshape, self.dtype)  # never-written tile: serve zeros
          buf = be.contiguous(piece)
          world.broadcast(buf, owner)
          return piece
        buf = be.empty(pshape, self.dtype)
        world.broadcast(buf, owner)
        return buf
      if owner == dst_rank:
        if want:
          return self._tile_piece(tid, ex, inter)
        return Absent(shape, self.dtype)
      if owner == world.rank:
        world.exchange([(dst_rank, be.contiguous(self._tile_piece(tid, ex, inter)))], [])
        return Absent(shape, self.dtype)
      if want:
        buf = be.empty(_slices_shape(extent.offset_slice(ex, inter), ex.shape), self.dtype)
        world.exchange([], [(owner, buf)])
        return buf
      return Absent(shape, self.dtype)

    if replicated:
      return self._fetch_replicated(region, splits)

    # Several workers of one kernel asking for the WHOLE array (dot's outer
    # mapper fetches all of B for every tile, outer.py:21-29): gather it once
    # on every rank with one all-gather and serve the later requests locally,
    # instead of one gather-to-one round per worker.
    cache = ctx.fetch_cache
    if cache is not None and region.shape == self.shape and region.ul == (0,) * len(self.shape):
      key = (self.id, region.ul, region.lr)
      if key not in cache:
        cache[key] = self._fetch_replicated(region, splits)
      return cache[key] if want else Absent(region.shape, self.dtype)

    sends, recvs, pieces = [], [], []
    for ex, inter in splits:
      tid = self.tiles[ex]
      owner = ctx.rank_of(tid.worker)
      if want:
        if owner == world.rank:
          pieces.append(self._tile_piece(tid, ex, inter))
        else:
          buf = be.empty(inter.shape, self.dtype)
          recvs.append((owner, buf))
          pieces.append(buf)
      elif owner == world.rank:
        sends.append((dst_rank, be.contiguous(self._tile_piece(tid, ex, inter))))
    world.exchange(sends, recvs)
    if not want:
      return Absent(region.shape, self.dtype)
    return self._stitch(region, splits, pieces)

  def _fetch_sparse(self, region, splits, replicated, dst_rank, want):
    """Sparse counterpart of fetch (distarray.py:338-353: the pieces are placed into one sparse matrix).
    Pieces owned by the destination stay where they are; pieces that have to cross ranks travel device to
    device in ONE grouped exchange (_ship_sparse)."""
    ctx = self.ctx
    be = ctx.backend
    world = ctx.world
    owners = [ctx.rank_of(self.tiles[ex].worker) for ex, _ in splits]

    def local_piece(i):
      ex, inter = splits[i]
      return self._tile_piece(self.tiles[ex], ex, inter)

    remote = {}
    if world.distributed:
      crossing = [i for i in range(len(splits)) if replicated or owners[i] != dst_rank]
      if crossing:   # known identically on every rank
        everyone = list(range(world.size))
        remote = _ship_sparse(ctx, {i: (everyone if replicated else [dst_rank], local_piece(i))
                                    for i in crossing if owners[i] == world.rank})
    if not want:
      return Absent(region.shape, self.dtype)
    pieces = []
    for i, (ex, inter) in enumerate(splits):
      if owners[i] == world.rank or not world.distributed:
        p = local_piece(i)
      else:
        p = be.sparse_blob(remote[i], self.dtype)
      pieces.append(p)
    if len(splits) == 1:
      return pieces[0]
    return be.sparse_paste(region.shape, self.dtype,
                           [(tuple(a - b for a, b in zip(inter.ul, region.ul)), p)
                            for (ex, inter), p in zip(splits, pieces)])

  def _stitch(self, region, splits, pieces):
    """distarray.py:355-365: allocate the region and paste the pieces."""
    be = self.ctx.backend
    if all(isinstance(p, tile.EmptyBlob) for p in pieces):
      return tile.EmptyBlob(region.shape, self.dtype)
    # pieces with unwritten cells (tile.MaskedBlob) and never-written pieces (tile.EmptyBlob: fully masked zeros)
    # make the stitched region masked as well
    partial = any(isinstance(p, (tile.MaskedBlob, tile.EmptyBlob)) for p in pieces)
    tgt = be.zeros(region.shape, self.dtype) if partial else be.empty(region.shape, self.dtype)
    valid = be.zeros(region.shape, np.uint8) if partial else None
    for (ex, inter), piece in zip(splits, pieces):
      dst_slice = extent.offset_slice(region, inter)
      if not extent.all_nonzero_shape(piece.shape) or isinstance(piece, tile.EmptyBlob):
        continue
      if isinstance(piece, tile.MaskedBlob):
        be.paste(tgt, dst_slice, piece.data)
        be.paste(valid, dst_slice, piece.valid)
      else:
        be.paste(tgt, dst_slice, piece)
        if valid is not None:
          be.assign_box(valid, dst_slice, 1)
    return tgt if valid is None else tile.MaskedBlob(tgt, valid)

  def _fetch_replicated(self, region, splits):
    """Every rank assembles `region` (glom and driver-level fetches): one
    all-gather when the pieces are the equal-size, one-per-rank, rank-ordered
    row blocks of the default tiling; per-piece broadcasts otherwise."""
    ctx = self.ctx
    be = ctx.backend
    world = ctx.world
    owners = [ctx.rank_of(self.tiles[ex].worker) for ex, _ in splits]
    order = sorted(range(len(splits)), key=lambda i: splits[i][1].ul)
    regular = (len(splits) == world.size and sorted(owners) == list(range(world.size)) and
               len(set(inter.shape for _, inter in splits)) == 1)
    if regular:
      # rank-ordered contiguous row blocks of `region`?
      for pos, i in enumerate(order):
        ex, inter = splits[i]
        if owners[i] != pos or inter.ul[1:] != region.ul[1:] or inter.lr[1:] != region.lr[1:]:
          regular = False
          break
    if regular:
      mine = [i for i in range(len(splits)) if owners[i] == world.rank][0]
      ex, inter = splits[mine]
      piece = be.contiguous(self._tile_piece(self.tiles[ex], ex, inter))
      tgt = be.empty(region.shape, self.dtype)
      world.all_gather_into(tgt, piece)
      return tgt
    tgt = be.empty(region.shape, self.dtype)
    for (ex, inter), owner in zip(splits, owners):
      if owner == world.rank:
        buf = be.contiguous(self._tile_piece(self.tiles[ex], ex, inter))
      else:
        buf = be.empty(inter.shape, self.dtype)
      world.broadcast(buf, owner)
      if extent.all_nonzero_shape(buf.shape):
        be.paste(tgt, extent.offset_slice(region, inter), buf)
    return tgt

  def fetch_whole_chunked(self, chunk_cols):
    """Replicate a row-tiled 2-D array on every rank as COLUMN chunks gathered by independent,
    asynchronous all-gathers (one per chunk), so that a consumer can start on chunk 0 while the
    later chunks are still on the wire (dot's outer path: the gather of B hides behind the GEMM).
    Returns ChunkedWhole or None when the pattern does not apply (the caller then uses fetch()).
    Collective: every rank must call it at the same point of the tile walk."""
    ctx = self.ctx
    be = ctx.backend
    world = ctx.world
    if not world.distributed or ctx.num_workers != world.size or len(self.shape) != 2:
      return None
    rows, cols = self.shape
    if chunk_cols <= 0 or cols % chunk_cols != 0 or cols // chunk_cols < 2:
      return None
    tiles = sorted(self.tiles.items(), key=lambda kv: kv[0].ul)
    if len(tiles) != world.size or rows % world.size != 0:
      return None
    step = rows // world.size
    for r, (ex, tid) in enumerate(tiles):
      if (ctx.rank_of(tid.worker) != r or ex.ul != (r * step, 0) or ex.lr != ((r + 1) * step, cols)):
        return None
    cache = ctx.fetch_cache
    key = (self.id, 'chunked', chunk_cols)
    if cache is not None and key in cache:
      return cache[key]
    ex, tid = tiles[world.rank]
    mine = self._tile_piece(tid, ex, ex)
    # only the owner knows whether its tile holds plain data: agreed over the control plane, so that either every
    # rank enters the gathers below or none does
    if not all(world.all_gather_object(not isinstance(mine, (tile.EmptyBlob, tile.MaskedBlob)))):
      return None
    chunks = []
    for c0 in range(0, cols, chunk_cols):
      piece = be.copy(mine[:, c0:c0 + chunk_cols])          # contiguous (step, chunk_cols) block of my rows
      out = be.empty((rows, chunk_cols), self.dtype)         # rank-ordered row blocks == B[:, c0:c1], ld = chunk_cols
      work = world.all_gather_into_async(out, piece)
      chunks.append((c0, c0 + chunk_cols, out, work, piece))
    whole = ChunkedWhole(self.shape, self.dtype, chunks)
    if cache is not None:
      cache[key] = whole
    return whole

  def update_slice(self, slc, data):
    return self.update(extent.from_slice(slc, self.shape), data)

  def _update_splits(self, region):
    """distarray.py:378-408: [(tile_id, src_slice, dst_slice)] sorted by the
    first-dimension start of the source slice."""
    if region in self.tiles:
      return [(self.tiles[region], extent.offset_slice(region, region), extent.offset_slice(region, region))]
    slices = []
    if region.shape == self.shape:
      for ex, tile_id in self.tiles.items():
        slices.append((tile_id, ex.to_slice(), extent.offset_slice(ex, ex)))
    else:
      for dst_extent, intersection in extent.find_overlapping(self.tiles, region):
        tile_id = self.tiles[dst_extent]
        src_slice = extent.offset_slice(region, intersection)
        dst_slice = extent.offset_slice(dst_extent, intersection)
        shape = [s.stop - s.start for s in dst_slice]
        if extent.all_nonzero_shape(shape):
          slices.append((tile_id, src_slice, dst_slice))
    if slices and len(slices[0][1]) > 0:
      slices.sort(key=lambda x: x[1][0].start)
    return slices

  def update(self, region, data, wait=True, owned=False):
    """distarray.py:372-422.  `data` is a backend tensor on the executing rank
    (`Absent` elsewhere).  Inside a kernel the update joins the kernel's batch;
    at driver level it is applied immediately."""
    if type(region) is not extent.TileExtent:
      Assert.isinstance(region, extent.TileExtent)
    if region.shape != tuple(data.shape):
      Assert.eq(region.shape, tuple(data.shape), 'Size of extent does not match size of data')
    ctx = self.ctx
    if tile.is_sparse_blob(data) and not ctx.executing:
      data = Absent(region.shape, self.dtype)    # a mapper that builds its block on every rank: only the executing rank's counts
    if tile.is_sparse_blob(data):
      if self.sparse:
        data = ctx.backend.sparse_blob(data, self.dtype)     # upload what a mapper yielded
      else:
        # A sparse block for a dense target travels and merges as a dense block.  (The reference adds
        # it cell by cell with REDUCE_ADD whatever the target's reducer is, tile.pyx:229-233; the two
        # agree for the add reducer and for non-overlapping blocks.)
        data = ctx.backend.sparse_to_dense(ctx.backend.sparse_blob(data, self.dtype))
        owned = True
    elif self.sparse and not isinstance(data, Absent):
      # A dense block for a sparse target (tile.pyx:284-297: the reference turns the tile into LIL and assigns the
      # slice, "this is SLOW"): its non-zero cells travel and merge as a sparse block.  That is the same array for
      # the two reducers a sparse tile has -- None replaces the cells of the box (zeros of the block clear what was
      # there: the box's old entries are dropped) and np.add adds, to which zeros contribute nothing.
      data = ctx.backend.dense_to_sparse(data, self.dtype)
      owned = True
    if ctx.pending is not None:
      ctx.pending.add(self, region, data, owned)
      return None
    batch = UpdateBatch(ctx)
    batch.add(self, region, data, owned)
    batch.flush()
    return None


class KernelResults(collections.OrderedDict):
  """{tile id: what its mapper produced}; `meta`: the (dtype, is_sparse) every mapper call derived for the tiles it
  made -- the same on every rank -- or None."""
  meta = None


def kernel_order(array, tile_ids, ctx):
  """The order in which the mappers of one kernel run, as the reference's workers run them: the request goes to EVERY
  worker (blob_ctx.py:270-271 `_send_all`) and each collects the tiles it holds in list order, sorts them by the size
  of their data -- a stable sort, ascending -- and POPS them from the end (worker.py:246-256): largest tile first,
  tiles of one size in REVERSE list order.  Concurrently there; worker after worker, lowest first, is the
  linearisation its recorded outputs were produced with, and it shows wherever a target keeps the last write (updates
  without a reducer: the k-means drivers' count / sum targets) or partial results are added in floating point.
  Size: what np.size gives the reference -- stored values of a sparse tile, 1 for a tile nothing was written to --
  when every tile is in this process; across ranks, where all must walk ONE order, the tile's extent."""
  if len(tile_ids) < 2:
    return list(tile_ids)
  single = ctx.world.size == 1
  blobs = ctx._blobs
  by_worker = {}
  for tid in tile_ids:
    by_worker.setdefault(tid.worker, []).append(tid)
  order = []
  for worker in sorted(by_worker):
    mine = by_worker[worker]
    if len(mine) > 1:
      sizes = {}
      for tid in mine:
        t = blobs.get(tid) if single else None
        if t is None:
          try:
            ex = array.extent_for_blob(tid)
          except (KeyError, AttributeError):
            ex = None
          sizes[tid] = int(np.prod(ex.shape, dtype=np.int64)) if ex is not None else 0
        elif t.data is None:
          sizes[tid] = 1
        elif tile.is_sparse_blob(t.data):
          sizes[tid] = int(getattr(t.data, 'nnz', 0))
        else:
          sizes[tid] = int(np.prod(t.shape, dtype=np.int64))
      mine = sorted(mine, key=sizes.__getitem__)
      mine.reverse()
    order.extend(mine)
  return order


def run_kernel(array, tile_ids, mapper_fn, kw):
  """blob_ctx.map + Worker._run_kernel: call the mapper for every tile (in a
  deterministic order, on every rank) and join the updates it issued."""
  ctx = context.get()
  kw = dict(kw)
  results = KernelResults()
  metas = []
  outer = ctx.pending
  outer_cache = ctx.fetch_cache
  batch = UpdateBatch(ctx)
  ctx.pending = batch
  ctx.fetch_cache = {}
  outer_worker = ctx.current_worker
  blobs, invoke = ctx._blobs, array._invoke_mapper
  tile_ids = kernel_order(array, tile_ids, ctx)
  try:
    for tile_id in tile_ids:
      ctx.current_worker = tile_id.worker          # (Context.on_worker, without the context manager)
      blob = blobs.get(tile_id)
      if blob is None and ctx.is_local(tile_id):
        raise KeyError('tile %r of a worker of this rank is not in the tile store (dropped by a failure that the '
                       'array has not recorded as a bad tile?)' % (tile_id,))
      res = invoke(tile_id, blob, mapper_fn, kw)
      if res is None:
        continue
      results[tile_id] = res.result
      metas.append(getattr(res, 'meta', None))
      # a mapper returning an existing tile id shares it (worker.py:285-295)
      if res.result:
        for ex, tid in res.result:
          if tid == tile_id:
            ctx.incref(tid)
  finally:
    ctx.current_worker = outer_worker
    ctx.pending = outer
    ctx.fetch_cache = outer_cache
  if batch.items:
    batch.flush()
  if metas and metas[0] is not None and all(m == metas[0] for m in metas):
    results.meta = metas[0]
  return results


def _invoke_default(self, tile_id, blob, mapper_fn, kw):
  return _tile_mapper(tile_id, blob, array=self, user_fn=mapper_fn, **kw)


DistArrayImpl._invoke_mapper = _invoke_default


def create(shape, dtype=float, sharder=None, reducer=None, tile_hint=None, sparse=False):
  """A new, empty array: cut into tiles (compute_extents), each tile given to a worker by the tile-assignment
  policy (array/placement.py; reference distarray.py:425-487)."""
  ctx = context.get()
  dtype = np.dtype(dtype)
  shape = tuple(int(s) for s in shape)
  if sparse:
    Assert.eq(len(shape), 2, 'sparse arrays are two-dimensional')
  ttype = tile.TYPE_SPARSE if sparse else tile.TYPE_DENSE
  cut = list(compute_extents(shape, tile_hint, ctx.num_workers).items())
  tiles = collections.OrderedDict()
  for (ex, _), worker in zip(cut, placement.place(cut, ctx)):
    t = tile.from_shape(ex.shape, dtype, ttype) if ctx.is_local_worker(worker) else None
    tiles[ex] = ctx.create(t, hint=worker)
  arr = DistArrayImpl(shape=shape, dtype=dtype, tiles=tiles, reducer_fn=reducer, sparse=bool(sparse))
  if not sparse:
    arr.written = set()        # nothing written yet (a sparse tile has no mask: tile.pyx:129-132)
  return arr


def from_table(extents, meta=None):
  """distarray.py:519-550.  meta: (dtype, is_sparse) of the tiles when the caller knows it on every rank; otherwise
  the owner of the first tile is asked (the reference's tile_op RPC, distarray.py:542)."""
  # (keys of a dict: no duplicates by construction -- the reference's check, distarray.py:528, guards its list form)
  if not extents:
    shape = tuple()
  else:
    shape = extent.find_shape(list(extents.keys()))
  if len(extents) > 0 and meta is not None:
    dtype, sparse = np.dtype(meta[0]), bool(meta[1])
  elif len(extents) > 0:
    key, tile_id = next(iter(extents.items()))
    dtype, sparse = context.get().tile_meta(tile_id)
  else:
    dtype = np.dtype(float)
    sparse = False
  return DistArrayImpl(shape=shape, dtype=dtype, tiles=extents, reducer_fn=None, sparse=sparse)


class LocalWrapper(DistArray):
  """distarray.py:553-602: the DistArray interface for driver-local data
  (NumPy arrays and scalars, replicated on every rank)."""

  def __init__(self, data):
    # Python scalars stay "weak" (see SURVEY 8c, NEP-50 note): fp32 + 1 is fp32
    self._scalar = data if isinstance(data, (bool, int, float)) and not isinstance(data, np.generic) else None
    self._data = np.asarray(data)
    self.sparse = False
    self.bad_tiles = []
    self._whole = None
    if not isinstance(data, (np.ndarray, int, float, bool, np.generic)):
      Assert.isinstance(data, (np.ndarray, int, float, bool, np.generic))
    self._dev = None

  @property
  def _ex(self):
    """The one pseudo-tile: all of the data."""
    if self._whole is None:
      self._whole = extent.from_slice(np.index_exp[:], self.shape)
    return self._whole

  dtype = property(lambda self: self._data.dtype)
  shape = property(lambda self: self._data.shape)
  is_weak_scalar = property(lambda self: self._scalar is not None)
  tiles = property(lambda self: {self._ex: TileId(-1, 0)})     # one pseudo-tile that lives on the driver

  def extent_for_blob(self, tile_id):
    return self._ex

  def fetch(self, ex):
    if self._data.ndim == 0:
      return self._scalar if self._scalar is not None else self._data
    return self._data[ex.to_slice()]

  def glom(self):
    return self._data

  def foreach_tile(self, mapper_fn, kw=None):
    if kw is None:
      kw = {}
    ctx = context.get()
    map_result = mapper_fn(self._ex, **kw)
    result = map_result.result
    assert len(result) == 1
    result_ex, tile_id = result[0]
    Assert.isinstance(tile_id, TileId)
    return from_table({result_ex if result_ex is not None else extent.create((), (), ()): tile_id})

  def map_to_array(self, mapper_fn, kw=None):
    return self.foreach_tile(mapper_fn=mapper_fn, kw=kw)


def as_array(data):
  """distarray.py:605-617."""
  if isinstance(data, DistArray):
    return data
  return LocalWrapper(data)


def largest_value(vals):
  """distarray.py:636-642."""
  return max(vals, key=lambda v: v.real_size())
