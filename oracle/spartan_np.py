"""TEST INFRASTRUCTURE -- the parity oracle.  Not part of the product path.

An independent, eager NumPy restatement of the reference's per-worker tile
path (spartan.expr MapExpr / ReduceExpr / dot over spartan.array tiles), with N
simulated workers executed serially in one process.  It shares no code with
spartan_amd/.  Every function cites the reference lines it follows.

Pinned: tests/test_oracle.py checks this module against the golden vectors in
tests/golden/ that were produced by RUNNING THE REFERENCE itself
(tests/golden/make_golden.py) -- extents and tilings, the Tile.merge truth
table, and 74 map/reduce/argmax/dot programs at 1, 3, 4 and 8 workers.

Users: tests/, __graft_entry__.smoke() (checker for the HIP result) and
bench.py's cpu_baseline leg (kind "port").  Nothing under spartan_amd/ imports
it.

Deliberate, documented choices:
  * Python scalars are weak operands (fp32 + 1 stays fp32), the NumPy-1.x
    behaviour the reference relied on;
  * reducer application order is the deterministic tile order (the real system
    applies updates in arrival order).
"""
import itertools
import math

import numpy as np


# ----------------------------------------------------------------- extents
class Extent(object):
  """extent.pyx:23-136: half-open box [ul, lr) of an array of `array_shape`."""

  def __init__(self, ul, lr, array_shape):
    self.ul = tuple(int(v) for v in ul)
    self.lr = tuple(int(v) for v in lr)
    self.array_shape = None if array_shape is None else tuple(int(v) for v in array_shape)

  @property
  def shape(self):  # extent.pyx:66-72
    return tuple((l - u) or 1 for u, l in zip(self.ul, self.lr))

  @property
  def size(self):
    return int(np.prod(self.shape)) if self.shape else 1

  def to_slice(self):
    return tuple(slice(u, l) for u, l in zip(self.ul, self.lr))

  def key(self):
    return (self.ul, self.lr)

  def __repr__(self):
    return 'ex(' + ','.join('%d:%d' % p for p in zip(self.ul, self.lr)) + ')'


def create(ul, lr, array_shape):
  """extent.create under the name user tile functions call it by."""
  return ex_create(ul, lr, array_shape)


def ex_create(ul, lr, array_shape):
  """extent.pyx:141-182: None when any ul >= lr."""
  for u, l in zip(ul, lr):
    if u >= l:
      return None
  return Extent(ul, lr, array_shape)


def ex_intersection(a, b):
  """extent.pyx:367-387."""
  ul, lr = [], []
  for i in range(len(a.ul)):
    if b.lr[i] < a.ul[i] or a.lr[i] < b.ul[i]:
      return None
    ul.append(max(a.ul[i], b.ul[i]))
    lr.append(min(a.lr[i], b.lr[i]))
  return ex_create(ul, lr, a.array_shape)


def ex_offset_slice(base, other):
  """extent.pyx:316-324."""
  return tuple(slice(o_u - b_u, o_l - b_u) for b_u, o_u, o_l in zip(base.ul, other.ul, other.lr))


def ex_drop_axis(ex, axis):
  """extent.pyx:411-432 (index_for_reduction)."""
  if axis is None:
    return Extent((), (), ())
  if axis < 0:
    axis += len(ex.ul)
  keep = [i for i in range(len(ex.ul)) if i != axis]
  return ex_create([ex.ul[i] for i in keep], [ex.lr[i] for i in keep], [ex.array_shape[i] for i in keep])


def ravelled_pos(idx, array_shape):
  """extent.pyx:211-219."""
  pos, mul = 0, 1
  for i in range(len(array_shape) - 1, -1, -1):
    pos += mul * int(idx[i])
    mul *= int(array_shape[i])
  return pos


def unravelled_pos(idx, array_shape):
  """extent.pyx:195-209."""
  out = []
  for dim in reversed(array_shape):
    out.append(idx % dim)
    idx //= dim
  return tuple(reversed(out))


def divup(a, b):
  """util.py:404-408."""
  return int(math.ceil(float(a) / b))


def change_partition_axis(ex, axis):
  """extent.pyx:501-570 (1-D re-partition cases used by map2 / outer)."""
  if axis < 0:
    axis += len(ex.array_shape)
  if len(ex.shape) == 1:
    return Extent((0,), ex.array_shape, ex.array_shape) if axis == 1 else ex
  old_axes = [i for i in range(len(ex.shape)) if ex.shape[i] != ex.array_shape[i]]
  if len(old_axes) > 1:
    blk = (ex.ul[0] // ex.shape[0]) * divup(ex.array_shape[1], ex.shape[1]) + ex.ul[1] // ex.shape[1]
    ul, lr = [0, 0], list(ex.array_shape)
    ul[axis], lr[axis] = blk, blk + 1
    return ex_create(ul, lr, ex.array_shape)
  if not old_axes or old_axes[0] == axis:
    return ex
  old = old_axes[0]
  ul, lr = list(ex.ul), list(ex.lr)
  ul[axis] = divup(ul[old] * ex.array_shape[axis], ex.array_shape[old])
  ul[old] = 0
  lr[axis] = divup(lr[old] * ex.array_shape[axis], ex.array_shape[old])
  lr[old] = ex.array_shape[old]
  return ex_create(ul, lr, ex.array_shape)


# ------------------------------------------------------------------ tiling
def good_tile_shape(shape, num_shards):
  """distarray.py:26-48."""
  tile_size = int(np.prod(shape)) // num_shards
  tile_shape = [1] * len(shape)
  idx = len(shape) - 1
  while tile_size > 1:
    tile_shape[idx] = min(shape[idx], tile_size)
    tile_size //= shape[idx]
    idx -= 1
  return tile_shape


def compute_extents(shape, tile_hint, num_shards):
  """distarray.py:51-110: [(extent, shard)] in product order, round-robin shards."""
  if len(shape) == 0:
    return [(Extent((), (), ()), 0)]
  if tile_hint is None:
    tile_hint = good_tile_shape(shape, num_shards)
  splits = []
  for dim in range(len(shape)):
    splits.append([(i, min(shape[dim], i + tile_hint[dim])) for i in range(0, shape[dim], tile_hint[dim])])
  out = []
  for idx, slc in enumerate(itertools.product(*splits)):
    ul, lr = zip(*slc)
    out.append((Extent(ul, lr, shape), idx % num_shards))
  return out


# -------------------------------------------------------------------- tiles
class Tile(object):
  """tile.pyx:24-62 (dense): data + bool mask; an empty tile has neither."""

  def __init__(self, shape, dtype):
    self.shape = tuple(shape)
    self.dtype = np.dtype(dtype)
    self.data = None
    self.mask = None

  def get(self, subslice):
    """tile.pyx:64-113."""
    if self.data is None:
      shp = self.shape if subslice is None else tuple(
          len(range(*s.indices(n))) for s, n in zip(subslice, self.shape))
      return np.zeros(shp, self.dtype)  # reference: uninitialised memory
    if len(self.shape) == 0:
      return self.data
    if not np.all(self.mask[subslice]):
      raise ValueError('masked read')
    return self.data[subslice]


def merge(t, subslice, update, reducer):
  """tile.pyx:200-297, dense -> dense branch."""
  update = np.asarray(update)
  if len(t.shape) == 0:                       # tile.pyx:212-217
    if t.data is None or reducer is None:
      t.data = update.astype(t.dtype)
    else:
      t.data = np.asarray(reducer(t.data, update)).astype(t.dtype)
    return t
  if t.data is None:                           # _initialize, tile.pyx:115-127
    t.data = np.zeros(t.shape, t.dtype)
    t.mask = np.zeros(t.shape, bool)
  if t.data.shape == update.shape:             # tile.pyx:261-268
    if reducer is not None and t.mask[np.unravel_index(0, t.shape)]:
      t.data = np.asarray(reducer(t.data, update)).astype(t.dtype)
    else:
      t.data = update.astype(t.dtype)
    t.mask = np.ones(t.shape, bool)
    return t
  replaced = ~t.mask[subslice]                 # tile.pyx:270-283
  updated = t.mask[subslice]
  region = t.data[subslice]
  if np.any(replaced):
    region[replaced] = update[replaced]
  if np.any(updated):
    region[updated] = reducer(region[updated], update[updated]) if reducer is not None else update[updated]
  t.mask[subslice] = True
  return t


# --------------------------------------------------------------- dist arrays
class DistArray(object):
  """distarray.py:223-422: extent -> (worker, Tile)."""

  def __init__(self, shape, dtype, reducer=None, tile_hint=None, num_workers=1):
    self.shape = tuple(int(s) for s in shape)
    self.dtype = np.dtype(dtype)
    self.reducer = reducer
    self.num_workers = num_workers
    self.tiles = []
    for ex, shard in compute_extents(self.shape, tile_hint, num_workers):
      self.tiles.append((ex, shard % num_workers, Tile(ex.shape, self.dtype)))
    self._by_key = dict((ex.key(), (ex, w, t)) for ex, w, t in self.tiles)

  def real_size(self):
    return int(np.prod(self.shape)) if self.shape else 1

  def fetch(self, region):
    """distarray.py:294-367."""
    hit = self._by_key.get(region.key())
    if hit is not None:
      return hit[2].get(ex_offset_slice(region, region))
    pieces = []
    for ex, _, t in self.tiles:
      inter = ex_intersection(ex, region)
      if inter is not None:
        pieces.append((inter, t.get(ex_offset_slice(ex, inter))))
    if len(pieces) == 1:
      return pieces[0][1]
    out = np.ndarray(region.shape, self.dtype)
    for inter, data in pieces:
      if all(s != 0 for s in data.shape):
        out[ex_offset_slice(region, inter)] = data
    return out

  def update(self, region, data):
    """distarray.py:372-422."""
    data = np.asarray(data)
    assert region.shape == data.shape, (region.shape, data.shape)
    hit = self._by_key.get(region.key())
    if hit is not None:
      merge(hit[2], ex_offset_slice(region, region), data, self.reducer)
      return
    slices = []
    if region.shape == self.shape:
      for ex, _, t in self.tiles:
        slices.append((t, ex.to_slice(), ex_offset_slice(ex, ex)))
    else:
      for ex, _, t in self.tiles:
        inter = ex_intersection(ex, region)
        if inter is None:
          continue
        dst = ex_offset_slice(ex, inter)
        if all((s.stop - s.start) != 0 for s in dst):
          slices.append((t, ex_offset_slice(region, inter), dst))
    if slices and len(slices[0][1]):
      slices.sort(key=lambda x: x[1][0].start)
    for t, src, dst in slices:                 # sparse.pyx:297-301 multiple_slice, dense branch
      merge(t, dst, data[src], self.reducer)

  def glom(self):
    if len(self.shape) == 0:
      return np.asarray(self.tiles[0][2].data)
    return np.asarray(self.fetch(Extent([0] * len(self.shape), self.shape, self.shape)))


class Scalar(object):
  """LocalWrapper of a Python scalar (distarray.py:553-602), kept weak."""

  def __init__(self, v):
    self.v = v
    self.shape = ()

  def real_size(self):
    return 1


class Cluster(object):
  """A simulated cluster of `num_workers` workers (default tile assignment
  round_robin, distarray.py:441-445)."""

  def __init__(self, num_workers=1):
    self.n = num_workers

  @staticmethod
  def kernel_order(array):
    """The tiles of `array` in the order their mappers run: a kernel request goes to EVERY worker (blob_ctx.py:270-271)
    and each collects the tiles it holds in list order, sorts them by np.size of their data (stable, ascending) and
    pops them from the END (worker.py:246-256) -- largest first, tiles of one size in reverse list order.
    Concurrently in the reference; one worker after the other, lowest first, in the serial runs its recorded outputs
    come from.  Only targets that keep the LAST write (updates without a reducer) or add floats can tell."""
    order = []
    for worker in sorted(set(entry[1] for entry in array.tiles)):
      mine = [entry for entry in array.tiles if entry[1] == worker]
      mine.sort(key=lambda entry: 1 if entry[2].data is None else int(np.size(entry[2].data)))
      order.extend(reversed(mine))
    return order

  # -- creation (creation.py) --------------------------------------------------
  def empty(self, shape, dtype=np.float32, reducer=None, tile_hint=None):
    return DistArray(shape, dtype, reducer, tile_hint, self.n)

  def from_numpy(self, a, tile_hint=None):
    a = np.asarray(a)
    d = self.empty(a.shape, a.dtype, None, tile_hint)
    for ex, _, t in d.tiles:
      merge(t, None, a[ex.to_slice()].reshape(ex.shape).copy(), None)
    return d

  def ones(self, shape, dtype=np.float32):
    return self.map(lambda t: np.ones(t.shape, t.dtype), self.empty(shape, dtype))     # creation.py:92-106

  def zeros(self, shape, dtype=np.float32):
    return self.map(lambda t: np.zeros(t.shape, t.dtype), self.empty(shape, dtype))    # creation.py:67-81

  def arange(self, shape, start=0, step=1, dtype=float):
    def fn(t, ex):                                                                       # creation.py:134-141
      pos = ravelled_pos(ex.ul, ex.array_shape)
      s0 = pos * step + start
      return np.arange(s0, np.prod(t.shape) * step + s0, step, dtype=dtype).reshape(t.shape)
    return self.map_with_location(fn, self.empty(shape, dtype))

  # -- map (map.py:33-88, broadcast.py) ---------------------------------------------
  def _broadcast_fetch(self, child, ex, out_ndim):
    if isinstance(child, Scalar):
      return child.v
    if child.shape == ex.array_shape:
      return child.fetch(ex)
    # Broadcast._base_ex (broadcast.py:72-91) + fetch_base_tile
    pad = out_ndim - len(child.shape)
    ul, lr = [], []
    for i, size in enumerate(child.shape):
      if size == 1:
        ul.append(0)
        lr.append(1)
      else:
        ul.append(ex.ul[i + pad])
        lr.append(ex.lr[i + pad])
    return child.fetch(Extent(ul, lr, child.shape))

  def _map(self, fn, children, with_location):
    children = [c if isinstance(c, (DistArray, Scalar)) else Scalar(c) for c in children]
    arrays = [c for c in children if isinstance(c, DistArray)]
    shape = np.broadcast_shapes(*[c.shape for c in arrays])
    # the largest (non-broadcast) input drives the tiling (map.py:155-169, broadcast.py:54-59)
    largest = max(arrays, key=lambda c: c.real_size() - (0 if c.shape == tuple(shape) else 1))
    if largest.shape != tuple(shape):
      raise NotImplementedError('map driven by a broadcast array')
    out_tiles = []
    for ex, w, _ in largest.tiles:
      vals = [self._broadcast_fetch(c, ex, len(shape)) for c in children]
      with np.errstate(all='ignore'):
        res = fn(*vals, ex) if with_location else fn(*vals)
      res = np.asarray(res)
      assert res.shape == ex.shape, (res.shape, ex.shape)
      out_tiles.append((ex, w, res))
    out = DistArray.__new__(DistArray)
    out.shape = tuple(shape)
    out.dtype = out_tiles[0][2].dtype
    out.reducer = None
    out.num_workers = self.n
    out.tiles = []
    for ex, w, res in out_tiles:
      t = Tile(ex.shape, res.dtype)
      t.data = res
      t.mask = np.ones(ex.shape, bool)           # tile.from_data, tile.pyx:145-159
      out.tiles.append((ex, w, t))
    out._by_key = dict((ex.key(), (ex, w, t)) for ex, w, t in out.tiles)
    return out

  def map(self, fn, *children):
    return self._map(fn, children, False)

  def map_with_location(self, fn, *children):
    return self._map(fn, children, True)

  # -- reduce (reduce.py:21-127) -----------------------------------------------------
  def reduce(self, x, axis, dtype, local_fn, accumulate):
    if axis is None:
      shape = ()
    else:
      shape = list(x.shape)
      del shape[axis]
    out = self.empty(shape, dtype, accumulate)
    for ex, _, t in self.kernel_order(x):
      with np.errstate(all='ignore'):
        local = local_fn(ex, x.fetch(ex), axis)
      dst = ex_drop_axis(ex, axis)
      local = np.asarray(local).reshape(dst.shape)
      assert local.size == dst.size
      out.update(dst, local)
    return out

  def sum(self, x, axis=None):      # mathematics.py:126-143
    return self.reduce(x, axis, x.dtype, lambda ex, d, a: d.sum(a), np.add)

  def max(self, x, axis=None):      # statistics.py:26-42
    return self.reduce(x, axis, x.dtype, lambda ex, d, a: d.max(a), np.maximum)

  def min(self, x, axis=None):      # statistics.py:45-61
    return self.reduce(x, axis, x.dtype, lambda ex, d, a: d.min(a), np.minimum)

  def _arg(self, x, axis, extreme):
    """sorting.py:67-123: the reference's three passes -- extreme-reduce,
    _arg_mapper candidate-index map (sentinel prod(array_shape)), min-reduce."""
    best = extreme(x, axis)
    if axis is not None:
      kshape = list(x.shape)
      kshape[axis] = 1
      best = self.from_numpy(best.glom().reshape(kshape))   # .reshape(...) of the reduce result
    else:
      best = Scalar(best.glom()[()])

    def arg_mapper(a, b, ex):                                # sorting.py:67-85
      c = np.zeros(a.shape)
      c[a == b] = 1
      max_index = np.argmax(c, axis)
      if axis is not None:
        shp = list(a.shape)
        shp[axis] = 1
        gidx = max_index.reshape(tuple(shp)) + ex.ul[axis]
      else:
        local = unravelled_pos(int(max_index), ex.shape)
        gidx = ravelled_pos(np.asarray(ex.ul) + local, ex.shape)   # NB: tile shape (sorting.py:76-81)
      out = np.zeros(a.shape, dtype=np.int64) + gidx
      out[a != b] = np.prod(np.asarray(ex.array_shape))
      return out

    cand = self.map_with_location(arg_mapper, x, best)
    return self.min(cand, axis)

  def argmax(self, x, axis=None):
    return self._arg(x, axis, self.max)

  def argmin(self, x, axis=None):
    return self._arg(x, axis, self.min)

  # -- map2 (map.py:243-375) ----------------------------------------------------------------
  def map2(self, arrays, axes, fn, shape, reducer=None, fn_kw=None):
    """join_mapper, map.py:243-286: every tile of arrays[0] is re-read as the slab that cuts axes[0]
    (change_partition_axis), the other arrays give the slab with the same range on their join axis; what
    fn(extents, slabs) yields is pushed into a target of `shape` with arrays[0]'s dtype (map.py:316-334)."""
    target = self.empty(shape, arrays[0].dtype, reducer)
    for ex, _, t in self.kernel_order(arrays[0]):
      if not axes:
        extents, slabs = ex, [a.fetch(ex) for a in arrays]
      else:
        lead = change_partition_axis(ex, axes[0])
        if lead is None:
          continue
        extents = [lead]
        for arr, axis in zip(arrays[1:], axes[1:]):
          ul, lr = [0] * len(arr.shape), list(arr.shape)
          ul[axis], lr[axis] = lead.ul[axes[0]], lead.lr[axes[0]]
          extents.append(ex_create(ul, lr, arr.shape))
        slabs = [a.fetch(e) for a, e in zip(arrays, extents)]
      for where, data in fn(extents, slabs, **(fn_kw or {})) or ():
        target.update(where, data)
    return target

  # -- outer (outer.py:12-99) -----------------------------------------------------------------
  def outer(self, arrays, axes, fn, shape, reducer=None, fn_kw=None, tile_hint=None, dtype=None):
    """outer_mapper, outer.py:12-59: every tile of arrays[0] (re-read on axes[0]) meets the WHOLE of arrays[1]
    (axes[1] None) or each of its tiles in table order, re-read on axes[1] (one call per pair); what
    fn(ex_a, tile_a, ex_b, tile_b) yields is pushed into a target of arrays[0]'s dtype (outer.py:91-97)."""
    a, b = arrays
    target = self.empty(shape, a.dtype if dtype is None else dtype, reducer, tile_hint)
    for ex, _, t in self.kernel_order(a):
      first = change_partition_axis(ex, axes[0])
      tile_a = a.fetch(first)
      if axes[1] is None:
        whole = ex_create([0] * len(b.shape), b.shape, b.shape)
        for where, data in fn(first, tile_a, whole, b.fetch(whole), **(fn_kw or {})) or ():
          target.update(where, data)
        continue
      done = set()
      for bex, _, _t in b.tiles:
        other = change_partition_axis(bex, axes[1])
        if other is None or other.key() in done:
          continue
        done.add(other.key())
        for where, data in fn(first, tile_a, other, b.fetch(other), **(fn_kw or {})) or ():
          target.update(where, data)
    return target

  # -- shuffle (shuffle.py:41-160) --------------------------------------------------------------
  def shuffle(self, source, fn, target=None, shape_hint=None, fn_kw=None):
    """target_mapper / notarget_mapper, shuffle.py:41-96: fn(source, extent, **kw) -> [(extent, data)] for every
    tile of `source` in kernel order; the pieces are pushed into `target` (merged by ITS reducer) or become the
    tiles of a new array (shape from the pieces' extents)."""
    pieces = []
    for ex, _, t in self.kernel_order(source):
      for where, data in fn(source, ex, **(fn_kw or {})) or ():
        if target is not None:
          target.update(where, data)
        else:
          pieces.append((where, np.asarray(data)))
    if target is not None:
      return target
    out = self.empty(pieces[0][0].array_shape, pieces[0][1].dtype, None, None)
    full = np.zeros(out.shape, out.dtype)
    for where, data in pieces:
      full[where.to_slice()] = data.reshape(where.shape)
    return self.from_numpy(full)

  # -- dot (dot.py:172-299, map.py:243-334, outer.py:12-99) -----------------------------
  def dot(self, a, b, tile_hint=None):
    if isinstance(b, np.ndarray):                            # dot_map2_np_mapper, dot.py:172-187
      if len(b.shape) == 1:
        shape = (a.shape[0],)
      else:
        shape = (a.shape[0], b.shape[1])
      target = self.empty(shape, a.dtype, np.add)
      for ex, _, t in self.kernel_order(a):
        # dot.py:254-262: map2(a, axes=[0], ...) -- every tile of `a` is re-read as the slab that cuts axis 0
        # (join_mapper, map.py:243-286): a column tile of a wide matrix becomes some ROWS with ALL their columns,
        # one full product per slab, nothing summed across slabs
        ex = change_partition_axis(ex, 0)
        if ex is None:
          continue
        blk = a.fetch(ex).dot(b[ex.ul[1]:ex.lr[1]])
        if len(b.shape) == 1:
          target.update(Extent((ex.ul[0],), (ex.lr[0],), shape), blk)
        else:
          target.update(Extent((ex.ul[0], 0), (ex.lr[0], b.shape[1]), shape), blk)
      return target
    if len(a.shape) == 1 and len(b.shape) == 1:              # dot_map2_vec_mapper, dot.py:189-191
      target = self.empty((1,), a.dtype, np.add)
      for ex, _, t in self.kernel_order(a):
        target.update(Extent((0,), (1,), (1,)), a.fetch(ex).dot(b.fetch(Extent(ex.ul, ex.lr, b.shape))).reshape(1,))
      return target
    if len(a.shape) == 1:                                     # vector . matrix, dot.py:296-299: the vector as a
      target = self.empty((b.shape[1],), a.dtype, np.add, tile_hint)   # 1 x n row, K-split join, is_vec
      for ex, _, t in self.kernel_order(a):
        rows = Extent((ex.ul[0], 0), (ex.lr[0], b.shape[1]), b.shape)
        target.update(Extent((0,), (b.shape[1],), (b.shape[1],)), a.fetch(ex).dot(b.fetch(rows)))
      return target
    if len(b.shape) == 1:
      shape = (a.shape[0],)
    else:
      shape = (a.shape[0], b.shape[1])
      if tile_hint is None:
        tile_hint = shape                                     # dot.py:277-278: ONE result tile
    target = self.empty(shape, a.dtype, np.add, tile_hint)
    if a.shape[0] > a.shape[1]:                               # outer, dot.py:281-285
      whole_b = b.fetch(Extent([0] * len(b.shape), b.shape, b.shape))
      for ex, _, t in self.kernel_order(a):
        first = change_partition_axis(ex, 0)
        blk = a.fetch(first).dot(whole_b)                    # dot_outer_mapper, dot.py:222-238
        if len(b.shape) == 1:
          target.update(Extent((first.ul[0],), (first.lr[0],), shape), blk)
        else:
          target.update(Extent((first.ul[0], 0), (first.lr[0], b.shape[1]), shape), blk)
      return target
    for ex, _, t in self.kernel_order(a):                     # map2 join, map.py:243-286
      first = change_partition_axis(ex, 1)
      if first is None:
        continue
      k0, k1 = first.ul[1], first.lr[1]
      ul, lr = [0] * len(b.shape), list(b.shape)
      ul[0], lr[0] = k0, k1
      a_slab = a.fetch(first)
      b_slab = b.fetch(Extent(ul, lr, b.shape))
      part = a_slab.dot(b_slab)                               # dot_map2_mapper, dot.py:195-217
      if len(b.shape) == 1:
        target.update(Extent((0,), (first.lr[0],), (first.shape[0],)), part)
      else:
        target.update(Extent((0, 0), (first.lr[0], b.shape[1]), (first.shape[0], b.shape[1])), part)
    return target


# ----------------------------------------------------------------- example drivers (BASELINE configs[3], [4])
def kmeans_fit_map2(cl, X, centers, n_clusters, n_iter, reducer=None):
  """KMeans.fit(implementation='map2') restated eagerly (examples/sklearn/cluster/k_means_.py:130-160):
  labels = argmin(cdist) per row tile (:61-66); per-tile counts (:69-72) and masked row sums (:75-97)
  written into ONE whole-array target tile -- with `reducer=None`, as the reference creates those
  targets (:135-141), every tile REPLACES the previous one (tile.pyx:263-268), so the last tile in
  KERNEL order (Cluster.kernel_order: worker by worker, a worker's tiles largest first / last listed first) wins; empty clusters re-seeded from np.random.randn (:145-155); centers = sums / counts."""
  from scipy.spatial.distance import cdist
  num_dim = X.shape[1]
  labels = None
  for _ in range(n_iter):
    labels = cl.empty((X.shape[0],), X.dtype, None)
    for ex, _, _t in cl.kernel_order(X):
      pts = X.fetch(ex)
      labels.update(Extent((ex.ul[0],), (ex.lr[0],), (X.shape[0],)),
                    np.argmin(cdist(pts, centers), axis=1))
    counts = cl.empty((n_clusters,), labels.dtype, reducer)
    for ex, _, _t in cl.kernel_order(labels):
      lab = labels.fetch(ex)
      counts.update(Extent((0,), (n_clusters,), (n_clusters,)),
                    np.bincount(lab.astype(np.int64), minlength=n_clusters))
    sums = cl.empty((n_clusters, num_dim), X.dtype, reducer)
    for ex, _, _t in cl.kernel_order(X):
      pts = X.fetch(ex)
      lab = labels.fetch(Extent((ex.ul[0],), (ex.lr[0],), (X.shape[0],)))
      new_centers = np.zeros((n_clusters, num_dim))
      for i in range(n_clusters):
        new_centers[i] = pts[lab == i].sum(axis=0)
      sums.update(Extent((0, 0), (n_clusters, num_dim), (n_clusters, num_dim)), new_centers)
    counts_h = counts.glom()
    centers = sums.glom()
    zero = (counts_h == 0).reshape(n_clusters)
    if np.any(zero):
      counts_h[zero] = 1
      centers[zero, :] = np.random.randn(np.count_nonzero(zero), num_dim)
    centers = centers / counts_h.reshape(n_clusters, 1)
  return centers, labels


def sgd_train(cl, x, y, iterations, update, alpha=1e-6):
  """SGDRegressor.train (examples/sgd.py:14-40): w drawn with np.random.rand, then
  w -= alpha * sum(update(w), axis=0)."""
  n_dim = x.shape[1]
  w = np.random.rand(n_dim, 1)
  for _ in range(iterations):
    diff = update(cl, x, y, w)
    grad = cl.sum(diff, 0).glom().reshape((n_dim, 1))
    w = w - grad * alpha
  return w


def linear_update(cl, x, y, w):
  """linear_regression.py:10-16: x * (dot(x, w) - y)."""
  yp = cl.dot(x, w)
  return cl.map(lambda a, b, c: a * (b - c), x, yp, y)


def logistic_update(cl, x, y, w):
  """logistic_regression.py:10-17: g = exp(dot(x, w)); x * (g / (g + 1) - y)."""
  g = cl.map(np.exp, cl.dot(x, w))
  yp = cl.map(lambda t: t / (t + 1), g)
  return cl.map(lambda a, b, c: a * (b - c), x, yp, y)
