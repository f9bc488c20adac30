"""sort / argsort / partition / argpartition: mirror of the reference's spartan/expr/operator/sort.py.

Along an axis the array is partitioned on ANOTHER axis (map2) so every line to be sorted lies inside one
tile, and the tile body -- `np.sort(tile, axis)` in the reference -- is sp_sort_rows on the GPU (a stable sort:
np.argsort(kind='stable'); NumPy's default argsort leaves the order of equal keys undefined, so the two agree
wherever the keys of a line are distinct and on the sorted VALUES always).  `sort(axis=None)` is the reference's sample sort: sort every tile, pick partition keys
from a sample, count every tile's elements per partition, let partition p fetch its pieces from every tile and
sort them; the result is a 1-D array of prod(shape) elements.
"""
import numpy as np

from . import base
from .map import map2
from .. import context
from ..array import distarray, extent
from ..context import LocalKernelResult


class _Payload(object):
  """A host value returned by a tile kernel to the driver (the reference's tile_operation results)."""
  __slots__ = ('v',)

  def __init__(self, v):
    self.v = v


def _backend():
  return context.get().backend


def _absent(t):
  return isinstance(t, distarray.Absent)


def _sort_mapper(extents, tiles, axis=None):
  """sort.py:68-69."""
  t = tiles[0]
  yield extents[0], (t if _absent(t) else _backend().sort_axis(t, axis, indices=False))


def _argsort_mapper(extents, tiles, axis=None):
  """sort.py:137-138."""
  t = tiles[0]
  yield extents[0], (distarray.Absent(t.shape, np.int64) if _absent(t) else _backend().sort_axis(t, axis, indices=True))


def _along_axis(array, axis, fn, dtype=None):
  array = base.lazify(array)
  if axis < 0:
    axis = len(array.shape) + axis
  partition_axis = extent.largest_dim_axis(array.shape, exclude_axes=[axis])
  return map2(array, partition_axis, fn=fn, fn_kw={'axis': axis}, shape=array.shape, dtype=dtype)


def argsort(array, axis=-1):
  """sort.py:141-156.  The map2 target takes the INPUT's dtype (map.py:317-318), so the indices of a float array
  come back as floats, like the reference's."""
  assert axis is not None, "Spartan doesn't support argsort when axis == None now"
  return _along_axis(array, axis, _argsort_mapper)


def partition(array, kth, axis=-1):
  """sort.py:76-99.  (The reference's tile body calls np.partition without `kth` and cannot run; a full sort
  satisfies np.partition's contract for every kth.)"""
  assert axis is not None, "Spartan doesn't support partition when axis == None now"
  return _along_axis(array, axis, _sort_mapper)


def argpartition(array, kth, axis=-1):
  """sort.py:163-178 (see partition)."""
  assert axis is not None, "Spartan doesn't support argpartition when axis == None now"
  return _along_axis(array, axis, _argsort_mapper)


# ------------------------------------------------------------------ sample sort (axis=None)
def _local_sort_kernel(ex, source=None, flat=None, offsets=None, rate=None):
  """sort.py:11-25 _sample_sort_mapper: sort the tile (flattened) into the 1-D array `flat` and return an evenly
  spaced sample of the sorted tile (the reference draws the sample at random; any sample gives the same result)."""
  ctx = context.get()
  data = source.fetch(ex)
  n = int(np.prod(ex.shape))
  off = offsets[ex]
  dst = extent.create((off,), (off + n,), flat.shape)
  if ctx.executing:
    srt = ctx.backend.sort_axis(data.reshape(1, n), 1, indices=False).reshape(n)
    k = max(1, int(n * rate))
    sample = ctx.backend.to_numpy(srt[:: max(1, n // k)])
  else:
    srt = distarray.Absent((n,), source.dtype)
    sample = None
  flat.update(dst, srt, wait=False, owned=True)
  return LocalKernelResult(result=[(ex, _Payload(sample))])


def _count_kernel(ex, flat=None, keys=None):
  """sort.py:28-40 _partition_count_mapper: position of every partition key in the sorted tile."""
  ctx = context.get()
  data = flat.fetch(ex)
  if ctx.executing:
    idx = np.searchsorted(ctx.backend.to_numpy(data), keys, side='left')   # tile-sized download: see DESIGN
    idx = np.concatenate([[0], idx, [ex.shape[0]]]).astype(np.int64)
  else:
    idx = None
  return LocalKernelResult(result=[(ex, _Payload(idx))])


class SortExpr(base.Expr):
  """sort(axis=None), sort.py:72-103."""
  members = ('array', 'sample_rate')

  def dependencies(self):
    return {'array': self.array}

  def visit(self, visitor):
    return base.expr_like(self, array=visitor.visit(self.array), sample_rate=self.sample_rate)

  def compute_shape(self):
    return (int(np.prod(self.array.shape)),)

  def pretty_str(self):
    return 'Sort[%d](%s)' % (self.expr_id, self.array)

  def _evaluate(self, ctx, deps):
    array = deps['array']
    world = ctx.world
    total = int(np.prod(array.shape))
    exts = sorted(array.tiles.keys(), key=lambda e: e.ul)
    offsets, off = {}, 0
    for e in exts:
      offsets[e] = off
      off += int(np.prod(e.shape))
    # 1. every tile sorted into its own segment of a 1-D array whose tiles are those segments
    tiles = distarray.collections.OrderedDict()
    for e in exts:
      n = int(np.prod(e.shape))
      worker = array.tiles[e].worker
      t = distarray.tile.from_shape((n,), array.dtype, distarray.tile.TYPE_DENSE) if ctx.is_local_worker(worker) else None
      tiles[extent.create((offsets[e],), (offsets[e] + n,), (total,))] = ctx.create(t, hint=worker)
    flat = distarray.DistArrayImpl(shape=(total,), dtype=array.dtype, tiles=tiles, reducer_fn=None)
    res = array.foreach_tile(mapper_fn=_local_sort_kernel,
                             kw=dict(source=array, flat=flat, offsets=offsets, rate=self.sample_rate))
    samples = {}
    for tid, items in res.items():
      for ex, s in items:
        samples[ex.ul] = s.v
    if world.distributed:
      merged = {}
      for part in world.all_gather_object({k: v for k, v in samples.items() if v is not None}):
        merged.update(part)
      samples = merged
    sorted_samples = np.sort(np.concatenate([samples[e.ul] for e in exts]), axis=None)
    # 2. partition keys and the position of each key in every sorted tile
    steps = max(1, sorted_samples.size // len(exts))
    keys = sorted_samples[steps::steps][:len(exts) - 1]
    res = flat.foreach_tile(mapper_fn=_count_kernel, kw=dict(flat=flat, keys=keys))
    counts = {}
    for tid, items in res.items():
      for ex, idx in items:
        counts[ex.ul[0]] = idx.v
    if world.distributed:
      merged = {}
      for part in world.all_gather_object({k: v for k, v in counts.items() if v is not None}):
        merged.update(part)
      counts = merged
    # 3. partition p = the p-th slice of every sorted tile, fetched and sorted on the p-th tile's worker
    fexts = sorted(flat.tiles.keys(), key=lambda e: e.ul)
    nparts = len(keys) + 1
    sizes = [sum(int(counts[f.ul[0]][p + 1] - counts[f.ul[0]][p]) for f in fexts) for p in range(nparts)]
    out_tiles = distarray.collections.OrderedDict()
    dst = 0
    plan = []
    for p in range(nparts):
      worker = flat.tiles[fexts[p % len(fexts)]].worker
      oex = extent.create((dst,), (dst + sizes[p],), (total,))
      plan.append((p, worker, oex))
      dst += sizes[p]
    for p, worker, oex in plan:
      if sizes[p] == 0:
        continue
      with ctx.on_worker(worker):
        pieces = []
        for f in fexts:
          a, b = int(counts[f.ul[0]][p]), int(counts[f.ul[0]][p + 1])
          if b > a:
            pieces.append(flat.fetch(extent.create((f.ul[0] + a,), (f.ul[0] + b,), (total,))))
        if ctx.executing:
          cat = pieces[0]
          for q in pieces[1:]:
            cat = ctx.backend.concat(cat, q, axis=0)
          srt = ctx.backend.sort_axis(cat.reshape(1, sizes[p]), 1, indices=False).reshape(sizes[p])
          t = distarray.tile.from_data(srt, dtype=array.dtype)
        else:
          t = None
        out_tiles[oex] = ctx.create(t, hint=worker)
    return distarray.DistArrayImpl(shape=(total,), dtype=array.dtype, tiles=out_tiles, reducer_fn=None)


def sort(array, axis=-1, sample_rate=0.1):
  """sort.py:72-103."""
  if axis is not None:
    return _along_axis(array, axis, _sort_mapper)
  return SortExpr(array=base.lazify(array), sample_rate=sample_rate)
