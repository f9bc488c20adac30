"""scan: cumulative sum / product along an axis, or over the flattened array (reference
spartan/expr/operator/scan.py).  Three steps as there: per-tile reductions along the axis (shuffle),
a small driver-side scan of those tile totals, then one pass per tile: the tile's own cumulative
scan (sp_cumscan) combined with the total of everything in front of it."""
import numpy as np

from .map import map_with_location
from .shuffle import shuffle
from .. import context
from ..array import distarray, extent
from ..util import divup

_PAIRS = {np.sum: ('SUM', np.cumsum, False), np.prod: ('PROD', np.cumprod, True)}


class _Probe(object):
  """Stands in for a tile to find out which array method a user function calls (the reference's own test passes
  `lambda x, **kw: x.sum(axis=kw['axis'])` / `x.cumsum(...)`, tests/test_scan.py:38-39)."""

  def __init__(self):
    self.called = None

  def _record(self, name):
    def method(*a, **k):
      self.called = name
      return self
    return method

  def __getattr__(self, name):
    if name in ('sum', 'prod', 'cumsum', 'cumprod'):
      return self._record(name)
    raise AttributeError(name)


def _canonical(fn):
  """np.sum / np.prod / np.cumsum / np.cumprod, or a user function that only calls the method of that name."""
  if fn in (np.sum, np.prod, np.cumsum, np.cumprod):
    return fn
  probe = _Probe()
  try:
    fn(probe, axis=0)
  except Exception:
    return fn
  return {'sum': np.sum, 'prod': np.prod, 'cumsum': np.cumsum, 'cumprod': np.cumprod}.get(probe.called, fn)


def _kind(reduce_fn, scan_fn):
  reduce_fn, scan_fn = _canonical(reduce_fn), _canonical(scan_fn)
  try:
    red, want_scan, product = _PAIRS[reduce_fn]
  except (KeyError, TypeError):
    raise NotImplementedError('scan: reduce_fn must be np.sum or np.prod on the GPU tile path, got %r' % (reduce_fn,))
  if scan_fn is not want_scan:
    raise NotImplementedError('scan: %s pairs with %s, got %r' % (reduce_fn.__name__, want_scan.__name__, scan_fn))
  return red, product


def _dense(data):
  """A sparse tile is scanned as a dense one (np.cumsum has no sparse form; the reference's test densifies the
  expected value the same way, tests/test_scan.py:33-35)."""
  from ..array import tile as tile_mod
  return context.get().backend.sparse_to_dense(data) if tile_mod.is_sparse_blob(data) else data


def _scan_reduce_mapper(array, ex, reduce_fn, axis):
  """scan.py:24-39: this tile's total along `axis`, filed under the tile's index along that axis."""
  ctx = context.get()
  red, _ = _kind(reduce_fn, _PAIRS[_canonical(reduce_fn)][1])
  data = array.fetch(ex)
  axis_shape = array.tile_shape()[axis]
  tid = (ex.lr[axis] - 1) // axis_shape
  new_ul, new_lr, new_shape = list(ex.ul), list(ex.lr), list(ex.array_shape)
  new_ul[axis], new_lr[axis] = tid, tid + 1
  new_shape[axis] = divup(array.shape[axis], axis_shape)
  dst_ex = extent.create(new_ul, new_lr, new_shape)
  if isinstance(data, distarray.Absent) or not ctx.executing:
    yield (dst_ex, distarray.Absent(dst_ex.shape, array.dtype))
    return
  local = ctx.backend.reduce_axis(_dense(data), red, axis)
  yield (dst_ex, local.reshape(dst_ex.shape))


def _scan_mapper(tile, ex, scan_fn=None, axis=None, scan_base=None, tile_shape=None, product=False):
  """scan.py:42-63: scan the tile along the axis and fold in what precedes it."""
  be = context.get().backend
  base_slice = [slice(ul, lr) for ul, lr in zip(ex[0], ex[1])]
  if axis is None:
    axis = 1
    tile_id = (ex[1][axis] - 1) // tile_shape[axis]
    base_slice[axis] = slice(tile_id, tile_id + 1)
    base = scan_base[tuple(base_slice)]
  else:
    tile_id = (ex[1][axis] - 1) // tile_shape[axis]
    base = None
    if tile_id > 0:
      base_slice[axis] = slice(tile_id - 1, tile_id)
      base = scan_base[tuple(base_slice)]
  out = be.cumscan(_dense(tile), axis, product)
  if base is not None:
    base = np.ascontiguousarray(base).astype(be.dtype_of(out))
    out = be.evaluate_fn(np.multiply if product else np.add, [out, base], {}, tuple(out.shape))
  return out


_scan_mapper._sp_tile_fn = True


def scan(array, reduce_fn=np.sum, scan_fn=np.cumsum, axis=None):
  """Scan `array` over `axis` (None: the flattened array, result in the array's shape); scan.py:67-97."""
  red, product = _kind(reduce_fn, scan_fn)
  reduce_result = shuffle(array, fn=_scan_reduce_mapper,
                          kw={'axis': axis if axis is not None else 1, 'reduce_fn': reduce_fn},
                          shape_hint=array.shape)
  fetch_result = reduce_result.optimized().glom()
  if axis is None:
    ident = np.ones(1) if product else np.zeros(1)
    fetch_result = np.concatenate((ident, scan_fn(fetch_result, axis=None)[:-1])).reshape(fetch_result.shape)
  else:
    fetch_result = scan_fn(fetch_result, axis=axis)
  return map_with_location(array, _scan_mapper,
                           fn_kw={'scan_fn': scan_fn, 'axis': axis, 'scan_base': fetch_result,
                                  'tile_shape': array.evaluate().tile_shape(), 'product': product})
