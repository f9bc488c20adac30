"""TEST INFRASTRUCTURE.  An eager, Spartan-flavoured facade over
oracle/spartan_np.py so that the shared test programs (tests/programs.py) can
be run through the oracle unchanged:  `facade(num_workers)` returns an object
with ones / arange / sum / dot / ... whose results support the arithmetic
operators and .glom() / .optimized() / .force().  Laziness and fusion do not
change values, so the facade evaluates every operation immediately."""
import numpy as np

from . import spartan_np as O


class OArr(object):
  def __init__(self, api, d):
    self.api = api
    self.d = d

  shape = property(lambda self: self.d.shape)
  dtype = property(lambda self: self.d.dtype)

  def glom(self):
    return self.d.glom()

  def optimized(self):
    return self

  def force(self):
    return self

  evaluate = force

  def diagonal(self):
    return self.api.diagonal(self)

  def prod(self, axis=None):
    return self.api.prod(self, axis)

  def _bin(self, other, fn, swap=False):
    a, b = (other, self) if swap else (self, other)
    return self.api.map((a, b), fn)

  def __add__(self, o): return self._bin(o, np.add)
  def __radd__(self, o): return self._bin(o, np.add, True)
  def __sub__(self, o): return self._bin(o, np.subtract)
  def __rsub__(self, o): return self._bin(o, np.subtract, True)
  def __mul__(self, o): return self._bin(o, np.multiply)
  def __rmul__(self, o): return self._bin(o, np.multiply, True)
  def __truediv__(self, o): return self._bin(o, np.divide)
  def __rtruediv__(self, o): return self._bin(o, np.divide, True)
  def __floordiv__(self, o): return self._bin(o, np.floor_divide)
  def __mod__(self, o): return self._bin(o, np.mod)
  def __pow__(self, o): return self._bin(o, np.power)
  def __gt__(self, o): return self._bin(o, np.greater)
  def __lt__(self, o): return self._bin(o, np.less)
  def __ge__(self, o): return self._bin(o, np.greater_equal)
  def __le__(self, o): return self._bin(o, np.less_equal)
  def __eq__(self, o): return self._bin(o, np.equal)
  def __ne__(self, o): return self._bin(o, np.not_equal)
  def __and__(self, o): return self._bin(o, np.logical_and)
  def __or__(self, o): return self._bin(o, np.logical_or)
  def __neg__(self): return self.api.map((self,), np.negative)
  __hash__ = object.__hash__

  # views (slice.py / transpose.py / reshape.py): the oracle materialises them
  def __getitem__(self, idx):
    a = self.glom()
    if isinstance(idx, (int, np.integer)):
      return self.api.from_numpy(np.ascontiguousarray(a[idx:idx + 1]))     # base.py:437-440 keeps the axis
    return self.api.from_numpy(np.ascontiguousarray(a[idx]))


class Facade(object):
  def __init__(self, num_workers=1):
    self.c = O.Cluster(num_workers)

  def _w(self, d):
    return OArr(self, d)

  def _u(self, x):
    if isinstance(x, OArr):
      return x.d
    if isinstance(x, np.ndarray):
      return self.c.from_numpy(x)
    return x

  def map(self, inputs, fn):
    return self._w(self.c.map(fn, *[self._u(i) for i in inputs]))

  # creation
  def ones(self, shape, dtype=np.float32, tile_hint=None): return self._w(self.c.ones(shape, dtype))
  def zeros(self, shape, dtype=np.float32, tile_hint=None): return self._w(self.c.zeros(shape, dtype))

  def full(self, shape, fill_value, dtype=np.float32, tile_hint=None):
    return self._w(self.c.map(lambda t: np.full(t.shape, fill_value, dtype=dtype), self.c.empty(shape, dtype)))

  def arange(self, start=None, stop=None, step=1, dtype=float, tile_hint=None):
    # creation.py:144-206
    shape = None
    if isinstance(start, (tuple, list)):
      shape = start
      start = 0
      if stop is not None:
        start, stop = stop, None
    elif start is None:
      start = 0
    elif stop is None:
      stop, start = start, 0
    if shape is None:
      shape = (int(np.ceil((stop - start) / float(step))),)
    return self._w(self.c.arange(tuple(shape), start, step, dtype))

  def eye(self, N, M=None, k=0, dtype=np.float32, tile_hint=None):
    M = N if M is None else M

    def fn(t, ex):  # creation.py:51-53 (+ column origin)
      return np.eye(ex.lr[0] - ex.ul[0], M=ex.lr[1] - ex.ul[1], k=ex.ul[0] - ex.ul[1] + k, dtype=dtype)
    return self._w(self.c.map_with_location(fn, self.c.empty((N, M), dtype)))

  def from_numpy(self, a, tile_hint=None): return self._w(self.c.from_numpy(a, tile_hint))
  def assign(self, a, idx, value):
    """a[idx] = value as a new array (assign.py:32-52 -> region_map: every tile that meets the region is copied with
    the region replaced, so the result keeps `a`'s dtype whatever the value's)."""
    out = np.array(a.glom(), copy=True)
    v = value.glom() if isinstance(value, OArr) else value
    if np.isscalar(idx):
      idx = slice(idx, idx + 1)
    out[idx] = np.asarray(v).astype(out.dtype) if not np.isscalar(v) else v
    return self.from_numpy(out)

  def write(self, array, src_slices, data, dst_slices):
    """array[src_slices] = data[dst_slices] (write_array.py:82-94: an update of the evaluated array's tiles, merged
    into the TARGET's dtype, tile.pyx:267)."""
    out = np.array(array.glom(), copy=True)
    d = data.glom() if isinstance(data, OArr) else np.asarray(data)
    out[src_slices] = d[dst_slices].astype(out.dtype)
    return self.from_numpy(out)

  def transpose(self, x): return self.from_numpy(np.ascontiguousarray(x.glom().T))
  def reshape(self, x, shape): return self.from_numpy(np.ascontiguousarray(x.glom().reshape(shape)))
  def ravel(self, x): return self.from_numpy(np.ascontiguousarray(x.glom().ravel()))

  # elementwise
  def sqrt(self, v): return self.map((v,), np.sqrt)
  def exp(self, v): return self.map((v,), np.exp)
  def ln(self, v): return self.map((v,), np.log)
  log = ln
  def abs(self, v): return self.map((v,), np.abs)
  def square(self, v): return self.map((v,), np.square)
  def maximum(self, a, b): return self.map((a, b), np.maximum)
  def minimum(self, a, b): return self.map((a, b), np.minimum)
  def astype(self, x, dtype): return self.map((x,), lambda t: t.astype(dtype))

  # reductions
  def sum(self, x, axis=None, tile_hint=None): return self._w(self.c.sum(self._u(x), axis))
  def max(self, x, axis=None, tile_hint=None): return self._w(self.c.max(self._u(x), axis))
  def min(self, x, axis=None, tile_hint=None): return self._w(self.c.min(self._u(x), axis))
  def argmax(self, x, axis=None): return self._w(self.c.argmax(self._u(x), axis))
  def argmin(self, x, axis=None): return self._w(self.c.argmin(self._u(x), axis))

  def prod(self, x, axis=None):
    d = self._u(x)
    dt = np.int64 if d.dtype == np.int32 else d.dtype          # mathematics.py:150-154
    return self._w(self.c.reduce(d, axis, dt, lambda ex, t, a: t.prod(a), np.multiply))

  def all(self, x, axis=None):
    return self._w(self.c.reduce(self._u(x), axis, np.bool_, lambda ex, t, a: np.all(t, axis=a), np.logical_and))

  def any(self, x, axis=None):
    return self._w(self.c.reduce(self._u(x), axis, np.bool_, lambda ex, t, a: np.any(t, axis=a), np.logical_or))

  def count_nonzero(self, x, axis=None, tile_hint=None):       # sorting.py:126-150
    fn = lambda ex, t, a: np.asarray(np.count_nonzero(t)) if a is None else (t > 0).sum(a)
    return self._w(self.c.reduce(self._u(x), axis, np.int64, fn, np.add))

  def count_zero(self, x, axis=None):                           # sorting.py:153-172
    fn = lambda ex, t, a: np.asarray(np.prod(ex.shape) - np.count_nonzero(t)) if a is None else (t == 0).sum(a)
    return self._w(self.c.reduce(self._u(x), axis, np.int64, fn, np.add))

  def mean(self, x, axis=None):                                 # statistics.py:64-76
    if axis is None:
      return self.sum(x, axis) / int(np.prod(x.shape))
    return self.sum(x, axis) / int(x.shape[axis])

  def std(self, a, axis=None):                                  # statistics.py:86-102
    c = self.astype(a, np.float64)
    return self.sqrt(self.mean(c ** 2, axis) - self.mean(c, axis) ** 2)

  # -- the rest of the builder namespace: statistics.py:105-219, creation.py:225-330, manipulation.py:44-80 ------
  def bincount(self, v, weights=None, minlength=None):          # statistics.py:105-137
    minval, maxval = self.min(v).glom(), self.max(v).glom()
    assert minval > 0
    minlength = int(maxval) + 1 if minlength is None else max(int(maxval) + 1, int(minlength))

    def mapper(ex, tiles):
      if len(tiles) > 1:
        res = np.bincount(tiles[0], weights=tiles[1], minlength=minlength)
        # tile.pyx:46: a tile that is not written whole is built around the float64 data with the int dtype
        assert ex.shape == ex.array_shape, 'Failed: float64 == %s' % tiles[0].dtype
      else:
        res = np.bincount(tiles[0], minlength=minlength)
      yield O.Extent((0,), res.shape, res.shape), res
    arrays = [self._u(v)] + ([self._u(weights)] if weights is not None else [])
    return self._w(self.c.map2(arrays, (), mapper, (minlength,), np.add))

  def normalize(self, array, axis=None):                         # statistics.py:140-182
    norm_value = self.sum(array, axis).glom()

    def mapper(tile, ex):
      tile = np.array(tile)              # (the reference divides the fetched tile in place)
      if axis is None:
        tile /= norm_value
      elif axis == 0:
        tile[:, 0] /= norm_value[ex.ul[1]]
      elif axis == 1:
        tile[0, :] /= norm_value[ex.ul[0]]
      return tile
    return self._w(self.c.map_with_location(mapper, self._u(array)))

  def norm(self, array, ord=2):                                  # statistics.py:185-219
    assert ord == 1 or ord == 2
    x = self._u(array)
    if ord == 1:
      return np.max(self.c.reduce(x, 0, x.dtype, lambda ex, d, a: np.abs(d).sum(a), np.add).glom())
    assert len(x.shape) == 1 or len(x.shape) == 2 and x.shape[1] == 1, 'matrix norm-2 is not support!'
    return np.sqrt(self.c.reduce(x, 0, x.dtype, lambda ex, d, a: np.square(d).sum(a), np.add).glom())

  def diagflat(self, array):                                      # creation.py:225-262
    x = self._u(array)
    n = int(np.prod(x.shape))
    shape = (n, n)

    def mapper(extents, tiles):
      ex, tile = extents[0], tiles[0]
      head = O.ravelled_pos(ex.ul, ex.array_shape)
      tail = O.ravelled_pos([l - 1 for l in ex.lr], ex.array_shape)
      result = np.diagflat(tile)
      if head != 0:
        result = np.hstack((np.zeros(((tail - head + 1), head)), result))
      if tail + 1 != shape[0]:
        result = np.hstack((result, np.zeros((tail - head + 1, shape[0] - (tail + 1)))))
      yield O.ex_create((head, 0), (tail + 1, shape[1]), shape), result
    return self._w(self.c.map2([x], (0,), mapper, shape))

  def diagonal(self, a):                                          # creation.py:265-302
    x = self._u(a)
    if len(x.shape) < 2:
      raise ValueError('diag requires an array of at least two dimensions')
    shape = (min(x.shape),)

    def mapper(ex, tiles):
      first = max(*ex.ul)
      slices = []
      for i in range(len(ex.ul)):
        if first >= ex.lr[i]:
          return
        slices.append(slice(first - ex.ul[i], ex.shape[i]))
      result = tiles[0][tuple(slices)].diagonal()
      yield O.ex_create((first,), (first + result.shape[0],), shape), result
    return self._w(self.c.map2([x], (), mapper, shape))

  def diag(self, array, offset=0):                                # creation.py:305-330
    if offset != 0:
      raise NotImplementedError
    if len(array.shape) == 1:
      return self.diagflat(array)
    if len(array.shape) == 2:
      return self.diagonal(array)
    raise ValueError('Input must be 1- or 2-d.')

  def concatenate(self, a, b, axis=0):                            # manipulation.py:44-80
    xa, xb = self._u(a), self._u(b)
    shape = [d1 + d2 if i == axis else d1 for i, (d1, d2) in enumerate(zip(xa.shape, xb.shape))]
    if any(d1 != d2 for i, (d1, d2) in enumerate(zip(xa.shape, xb.shape)) if i != axis):
      raise ValueError('all the input array dimensions except for the concatenation axis must match exactly')
    if len(xa.shape) > 1:                                         # extent.pyx largest_dim_axis, the join axis excluded
      part = max((i for i in range(len(xa.shape)) if i != axis), key=lambda i: (xa.shape[i], -i))
    else:
      part = 0

    def mapper(extents, tiles):
      if len(extents[0].shape) > 1:
        lr = list(extents[0].lr)
        lr[axis] += extents[1].shape[axis]
        yield O.ex_create(extents[0].ul, lr, shape), np.concatenate((tiles[0], tiles[1]), axis=axis)
      else:
        yield O.ex_create(extents[0].ul, extents[0].lr, shape), tiles[0]
        off = extents[0].array_shape[0]
        yield O.ex_create((off + extents[1].ul[0],), (off + extents[1].lr[0],), shape), tiles[1]
    return self._w(self.c.map2([xa, xb], (part, part), mapper, shape))

  def dot(self, a, b, tile_hint=None):
    return self._w(self.c.dot(self._u(a), b if isinstance(b, np.ndarray) else self._u(b), tile_hint))

  # user-function joins (tests/join_programs.py): the tile functions see the oracle's own arrays and extents
  def ndarray(self, shape, dtype=np.float32, tile_hint=None, reduce_fn=None):
    return self._w(self.c.empty(shape, dtype, reduce_fn, tile_hint))

  def map2(self, arrays, axes=(), fn=None, fn_kw=None, shape=None, reducer=None, tile_hint=None):
    arrays = list(arrays) if isinstance(arrays, (list, tuple)) else [arrays]
    axes = tuple(axes) if isinstance(axes, (list, tuple)) else (axes,)
    return self._w(self.c.map2([self._u(a) for a in arrays], axes, fn, shape, reducer, fn_kw))

  def outer(self, arrays, axes, fn, fn_kw=None, shape=None, tile_hint=None, reducer=None, dtype=None):
    return self._w(self.c.outer([self._u(a) for a in arrays], tuple(axes), fn, shape, reducer, fn_kw, tile_hint, dtype))

  def shuffle(self, v, fn, shape_hint=None, target=None, kw=None):
    return self._w(self.c.shuffle(self._u(v), fn, None if target is None else self._u(target), shape_hint, kw))


def facade(num_workers=1):
  return Facade(num_workers)
