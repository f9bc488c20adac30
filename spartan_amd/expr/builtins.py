"""NumPy-style builder functions feeding the tile path: mirrors of the
reference's spartan/expr/{creation,mathematics,statistics,logic,sorting,arrays}.py.

The module-level `_xxx` functions are the *local* (per-tile) functions the
reference hands to map/reduce.  Here they are never called by the product: they
are symbols the backend lowers to kernel ops (spartan_amd/lower.py registers a
rule for each); their bodies are kept as the NumPy definition of the semantics
(they are what oracle/np_backend.py executes in the CPU tests).
"""
import numpy as np

from . import base
from .base import Expr
from .map import map, map2, map_with_location
from .ndarray import ndarray
from .optimize import not_idempotent
from .reduce import reduce
from .. import context
from ..array import distarray, extent
from ..context import LocalKernelResult

# ---------------------------------------------------------------- creation.py
def empty(shape, dtype=np.float32, tile_hint=None):
  return ndarray(shape, dtype=dtype, tile_hint=tile_hint)


def empty_like(array, dtype=None, tile_hint=None):
  if dtype is None:
    dtype = array.dtype
  return ndarray(array.shape, dtype=dtype, tile_hint=tile_hint, sparse=getattr(array, 'sparse', False))


def _tocoo(data):
  """arrays.py:44-45 `data.tocoo()`: a device tile has one format (CSR), so this is the identity on it."""
  return data.tocoo() if hasattr(data, 'tocoo') else data


_tocoo._sp_tile_fn = True


def tocoo(array):
  """arrays.py:48-55."""
  return map(array, fn=_tocoo)


def sparse_empty(shape, dtype=np.float32, tile_hint=None):
  """creation.py:25-32."""
  return ndarray(shape, dtype=dtype, tile_hint=tile_hint, sparse=True)


def _make_sparse_diagonal(tile, ex):
  """creation.py:209-220: the part of the unit diagonal that crosses this tile, as a sparse block
  (built as index arrays on the host -- min(rows, cols) entries -- and handed to the backend)."""
  import scipy.sparse
  ul, lr = ex[0], ex[1]
  lo = ul[0] if ul[0] > ul[1] else ul[1]      # (max / min are the array reductions in this module)
  hi = lr[0] if lr[0] < lr[1] else lr[1]
  n = hi - lo if hi > lo else 0
  idx = np.arange(lo, lo + n)
  dtype = np.dtype(tile.dtype)
  mat = scipy.sparse.coo_matrix((np.ones(n, dtype=dtype), (idx - ul[0], idx - ul[1])),
                                shape=(lr[0] - ul[0], lr[1] - ul[1]), dtype=dtype)
  return context.get().backend.sparse_blob(mat, dtype)


_make_sparse_diagonal._sp_tile_fn = True


def sparse_diagonal(shape, dtype=np.float32, tile_hint=None):
  """creation.py:223-225."""
  return map_with_location(ndarray(shape, dtype, tile_hint, sparse=True), _make_sparse_diagonal)


def _make_sparse_rand(tile, density=None, dtype=None, format=None):
  """srandom.py:57-65 (scipy.sparse.rand per tile): density * size entries at uniform random positions with
  uniform [0, 1) values, drawn by the backend's generator (positions that collide are merged)."""
  return context.get().backend.sparse_random(tuple(tile.shape), density, dtype)


_make_sparse_rand._sp_tile_fn = True


@not_idempotent
def sparse_rand(shape, density=0.001, format='lil', dtype=np.float32, tile_hint=None):
  """srandom.py:121-146.  `format` is accepted and ignored: a device tile has one format (CSR)."""
  for s in shape:
    assert isinstance(s, (int, np.integer))
  return map(ndarray(shape, dtype=dtype, tile_hint=tile_hint, sparse=True), fn=_make_sparse_rand,
             fn_kw={'dtype': dtype, 'density': density, 'format': format})


def _make_zeros(input):
  """creation.py:67-68."""
  return np.zeros(input.shape, input.dtype)


def zeros(shape, dtype=np.float32, tile_hint=None):
  """creation.py:71-81."""
  return map(ndarray(shape, dtype=dtype, tile_hint=tile_hint), fn=_make_zeros)


def zeros_like(array, dtype=None, tile_hint=None):
  if dtype is None:
    dtype = array.dtype
  return zeros(array.shape, dtype=dtype, tile_hint=tile_hint)


def _make_ones(input):
  """creation.py:92-93."""
  return np.ones(input.shape, input.dtype)


def ones(shape, dtype=np.float32, tile_hint=None):
  """creation.py:96-106."""
  return map(ndarray(shape, dtype=dtype, tile_hint=tile_hint), fn=_make_ones)


def ones_like(array, dtype=None, tile_hint=None):
  if dtype is None:
    dtype = array.dtype
  return ones(array.shape, dtype=dtype, tile_hint=tile_hint)


def _full_mapper(tile, fill_value=None, dtype=None):
  """creation.py:117-118."""
  return np.full(tile.shape, fill_value, dtype=dtype)


def full(shape, fill_value, dtype=np.float32, tile_hint=None):
  """creation.py:121-123 (the reference forgets to forward fill_value; we pass it)."""
  return map(ndarray(shape, dtype=dtype, tile_hint=tile_hint), fn=_full_mapper,
             fn_kw={'fill_value': fill_value, 'dtype': dtype})


def full_like(array, fill_value, dtype=None, tile_hint=None):
  if dtype is None:
    dtype = array.dtype
  return full(array.shape, fill_value, dtype, tile_hint)


def ravel_contiguous(ul, lr, array_shape):
  """Do the elements of the box [ul, lr) follow one another in the row-major order of the whole array?  (Every axis
  after the LAST cut one is complete by definition; every axis before it must have extent 1: row bands of a
  matrix, any range of a vector.)"""
  cut = [i for i in range(len(array_shape)) if lr[i] - ul[i] != array_shape[i]]
  if not cut:
    return True
  for i in range(cut[-1]):          # (`all` is the array builder in this module)
    if lr[i] - ul[i] != 1:
      return False
  return True


def _arange_mapper(tile, ex, start=None, stop=None, step=None, dtype=None):
  """creation.py:134-141: a tile whose elements are consecutive in the array's row-major order counts on from the
  position of its first element.  The reference applies that formula to EVERY tile, which is wrong for a column
  or block tile (its auto-tiling pass produces them, and its arange then differs from np.arange); such a tile gets
  the value of each element's own position here."""
  ul, lr, array_shape = ex
  if ravel_contiguous(ul, lr, array_shape):
    pos = extent.ravelled_pos(ul, array_shape)
    ex_start = pos * step + start
    ex_stop = np.prod(tile.shape) * step + ex_start
    return np.arange(ex_start, ex_stop, step, dtype=dtype).reshape(tile.shape)
  where = np.ravel_multi_index([np.arange(u, l).reshape([-1 if j == i else 1 for j in range(len(ul))])
                                for i, (u, l) in enumerate(zip(ul, lr))], array_shape)
  return np.asarray(where * step + start).astype(dtype).reshape(tile.shape)


def arange(start=None, stop=None, step=1, dtype=float, tile_hint=None):
  """np.arange spread over tiles.  Call forms are the reference's (creation.py:144-206): arange(stop),
  arange(start, stop[, step]), and -- with a shape in first place -- arange(shape) / arange(shape, first_value):
  an array of that shape counting on in row-major order."""
  if start is None and stop is None:
    raise ValueError('No valid parameters')
  if isinstance(start, (tuple, list)):
    shape, first, last = tuple(start), (0 if stop is None else stop), None
  else:
    first, last = (0, start) if stop is None else (0 if start is None else start, stop)
    shape = (int(np.ceil((last - first) / float(step))),)
  return map_with_location(ndarray(shape, dtype, tile_hint), _arange_mapper,
                           fn_kw={'start': first, 'stop': last, 'step': step, 'dtype': dtype})


def _eye_mapper(tile, ex, k=None, dtype=None):
  """creation.py:51-53."""
  return np.eye(ex[1][0] - ex[0][0], M=(ex[1][1] - ex[0][1]), k=(ex[0][0] - ex[0][1] + k), dtype=dtype)


def eye(N, M=None, k=0, dtype=np.float32, tile_hint=None):
  """creation.py:56-60 (the reference's k offset ignores the column origin of
  the tile, which is only right for row tiles; the column origin is included)."""
  if M is None:
    M = N
  return map_with_location(ndarray((N, M), dtype, tile_hint), _eye_mapper, fn_kw={'k': k, 'dtype': dtype})


def identity(n, dtype=np.float32, tile_hint=None):
  return eye(n, dtype=dtype, tile_hint=tile_hint)


def from_numpy(npa, tile_hint=None):
  """write_array.py:424-445 (`from_numpy`): load a host array -- or a scipy.sparse matrix, which becomes a
  sparse array (:435-441) -- into tiles."""
  ctx = context.get()
  from ..array import tile as tile_mod
  if tile_mod.is_sparse_blob(npa):
    csr = npa.tocsr()
    arr = distarray.create(csr.shape, csr.dtype, tile_hint=tile_hint, sparse=True)
    for ex, tid in arr.tiles.items():
      if ctx.is_local(tid):
        ctx.tile(tid).update(ctx.backend, None, ctx.backend.sparse_blob(csr[ex.to_slice()], csr.dtype), None)
    arr._touched = True
    arr.mark_written()
    return base.Val(val=arr)
  npa = np.asarray(npa)
  arr = distarray.create(npa.shape, npa.dtype, tile_hint=tile_hint)
  for ex, tid in arr.tiles.items():
    if ctx.is_local(tid):
      data = ctx.backend.from_numpy(np.ascontiguousarray(npa[ex.to_slice()]))
      ctx.tile(tid).update(ctx.backend, None, data.reshape(ex.shape), None, owned=True)
  arr._touched = True
  arr.mark_written()
  return base.Val(val=arr)


def from_tile_fn(shape, dtype, fn, tile_hint=None, sparse=False):
  """Build a DistArray whose tiles are produced IN PLACE on their owning worker:
  fn(extent) -> backend tensor (or, with sparse=True, sparse blob) of extent.shape (e.g. device-side
  RNG).  The loader analogue of from_numpy for data that never exists on the host."""
  ctx = context.get()
  arr = distarray.create(shape, dtype, tile_hint=tile_hint, sparse=sparse)
  for ex, tid in arr.tiles.items():
    if ctx.is_local(tid):
      data = fn(ex)
      if not sparse:
        data = data.reshape(ex.shape)
      ctx.tile(tid).update(ctx.backend, None, data, None, owned=True)
  arr._touched = True
  arr.mark_written()
  return base.Val(val=arr)


# ---------------------------------------------------------------- srandom.py
def _make_rand(input):
  """srandom.py:38-40."""
  return np.random.rand(*input.shape)


def _make_randn(input):
  """srandom.py:43-45."""
  return np.random.randn(*input.shape)


def _make_randint(input, low=0, high=10):
  """srandom.py:48-50."""
  return np.random.randint(low, high, size=input.shape)


# (kind, result dtype): the HIP backend fills these tiles with its counter-based generator
# (sp_random_fill) instead of calling NumPy on the host
_make_rand._sp_random = ('uniform', np.float64)
_make_randn._sp_random = ('normal', np.float64)
_make_randint._sp_random = ('randint', np.int64)


def set_random_seed(seed=None):
  """srandom.py:23-35 re-seeds every worker from the clock.  Here two generators are seeded: the backend's
  per-tile generator, mixed with the rank so the ranks fill their tiles from distinct streams, and the DRIVER's
  np.random, with the SAME value on every rank -- the driver program runs on every rank and whatever it draws
  (start weights, start centers) is replicated state.  A clock seed is taken on rank 0 and sent to the others."""
  import os
  import time
  ctx = context.get()
  if seed is None:
    seed = ctx.world.broadcast_object((int(time.time() * 100000) + os.getpid()) % 4294967295, 0)
  if hasattr(ctx.backend, 'seed_random'):
    ctx.backend.seed_random(int(seed) * 1000003 + ctx.world.rank)
  np.random.seed(int(seed) % 4294967295)


def _tile_hint_kw(kw):
  tile_hint = kw.pop('tile_hint', None)
  return tile_hint


@not_idempotent
def rand(*shape, **kw):
  """Uniform [0, 1) array (srandom.py:68-85)."""
  tile_hint = _tile_hint_kw(kw)
  assert len(kw) == 0, 'Unknown keywords %s' % kw
  for s in shape:
    assert isinstance(s, (int, np.integer))
  return map(ndarray(shape, dtype=np.float64, tile_hint=tile_hint), fn=_make_rand)


@not_idempotent
def randn(*shape, **kw):
  """Standard normal array (srandom.py:87-101)."""
  tile_hint = _tile_hint_kw(kw)
  for s in shape:
    assert isinstance(s, (int, np.integer))
  return map(ndarray(shape, dtype=np.float64, tile_hint=tile_hint), fn=_make_randn)


@not_idempotent
def randint(*shape, **kw):
  """Integers in [low, high) (srandom.py:103-118)."""
  tile_hint = _tile_hint_kw(kw)
  for s in shape:
    assert isinstance(s, (int, np.integer))
  return map(ndarray(shape, dtype=np.float64, tile_hint=tile_hint), fn=_make_randint, fn_kw=kw)


# ------------------------------------------------------------- mathematics.py
def add(a, b): return map((a, b), fn=np.add)
def reciprocal(a): return map(a, fn=np.reciprocal)
def negative(a): return map(a, fn=np.negative)
def sub(a, b): return map((a, b), fn=np.subtract)
def multiply(a, b): return map((a, b), fn=np.multiply)
def divide(a, b): return map((a, b), fn=np.divide)
def true_divide(a, b): return map((a, b), fn=np.true_divide)
def floor_divide(a, b): return map((a, b), fn=np.floor_divide)
def fmod(a, b): return map((a, b), fn=np.fmod)
def mod(a, b): return map((a, b), fn=np.mod)
def remainder(a, b): return map((a, b), fn=np.remainder)
def power(a, b): return map((a, b), fn=np.power)
def maximum(a, b): return map((a, b), np.maximum)
def minimum(a, b): return map((a, b), np.minimum)
def ln(v): return map(v, fn=np.log)
def log(v): return map(v, fn=np.log)
def exp(v): return map(v, fn=np.exp)
def square(v): return map(v, fn=np.square)
def sqrt(v): return map(v, fn=np.sqrt)
def abs(v): return map(v, fn=np.abs)


def norm_cdf(v):
  """Standard normal CDF (statistics.py:224-225)."""
  import scipy.stats
  return map(v, fn=scipy.stats.norm.cdf)


# dtype rules of the reductions: module-level functions (the reference writes lambdas at the call sites; one object per
# rule lets two DAGs built by the same calls be recognised as the same structure, expr/plan.py)
def _same_dtype(input):
  return input.dtype


def _bool_dtype(input):
  return np.bool_


def _count_dtype(input):
  return np.int64


def _sum_local(ex, data, axis):
  """mathematics.py:126-127."""
  return data.sum(axis)


def sum(x, axis=None, tile_hint=None):
  """mathematics.py:130-143."""
  return reduce(x, axis=axis, dtype_fn=_same_dtype, local_reduce_fn=_sum_local,
                accumulate_fn=np.add, tile_hint=tile_hint)


def _prod_local(ex, data, axis):
  return data.prod(axis)


def _prod_dtype_fn(input):
  """mathematics.py:150-154."""
  if input.dtype == np.int32:
    return np.dtype(np.int64)
  return input.dtype


def prod(x, axis=None, tile_hint=None):
  """mathematics.py:157-170."""
  return reduce(x, axis=axis, dtype_fn=_prod_dtype_fn, local_reduce_fn=_prod_local,
                accumulate_fn=np.multiply, tile_hint=tile_hint)


# -------------------------------------------------------------- statistics.py
def _max_local(ex, data, axis):
  """statistics.py:40 (a lambda in the reference)."""
  return data.max(axis)


def _min_local(ex, data, axis):
  """statistics.py:59."""
  return data.min(axis)


def max(x, axis=None, tile_hint=None):
  """statistics.py:26-42."""
  return reduce(x, axis=axis, dtype_fn=_same_dtype, local_reduce_fn=_max_local,
                accumulate_fn=np.maximum, tile_hint=tile_hint)


def min(x, axis=None, tile_hint=None):
  """statistics.py:45-61."""
  return reduce(x, axis=axis, dtype_fn=_same_dtype, local_reduce_fn=_min_local,
                accumulate_fn=np.minimum, tile_hint=tile_hint)


def mean(x, axis=None):
  """statistics.py:64-76.  The divisor is passed as a Python int so that fp32
  stays fp32 (the NumPy-1.x behaviour the reference relied on)."""
  if axis is None:
    return sum(x, axis) / int(np.prod(x.shape, dtype=np.int64))
  return sum(x, axis) / int(x.shape[axis])


def std(a, axis=None):
  """statistics.py:86-102."""
  a_casted = astype(a, np.float64)
  return sqrt(mean(a_casted ** 2, axis) - mean(a_casted, axis) ** 2)


# ------------------------------------------------------------------- logic.py
def _all_reducer(ex, tile, axis=None):
  return np.all(tile, axis=axis)


def all(array, axis=None):
  """logic.py:29-34."""
  return reduce(array, axis=axis, dtype_fn=_bool_dtype, local_reduce_fn=_all_reducer,
                accumulate_fn=np.logical_and)


def _any_reducer(ex, tile, axis=None):
  return np.any(tile, axis=axis)


def any(array, axis=None):
  """logic.py:41-46."""
  return reduce(array, axis=axis, dtype_fn=_bool_dtype, local_reduce_fn=_any_reducer,
                accumulate_fn=np.logical_or)


def equal(a, b): return map((a, b), fn=np.equal)
def not_equal(a, b): return map((a, b), fn=np.not_equal)
def greater(a, b): return map((a, b), fn=np.greater)
def greater_equal(a, b): return map((a, b), fn=np.greater_equal)
def less(a, b): return map((a, b), fn=np.less)
def less_equal(a, b): return map((a, b), fn=np.less_equal)
def logical_and(a, b): return map((a, b), fn=np.logical_and)
def logical_or(a, b): return map((a, b), fn=np.logical_or)
def logical_xor(a, b): return map((a, b), fn=np.logical_xor)


# ------------------------------------------------------------------ arrays.py
def _astype_mapper(t, dtype=None):
  """arrays.py:24-26."""
  return t.astype(dtype)


def astype(x, dtype):
  """arrays.py:29-40."""
  assert x is not None
  return map(x, _astype_mapper, fn_kw={'dtype': np.dtype(dtype).str})


def size(x, axis=None):
  if axis is None:
    return int(np.prod(x.shape, dtype=np.int64))
  return x.shape[axis]


# ----------------------------------------------------------------- sorting.py
def _countnonzero_local(ex, data, axis):
  """sorting.py:126-133."""
  if axis is None:
    return np.asarray(np.count_nonzero(data))
  return (data > 0).sum(axis)


def count_nonzero(array, axis=None, tile_hint=None):
  """sorting.py:136-150."""
  return reduce(array, axis, dtype_fn=_count_dtype, local_reduce_fn=_countnonzero_local,
                accumulate_fn=np.add, tile_hint=tile_hint)


def _countzero_local(ex, data, axis):
  """sorting.py:153-157."""
  if axis is None:
    return np.asarray(np.prod(ex.shape) - np.count_nonzero(data))
  return (data == 0).sum(axis)


def count_zero(array, axis=None):
  """sorting.py:160-172."""
  return reduce(array, axis, dtype_fn=_count_dtype, local_reduce_fn=_countzero_local,
                accumulate_fn=np.add)


def _arg_candidates(idx, val, best, sentinel=None):
  """The per-tile contribution to the final `min` reduce of the reference's
  argmax/argmin (sorting.py:82-85): the tile's index where its extreme equals
  the global extreme, the sentinel prod(array_shape) elsewhere."""
  return np.where(val == best, idx, sentinel)


def _argreduce_mapper1(ex, src, axis, which, val_out, state):
  ctx = context.get()
  array = src
  data = array.fetch(ex)
  distarray.tile.reject_masked((data,), 'argmax / argmin')
  dst_extent = extent.index_for_reduction(ex, axis)
  if ctx.executing:
    if axis is None:
      offset = extent.ravelled_pos(ex.ul, ex.shape)     # sorting.py:76-81: TILE shape
    else:
      offset = ex.ul[axis]
    sentinel = int(np.prod(ex.array_shape, dtype=np.int64))
    idx, val = ctx.backend.evaluate_argreduce(data, ex, axis, which, offset, sentinel)
    idx = idx.reshape(dst_extent.shape)
    val = val.reshape(dst_extent.shape)
    state[ex] = (idx, val)
  else:
    val = distarray.Absent(dst_extent.shape, array.dtype)
  val_out.update(dst_extent, val, owned=False)
  return LocalKernelResult(result=[])


def _argreduce_mapper2(ex, src, axis, val_out, idx_out, state):
  ctx = context.get()
  dst_extent = extent.index_for_reduction(ex, axis)
  best = val_out.fetch(dst_extent)
  if ctx.executing:
    idx, val = state[ex]
    sentinel = int(np.prod(ex.array_shape, dtype=np.int64))
    cand = ctx.backend.evaluate_fn(_arg_candidates, [idx, val, best.reshape(dst_extent.shape)],
                                   {'sentinel': sentinel}, dst_extent.shape)
  else:
    cand = distarray.Absent(dst_extent.shape, np.int64)
  idx_out.update(dst_extent, cand, owned=True)
  return LocalKernelResult(result=[])


class ArgReduceExpr(Expr):
  """argmax / argmin (sorting.py:88-123).  The reference computes them in three
  full passes (max-reduce, _arg_mapper index map with an int64 N x N temporary,
  min-reduce); here each tile is read ONCE by a fused (value, first-index)
  reduction and the cross-tile combine reproduces the reference's final
  `min`-of-candidate-indices exactly (first occurrence wins, sentinel
  prod(array_shape) for non-matching tiles)."""
  members = ('array', 'axis', 'which')

  def dependencies(self):
    return {'array': self.array}

  def visit(self, visitor):
    return base.expr_like(self, array=visitor.visit(self.array), axis=self.axis, which=self.which)

  def compute_shape(self):
    return tuple(extent.shape_for_reduction(self.array.shape, self.axis))

  def pretty_str(self):
    return 'ArgReduce[%d](%s, axis=%s, %s)' % (self.expr_id, 'max' if self.which == 0 else 'min',
                                               self.axis, self.array)

  def _evaluate(self, ctx, deps):
    array = deps['array']
    shape = extent.shape_for_reduction(array.shape, self.axis)
    # The extreme values are a scratch array of this node.  When every source tile reduces onto the WHOLE result (the
    # reduced axis is the one the array is cut along, or axis=None) it is ONE tile: each source tile merges its partial
    # with one launch and reads the combined values back in place, instead of a cut into num_workers pieces on both
    # ways.  Otherwise it is tiled like the result (reduce.py:117-118) and every partial meets exactly its own tile.
    targets = {extent.index_for_reduction(ex, self.axis) for ex in array.tiles}
    whole = len(targets) == 1 and len(shape) > 0
    val_out = distarray.create(shape, array.dtype, reducer=np.maximum if self.which == 0 else np.minimum,
                               tile_hint=tuple(shape) if whole else None)
    idx_out = distarray.create(shape, np.int64, reducer=np.minimum)
    state = {}
    array.foreach_tile(_argreduce_mapper1, kw=dict(src=array, axis=self.axis, which=self.which,
                                                   val_out=val_out, state=state))
    array.foreach_tile(_argreduce_mapper2, kw=dict(src=array, axis=self.axis, val_out=val_out,
                                                   idx_out=idx_out, state=state))
    return idx_out


def argmax(x, axis=None):
  """sorting.py:107-123."""
  return ArgReduceExpr(array=base.as_array(x), axis=axis, which=0)


def argmin(x, axis=None):
  """sorting.py:88-104."""
  return ArgReduceExpr(array=base.as_array(x), axis=axis, which=1)
