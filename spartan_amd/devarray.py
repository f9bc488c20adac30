"""Device arrays: views of tile-store blobs.

What a tile's `data` is on the HIP backend: a typed, strided VIEW (shape, element strides, element offset) of one
HBM allocation that belongs to the library's tile store (`sp_blob_create` / `sp_blob_destroy`,
include/spartan_hip.h) -- the counterpart of the NumPy arrays the reference's workers keep in `Worker._blobs`
(spartan/worker.py:70, spartan/blob_ctx.py:103-284).  No torch: allocation, host transfers, streams and events are
the C-ABI's; all arithmetic is a kernel launch through the backend.

A DevArray quacks like a NumPy array where the reference's per-tile functions need it to (shape / dtype / ndim /
size / T / reshape / basic indexing / astype / dot / sum / ... / operators / NumPy ufuncs and a few NumPy
functions, each lowered to the sp_* kernels), so the reference's own mappers -- written against ndarrays -- run
unchanged on HBM tiles; and it answers the handful of tensor-style questions the host framework asks
(`dim()`, `numel()`, `stride()`, `data_ptr()`, `is_contiguous()`).

View arithmetic is pure Python on integers and is unit-tested on the CPU against NumPy's own strides
(tests/test_devarray.py) through `HostStorage`, a stand-in allocation in host memory that only the tests use.
"""
import ctypes as C
import weakref

import numpy as np

from . import _hip
from ._hip import check


# ------------------------------------------------------------------------------------------------ streams / events
class Stream(object):
  """A HIP stream of the library (sp_stream_create); `handle` None is the device's default stream."""

  def __init__(self, high_priority=False, _default=False):
    self.handle = None
    if not _default:
      h = C.c_void_p()
      check(_hip.lib().sp_stream_create_priority(C.byref(h), 1 if high_priority else 0))
      self.handle = h

  @property
  def ptr(self):
    return self.handle if self.handle is not None else C.c_void_p(0)

  def synchronize(self):
    check(_hip.lib().sp_stream_synchronize(self.ptr))

  def query(self):
    done = C.c_int32(0)
    check(_hip.lib().sp_stream_query(self.ptr, C.byref(done)))
    return bool(done.value)

  def wait_event(self, event):
    check(_hip.lib().sp_stream_wait_event(self.ptr, event.h))

  def wait_stream(self, other):
    ev = Event()
    ev.record(other)
    self.wait_event(ev)

  def __del__(self):
    try:
      if self.handle is not None:
        _hip.lib().sp_stream_destroy(self.handle)
    except Exception:
      pass


DEFAULT_STREAM = Stream(_default=True)
_current = [DEFAULT_STREAM]


def current_stream():
  """The stream kernels are launched on (the device's default stream unless `use_stream` is active).  The tile
  store's contract is ONE compute stream per device: memory handed back by a dead array is reused in stream order."""
  return _current[-1]


class use_stream(object):
  """with use_stream(s): launches inside go to `s`."""

  def __init__(self, stream):
    self.stream = stream

  def __enter__(self):
    _current.append(self.stream)
    return self.stream

  def __exit__(self, *exc):
    _current.pop()


class Event(object):
  """HIP event (sp_event_*): timing on the launch stream and ordering between streams."""

  def __init__(self):
    self.h = C.c_void_p()
    check(_hip.lib().sp_event_create(C.byref(self.h)))

  def record(self, stream=None):
    check(_hip.lib().sp_event_record(self.h, (stream or current_stream()).ptr))
    return self

  def synchronize(self):
    check(_hip.lib().sp_event_synchronize(self.h))

  def elapsed_ms(self, later):
    ms = C.c_float()
    check(_hip.lib().sp_event_elapsed_ms(self.h, later.h, C.byref(ms)))
    return ms.value

  def __del__(self):
    try:
      if self.h:
        _hip.lib().sp_event_destroy(self.h)
    except Exception:
      pass


def synchronize():
  """Wait for everything enqueued on the device."""
  check(_hip.lib().sp_device_synchronize())


# ------------------------------------------------------------------------------------------------ storage
class Storage(object):
  """One allocation of the tile store: a 1-D byte blob (sp_blob_create), given back by sp_blob_destroy when the
  last view of it dies.  The store pools freed allocations by size class and reuses them in stream order."""
  __slots__ = ('handle', 'ptr', 'nbytes', '__weakref__')
  on_device = True

  def __init__(self, nbytes):
    nbytes = max(int(nbytes), 1)
    h = C.c_uint64()
    check(_hip.lib().sp_blob_create(_hip.i64_array([nbytes]), 1, _hip.SP_U8, C.byref(h)))
    p = C.c_void_p()
    check(_hip.lib().sp_blob_info(h, C.byref(p), None, None, None))
    self.handle, self.ptr, self.nbytes = h, int(p.value or 0), nbytes

  def h2d(self, byte_offset, host_ptr, nbytes):
    if nbytes:
      st = current_stream().ptr
      consumed = C.c_int32(0)
      check(_hip.lib().sp_blob_h2d_staged(self.handle, C.c_void_p(host_ptr), _hip.i64_array([byte_offset]),
                                          _hip.i64_array([byte_offset + nbytes]), st, C.byref(consumed)))
      if not consumed.value:
        check(_hip.lib().sp_stream_synchronize(st))      # the host buffer may be a temporary

  def d2h(self, byte_offset, host_ptr, nbytes):
    if nbytes:
      st = current_stream().ptr
      check(_hip.lib().sp_blob_d2h_staged(self.handle, C.c_void_p(host_ptr), _hip.i64_array([byte_offset]),
                                          _hip.i64_array([byte_offset + nbytes]), st))      # (complete on return)

  def __del__(self):
    try:
      _hip.lib().sp_blob_destroy(self.handle)
    except Exception:      # interpreter shutdown
      pass


class HostStorage(object):
  """Test stand-in for `Storage` in host memory (view arithmetic is checked against NumPy without a GPU)."""
  __slots__ = ('buf', 'ptr', 'nbytes', '__weakref__')
  on_device = False

  def __init__(self, nbytes):
    self.buf = np.zeros(max(int(nbytes), 1), np.uint8)
    self.ptr = self.buf.ctypes.data
    self.nbytes = self.buf.nbytes

  def h2d(self, byte_offset, host_ptr, nbytes):
    C.memmove(self.ptr + byte_offset, host_ptr, nbytes)

  def d2h(self, byte_offset, host_ptr, nbytes):
    C.memmove(host_ptr, self.ptr + byte_offset, nbytes)


def blob_stats():
  """(live blobs, bytes pooled for reuse) of the library's tile store."""
  live, pooled = C.c_int64(), C.c_int64()
  check(_hip.lib().sp_blob_stats(C.byref(live), C.byref(pooled)))
  return live.value, pooled.value


def trim_pool():
  """Give the pooled (free) allocations back to the driver."""
  check(_hip.lib().sp_blob_trim())


# ------------------------------------------------------------------------------------------------ view arithmetic
def dense_strides(shape):
  st, s = [], 1
  for n in reversed(shape):
    st.append(s)
    s *= max(int(n), 1)
  return tuple(reversed(st))


def _prod(shape):
  n = 1
  for s in shape:
    n *= int(s)
  return n


def _view_reshape(shape, strides, new_shape):
  """Strides of `new_shape` over the same memory, or None if the reshape needs a copy (NumPy's no-copy rule:
  walk both shapes, a group of old axes can be re-cut iff it is contiguous within itself)."""
  old = [(n, s) for n, s in zip(shape, strides) if n != 1]
  new_strides = [0] * len(new_shape)
  oi = 0
  ni = 0
  nn = len(new_shape)
  while ni < nn and new_shape[ni] == 1:
    new_strides[ni] = 1
    ni += 1
  while oi < len(old) and ni < nn:
    np_, op_ = new_shape[ni], old[oi][0]
    nj, oj = ni + 1, oi + 1
    while np_ != op_:
      if np_ < op_:
        np_ *= new_shape[nj]
        nj += 1
      else:
        op_ *= old[oj][0]
        oj += 1
    for k in range(oi, oj - 1):                    # the old group must be contiguous in itself
      if old[k][1] != old[k + 1][1] * old[k + 1][0]:
        return None
    s = old[oj - 1][1]
    for k in range(nj - 1, ni - 1, -1):
      new_strides[k] = s
      s *= new_shape[k]
    ni, oi = nj, oj
    while ni < nn and new_shape[ni] == 1:
      new_strides[ni] = 1
      ni += 1
  if oi != len(old) or ni != nn:
    return None
  return tuple(new_strides)


def _norm_shape(shape):
  if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
    shape = tuple(shape[0])
  return tuple(int(s) for s in shape)


_backend = [None]


def _be():
  """The HIP backend that runs the kernels behind a DevArray's NumPy-style methods."""
  from . import context
  if context.initialized() and getattr(context.get().backend, 'name', '') == 'hip':
    return context.get().backend
  if _backend[0] is None:
    from .backend_hip import HipBackend
    _backend[0] = HipBackend()
  return _backend[0]


class _HostCopy(object):
  """`.cpu()` of a DevArray: the host copy, with the one method callers chain on it."""

  def __init__(self, arr):
    self._arr = arr

  def numpy(self):
    return self._arr

  def __array__(self, dtype=None, copy=None):
    return self._arr if dtype is None else self._arr.astype(dtype)


class DeviceTileCannot(TypeError, AttributeError):
  """A device tile was asked for something only a host ndarray can do (an attribute or NumPy function it does not
  have, an index array, a ufunc method without a kernel).  Raised by DevArray itself and by nothing else, so that the
  backend can tell "run this user function on host copies instead" (backend_hip.call_local_fn) from an error of the
  user function's own.  A TypeError (what NumPy raises for an array-like it cannot dispatch on) and an AttributeError
  (so that hasattr() probes keep working)."""


# ------------------------------------------------------------------------------------------------ the array
class DevArray(object):
  __slots__ = ('storage', 'offset', 'shape', 'strides', 'dtype', '__weakref__')
  __array_priority__ = 1000.0
  is_cuda = True
  is_sparse_tile = False        # (asked of every operand by tile.is_sparse_blob: a class attribute, not a trip through __getattr__)

  def __init__(self, storage, offset, shape, strides, dtype):
    self.storage = storage
    self.offset = int(offset)                 # elements
    self.shape = tuple(int(s) for s in shape)
    self.strides = tuple(int(s) for s in strides)   # elements (NumPy's .strides are bytes: see stride())
    self.dtype = np.dtype(dtype)

  # -- shape questions (NumPy style and the tensor style the host framework uses) --------------------------
  ndim = property(lambda self: len(self.shape))
  size = property(lambda self: _prod(self.shape))
  itemsize = property(lambda self: self.dtype.itemsize)
  nbytes = property(lambda self: _prod(self.shape) * self.dtype.itemsize)
  device = property(lambda self: 'hip' if self.storage.on_device else 'host')

  def dim(self):
    return len(self.shape)

  def numel(self):
    return _prod(self.shape)

  def element_size(self):
    return self.dtype.itemsize

  def stride(self, axis=None):
    return self.strides if axis is None else self.strides[axis]

  def storage_offset(self):
    return self.offset

  def data_ptr(self):
    return self.storage.ptr + self.offset * self.dtype.itemsize

  def is_contiguous(self):
    if 0 in self.shape:
      return True
    expect = 1
    for n, s in zip(reversed(self.shape), reversed(self.strides)):
      if n == 1:
        continue
      if s != expect:
        return False
      expect *= n
    return True

  def __len__(self):
    if not self.shape:
      raise TypeError('len() of a 0-d array')
    return self.shape[0]

  def __repr__(self):
    return 'DevArray(shape=%s, dtype=%s, strides=%s)' % (self.shape, self.dtype, self.strides)

  __hash__ = object.__hash__

  # -- views -------------------------------------------------------------------------------------------------
  def _view(self, offset, shape, strides):
    return DevArray(self.storage, offset, shape, strides, self.dtype)

  def __getitem__(self, idx):
    """Basic indexing: integers, slices (any step), None, Ellipsis -> a view."""
    if not isinstance(idx, tuple):
      idx = (idx,)
    if any(isinstance(i, (DevArray, np.ndarray, list)) for i in idx):
      raise DeviceTileCannot('device arrays take basic indices only (integers, slices, None, ...)')
    n_real = sum(1 for i in idx if i is not None and i is not Ellipsis)
    if n_real > len(self.shape):
      raise IndexError('too many indices for a %d-d array' % len(self.shape))
    if sum(1 for i in idx if i is Ellipsis) > 1:
      raise IndexError('an index can only have a single ellipsis')
    out = []
    for i in idx:
      if i is Ellipsis:
        out.extend([slice(None)] * (len(self.shape) - n_real))
      else:
        out.append(i)
    offset, shape, strides, axis = self.offset, [], [], 0
    for i in out:
      if i is None:
        shape.append(1)
        strides.append(1)
        continue
      n, s = self.shape[axis], self.strides[axis]
      if isinstance(i, slice):
        start, stop, step = i.indices(n)
        count = len(range(start, stop, step))
        offset += start * s if count else 0
        shape.append(count)
        strides.append(s * step)
      else:
        i = int(i)
        if i < -n or i >= n:
          raise IndexError('index %d is out of bounds for axis %d with size %d' % (i, axis, n))
        offset += (i % n if n else 0) * s
      axis += 1
    shape.extend(self.shape[axis:])
    strides.extend(self.strides[axis:])
    return self._view(offset, shape, strides)

  def reshape(self, *shape):
    shape = list(_norm_shape(shape))
    total = _prod(self.shape)
    if shape.count(-1) == 1:
      known = _prod([s for s in shape if s != -1])
      shape[shape.index(-1)] = total // known if known else 0
    if _prod(shape) != total:
      raise ValueError('cannot reshape array of size %d into shape %s' % (total, tuple(shape)))
    if total == 0:
      return self._view(self.offset, shape, dense_strides(shape))
    st = _view_reshape(self.shape, self.strides, shape)
    if st is None:
      dense = self.contiguous()
      return dense._view(dense.offset, shape, dense_strides(shape))
    return self._view(self.offset, shape, st)

  view = reshape

  def permute(self, *axes):
    axes = _norm_shape(axes)
    nd = len(self.shape)
    axes = tuple(a % nd for a in axes)
    if sorted(axes) != list(range(nd)):
      raise ValueError('axes do not match the array')
    return self._view(self.offset, [self.shape[a] for a in axes], [self.strides[a] for a in axes])

  def transpose(self, *axes):
    if not axes or axes == (None,):
      return self.permute(*reversed(range(len(self.shape))))
    return self.permute(*axes)

  def t(self):
    return self.transpose()

  T = property(transpose)

  def movedim(self, src, dst):
    nd = len(self.shape)
    src, dst = src % nd, dst % nd
    order = [a for a in range(nd) if a != src]
    order.insert(dst, src)
    return self.permute(*order)

  def swapaxes(self, a, b):
    order = list(range(len(self.shape)))
    order[a], order[b] = order[b], order[a]
    return self.permute(*order)

  def ravel(self):
    return self.reshape(-1)

  def diagonal(self, offset=0, axis1=0, axis2=1):
    """ndarray.diagonal: a view; the diagonal becomes the last axis."""
    nd = len(self.shape)
    if nd < 2:
      raise ValueError('diag requires an array of at least two dimensions')
    axis1, axis2 = axis1 % nd, axis2 % nd
    if axis1 == axis2:
      raise ValueError('axis1 and axis2 cannot be the same')
    n1, n2, s1, s2 = self.shape[axis1], self.shape[axis2], self.strides[axis1], self.strides[axis2]
    start = self.offset + (offset * s2 if offset >= 0 else -offset * s1)
    count = max(0, min(n1, n2 - offset) if offset >= 0 else min(n1 + offset, n2))
    keep = [i for i in range(nd) if i not in (axis1, axis2)]
    return self._view(start if count else self.offset, [self.shape[i] for i in keep] + [count],
                      [self.strides[i] for i in keep] + [s1 + s2])

  flatten = ravel

  def squeeze(self, axis=None):
    keep = [i for i, n in enumerate(self.shape) if not (n == 1 and (axis is None or i == axis % len(self.shape)))]
    return self._view(self.offset, [self.shape[i] for i in keep], [self.strides[i] for i in keep])

  # -- copies ------------------------------------------------------------------------------------------------
  def contiguous(self):
    return self if self.is_contiguous() else self.copy()

  def copy(self):
    if not self.storage.on_device:           # (tests of the view arithmetic)
      return from_numpy(self.numpy(), storage_cls=HostStorage)
    return _be().copy(self)

  clone = copy

  def numpy(self):
    """The values as a NumPy array (device -> host, synchronises the launch stream)."""
    if self.storage.on_device:
      src = self.contiguous()
      out = np.empty(self.shape, self.dtype)
      src.storage.d2h(src.offset * self.dtype.itemsize, out.ctypes.data, out.nbytes)
      return out
    base = self.storage.buf.view(self.dtype) if self.storage.nbytes % self.dtype.itemsize == 0 else \
        self.storage.buf[:self.storage.nbytes // self.dtype.itemsize * self.dtype.itemsize].view(self.dtype)
    return np.lib.stride_tricks.as_strided(base[self.offset:], self.shape,
                                           [s * self.dtype.itemsize for s in self.strides]).copy()

  def cpu(self):
    return _HostCopy(self.numpy())

  def upload(self, arr):
    """Overwrite this (contiguous) array with host data of the same shape."""
    if not self.is_contiguous():
      raise ValueError('upload() needs a contiguous device array')
    src = np.ascontiguousarray(np.asarray(arr).reshape(self.shape), dtype=self.dtype)
    self.storage.h2d(self.offset * self.dtype.itemsize, src.ctypes.data, src.nbytes)
    return self

  def __array__(self, dtype=None, copy=None):
    out = self.numpy()
    return out if dtype is None else out.astype(dtype)

  @property
  def __cuda_array_interface__(self):
    """Zero-copy hand-over to other GPU libraries (version 3 of the interface: CuPy, Numba, torch.as_tensor);
    HIP devices are exposed to those libraries under this name as well."""
    if not self.storage.on_device:
      raise AttributeError('not a device array')
    item = self.dtype.itemsize
    return {'shape': self.shape, 'typestr': self.dtype.str, 'data': (self.data_ptr(), False), 'version': 3,
            'strides': None if self.is_contiguous() else tuple(s * item for s in self.strides)}

  def item(self):
    if _prod(self.shape) != 1:
      raise ValueError('can only convert an array of size 1 to a Python scalar')
    return self.numpy().reshape(()).item()

  def tolist(self):
    return self.numpy().tolist()

  def __bool__(self):
    return bool(self.item())

  def __float__(self):
    return float(self.item())

  def __int__(self):
    return int(self.item())

  def __index__(self):
    if self.dtype.kind not in 'iu':
      raise TypeError('only integer arrays convert to an index')
    return int(self.item())

  # -- NumPy-style compute: every method is a kernel launch through the backend ------------------------------
  def astype(self, dtype, copy=True):
    out = _be().astype(self, dtype)
    return out.copy() if (copy and out is self) else out

  def dot(self, other):
    return _be().dot(self, _be()._as_device(other) if isinstance(other, np.ndarray) else other)

  __matmul__ = dot

  def fill(self, value):
    _be().assign_box(self, tuple(slice(0, n) for n in self.shape), value)

  def __setitem__(self, idx, value):
    view = self[idx]
    _be().assign_box(view, tuple(slice(0, n) for n in view.shape), value)

  def _reduce(self, red_op, axis, keepdims=False, dtype=None):
    be = _be()
    src = self if dtype is None else be.astype(self, dtype)
    if axis is None:
      out = be.evaluate_reduce_tensor(src.contiguous(), red_op)
      return out.reshape((1,) * len(self.shape)) if keepdims else out
    nd = len(self.shape)
    axes = sorted(set(a % nd for a in (axis if isinstance(axis, (tuple, list)) else (axis,))), reverse=True)
    for a in axes:
      src = be.reduce_axis(src, red_op, a)
    if keepdims:
      shape = list(self.shape)
      for a in axes:
        shape[a] = 1
      src = src.reshape(shape)
    return src

  def _accumulator(self, dtype, what):
    """NumPy's promotion for sum / prod: bool and small signed integers accumulate in int64; unsigned ones in uint64,
    which has no device type (include/spartan_hip.h: sp_dtype) -- said with the sentinel instead of answering int64."""
    if dtype is not None or self.dtype.kind not in 'bui' or self.dtype.itemsize >= 8:
      return dtype
    if self.dtype.kind == 'u':
      raise DeviceTileCannot('%s of a %s tile accumulates in uint64, which device tiles do not have; '
                             'pass dtype=np.int64' % (what, self.dtype))
    return np.int64

  def sum(self, axis=None, dtype=None, keepdims=False, **kw):
    return self._reduce('SUM', axis, keepdims, self._accumulator(dtype, 'sum'))

  def prod(self, axis=None, dtype=None, keepdims=False, **kw):
    return self._reduce('PROD', axis, keepdims, self._accumulator(dtype, 'prod'))

  def max(self, axis=None, keepdims=False, **kw):
    return self._reduce('MAX', axis, keepdims)

  def min(self, axis=None, keepdims=False, **kw):
    return self._reduce('MIN', axis, keepdims)

  def all(self, axis=None, keepdims=False, **kw):
    return self._reduce('AND', axis, keepdims)

  def any(self, axis=None, keepdims=False, **kw):
    return self._reduce('OR', axis, keepdims)

  def mean(self, axis=None, dtype=None, keepdims=False, **kw):
    src = self if self.dtype.kind == 'f' else self.astype(np.float64)
    total = src.sum(axis, dtype, keepdims)
    count = _prod(self.shape) // max(_prod(total.shape), 1)
    return total / total.dtype.type(count)

  def _arg(self, which, axis):
    be = _be()
    src = self.contiguous()
    if axis is None:
      idx, _ = be.evaluate_argreduce(src.reshape(-1), None, None, which, 0, 0)
      return idx.reshape(())
    nd = len(self.shape)
    axis %= nd
    idx, _ = be.evaluate_argreduce(src, None, axis, which, 0, 0)
    return idx.reshape(self.shape[:axis] + self.shape[axis + 1:])

  def argmax(self, axis=None, **kw):
    return self._arg(0, axis)

  def argmin(self, axis=None, **kw):
    return self._arg(1, axis)

  # -- operators and NumPy's dispatch protocols ---------------------------------------------------------------
  def _ufunc(self, ufunc, *args):
    return _be().evaluate_fn(ufunc, list(args), {}, None)

  def __array_ufunc__(self, ufunc, method, *inputs, **kw):
    out = kw.pop('out', None)
    if method == '__call__' and not kw:
      res = _be().evaluate_fn(ufunc, list(inputs), {}, None)
      if out is not None:
        dst = out[0] if isinstance(out, tuple) else out
        _be().assign_box(dst, tuple(slice(0, n) for n in dst.shape), res)
        return dst
      return res
    if method == 'reduce' and ufunc in _UFUNC_REDUCE:
      axis = kw.get('axis', 0)
      return inputs[0]._reduce(_UFUNC_REDUCE[ufunc], axis, kw.get('keepdims', False), kw.get('dtype'))
    raise DeviceTileCannot('no kernel behind %s.%s(%s) on a device array' % (ufunc.__name__, method, ', '.join(sorted(kw))))

  def __array_function__(self, func, types, args, kwargs):
    impl = _NP_FUNCTIONS.get(func)
    if impl is None:
      raise DeviceTileCannot('numpy.%s is not implemented for device arrays' % getattr(func, '__name__', func))
    return impl(*args, **kwargs)

  def __getattr__(self, name):
    # (only reached for names the class does not define)  An ndarray attribute that has no device form sends a
    # user's tile function to the host path; anything else is a mistake in the caller and says so.
    if hasattr(np.ndarray, name):
      raise DeviceTileCannot('ndarray.%s is not implemented for device arrays' % name)
    raise AttributeError('%s object has no attribute %r' % (type(self).__name__, name))


_UFUNC_REDUCE = {np.add: 'SUM', np.multiply: 'PROD', np.maximum: 'MAX', np.minimum: 'MIN',
                 np.logical_and: 'AND', np.logical_or: 'OR'}


def _binary(ufunc, swap=False):
  def op(self, other):
    if isinstance(other, (list, tuple)):
      other = np.asarray(other)
    return _be().evaluate_fn(ufunc, [other, self] if swap else [self, other], {}, None)
  return op


def _unary(ufunc):
  def op(self):
    return _be().evaluate_fn(ufunc, [self], {}, None)
  return op


for _name, _uf in dict(add=np.add, sub=np.subtract, mul=np.multiply, truediv=np.true_divide,
                       floordiv=np.floor_divide, mod=np.remainder, pow=np.power, lt=np.less, le=np.less_equal,
                       gt=np.greater, ge=np.greater_equal, eq=np.equal, ne=np.not_equal,
                       **{'and': np.logical_and, 'or': np.logical_or, 'xor': np.logical_xor}).items():
  setattr(DevArray, '__%s__' % _name, _binary(_uf))
  if _name not in ('lt', 'le', 'gt', 'ge', 'eq', 'ne'):
    setattr(DevArray, '__r%s__' % _name, _binary(_uf, swap=True))
DevArray.__neg__ = _unary(np.negative)
DevArray.__abs__ = _unary(np.abs)
DevArray.__pos__ = lambda self: self
DevArray.__invert__ = _unary(np.logical_not)


# -- the NumPy functions user mappers call on tiles -------------------------------------------------------------
def _np_where(cond, a=None, b=None):
  if a is None and b is None:
    raise DeviceTileCannot('np.where(cond) (index form) on a device array')
  return _be().evaluate_fn(np.where, [cond, a, b], {}, None)


def _np_concatenate(arrays, axis=0, **kw):
  be = _be()
  arrays = [be._as_device(a) if isinstance(a, np.ndarray) else a for a in arrays]
  out = arrays[0]
  for nxt in arrays[1:]:
    out = be.concat(out, nxt, axis=axis)
  return out


def _np_bincount(x, weights=None, minlength=0):
  if weights is not None:
    raise DeviceTileCannot('np.bincount(weights=...) on a device array')
  be = _be()
  k = max(int(minlength), int(x.max().item()) + 1 if x.size else 0)
  return be.bincount(x, k)


def _np_zeros_like(a, dtype=None, **kw):
  return zeros(a.shape, dtype or a.dtype)


def _np_empty_like(a, dtype=None, **kw):
  return empty(a.shape, dtype or a.dtype)


def _np_ones_like(a, dtype=None, **kw):
  return full(a.shape, 1, dtype or a.dtype)


def _np_full_like(a, fill_value, dtype=None, **kw):
  return full(a.shape, fill_value, dtype or a.dtype)


_NP_FUNCTIONS = {
    np.dot: lambda a, b, out=None: (a if isinstance(a, DevArray) else _be()._as_device(np.asarray(a))).dot(b),
    np.matmul: lambda a, b, **kw: (a if isinstance(a, DevArray) else _be()._as_device(np.asarray(a))).dot(b),
    np.sum: lambda a, axis=None, dtype=None, out=None, keepdims=False, **kw: a.sum(axis, dtype, keepdims),
    np.prod: lambda a, axis=None, dtype=None, out=None, keepdims=False, **kw: a.prod(axis, dtype, keepdims),
    np.max: lambda a, axis=None, out=None, keepdims=False, **kw: a.max(axis, keepdims),
    np.min: lambda a, axis=None, out=None, keepdims=False, **kw: a.min(axis, keepdims),
    np.amax: lambda a, axis=None, out=None, keepdims=False, **kw: a.max(axis, keepdims),
    np.amin: lambda a, axis=None, out=None, keepdims=False, **kw: a.min(axis, keepdims),
    np.mean: lambda a, axis=None, dtype=None, out=None, keepdims=False, **kw: a.mean(axis, dtype, keepdims),
    np.all: lambda a, axis=None, out=None, keepdims=False, **kw: a.all(axis, keepdims),
    np.any: lambda a, axis=None, out=None, keepdims=False, **kw: a.any(axis, keepdims),
    np.argmax: lambda a, axis=None, out=None, **kw: a.argmax(axis),
    np.argmin: lambda a, axis=None, out=None, **kw: a.argmin(axis),
    np.where: _np_where,
    np.concatenate: _np_concatenate,
    np.bincount: _np_bincount,
    np.transpose: lambda a, axes=None: a.transpose(*(axes or ())),
    np.reshape: lambda a, shape=None, *args, **kw: a.reshape(shape if shape is not None else kw.get('newshape')),
    np.ravel: lambda a, **kw: a.ravel(),
    np.squeeze: lambda a, axis=None: a.squeeze(axis),
    np.shape: lambda a: a.shape,
    np.ndim: lambda a: a.ndim,
    np.size: lambda a, axis=None: a.size if axis is None else a.shape[axis],
    np.copy: lambda a, **kw: a.copy(),
    np.ascontiguousarray: lambda a, dtype=None, **kw: (a if dtype is None else a.astype(dtype, copy=False)).contiguous(),
    np.zeros_like: _np_zeros_like,
    np.empty_like: _np_empty_like,
    np.ones_like: _np_ones_like,
    np.full_like: _np_full_like,
    np.square: lambda a, **kw: a * a,
}


# ------------------------------------------------------------------------------------------------ constructors
_storage_cls = [Storage]       # (jit_seed.py swaps in HostStorage to drive the launch path on a machine without a GPU)


def empty(shape, dtype, storage_cls=None):
  shape = _norm_shape((shape,)) if not isinstance(shape, (tuple, list)) else tuple(int(s) for s in shape)
  dtype = np.dtype(dtype)
  _hip.sp_dtype(dtype)          # raises for dtypes the tile kernels do not take
  st = (storage_cls or _storage_cls[0])(_prod(shape) * dtype.itemsize)
  return DevArray(st, 0, shape, dense_strides(shape), dtype)


def zeros(shape, dtype):
  out = empty(shape, dtype)
  if out.nbytes:
    check(_hip.lib().sp_memset(C.c_void_p(out.data_ptr()), 0, out.nbytes, current_stream().ptr))
  return out


def full(shape, value, dtype):
  out = empty(shape, dtype)
  if out.nbytes:
    out.fill(value)
  return out


def from_numpy(arr, storage_cls=None):
  """Host -> a new device array (one contiguous transfer)."""
  arr = np.asarray(arr)
  src = arr if arr.flags['C_CONTIGUOUS'] else arr.copy(order='C')     # (ascontiguousarray would make 0-d -> 1-d)
  out = empty(src.shape, src.dtype, storage_cls)
  out.storage.h2d(0, src.ctypes.data, src.nbytes)
  return out


def is_devarray(x):
  return isinstance(x, DevArray)


def keep_alive_until(event, arrays):
  """Hold `arrays` (their allocations) until `event` has completed: for memory a side stream still works on when
  its last Python reference goes away -- the store reuses freed memory in the order of the COMPUTE stream only."""
  _pending.append((event, list(arrays)))
  if len(_pending) > 64:
    reap()


_pending = []


def reap(block=False):
  keep = []
  for ev, arrays in _pending:
    if block:
      ev.synchronize()
    elif not _event_done(ev):
      keep.append((ev, arrays))
  _pending[:] = keep


def _event_done(ev):
  done = C.c_int32(0)
  check(_hip.lib().sp_event_query(ev.h, C.byref(done)))
  return bool(done.value)
