"""The HIP tile-kernel backend: the product path.

Implements the backend seam (the role ParakeetExpr.evaluate plays in the
reference, spartan/expr/operator/local.py:187-209) on top of the C-ABI
(include/spartan_hip.h) for tile blobs that live in HBM in the library's tile store
(spartan_amd/devarray.py: DevArray views of sp_blob_* allocations; streams and events are
the C-ABI's too -- no torch on this path).  Constructing it without a GPU or without the
built library raises: there is no CPU fallback in the product path.
"""
import collections
import os
import time

import ctypes

import numpy as np

from . import _hip, kernels, lower
from . import devarray as D
from . import sparse as sparse_mod
from .array import distarray, tile
from .expr.local import LocalMapLocationExpr, FnCallExpr, LocalInput
from .program import ProgramTooLarge, class_of, join_class

_REDUCER_NAMES = {
    None: 'NONE', np.add: 'ADD', np.multiply: 'MUL', np.maximum: 'MAX', np.minimum: 'MIN',
    np.logical_and: 'AND', np.logical_or: 'OR',
}


try:
  from xxhash import xxh3_64_intdigest as _digest     # ~10 GB/s: 0.05 ms for the 2 MB of k-means centers
except ImportError:                                     # pragma: no cover
  from zlib import crc32 as _digest


def _content_stamp(view):
  """A number that changes when the bytes of a (small) driver-side operand do."""
  if not view.flags['C_CONTIGUOUS']:
    view = np.ascontiguousarray(view)
  return (view.shape, view.dtype.str, _digest(memoryview(view).cast('B')))


class _FixedPoints(object):
  def __init__(self, backend):
    self.backend = backend

  def __enter__(self):
    self.outer = self.backend._fixed_points
    if self.outer is None:
      self.backend._fixed_points = {}
      self.backend._fit_labels = collections.OrderedDict()
    return self

  def __exit__(self, *exc):
    self.backend._fixed_points = self.outer
    if self.outer is None:
      self.backend._fit_labels = None


class _HostCopyLater(object):
  """What HipBackend.to_numpy_later hands out: get() waits for the copy (once) and returns the array."""

  def __init__(self, tensor, host, free, done):
    self._tensor, self._host, self._free, self._done = tensor, host, free, done      # (the tensor outlives the copy)
    self._value = None

  def get(self):
    if self._value is None:
      self._done.synchronize()
      t = self._tensor
      self._value = np.frombuffer(ctypes.string_at(self._host, t.nbytes), np.dtype(t.dtype)).reshape(tuple(t.shape)).copy()
      self._release()
    return self._value

  def _release(self):
    if self._host is not None:
      self._free.append(self._host)
      self._host = self._tensor = None

  def __del__(self):
    # dropped unread: the buffer may be handed out again only once the copy into it has landed
    try:
      if self._host is not None:
        self._done.synchronize()
        self._release()
    except Exception:      # noqa: BLE001  (interpreter shutdown)
      pass


def _no_copy(t):
  raise lower.NotLowerable('an operand the kernels cannot address in place')


class HipBackend(object):
  name = 'hip'

  def __init__(self, device=None):
    lib = _hip.lib()  # raises HipLibraryMissing if the extension has not been built
    count = ctypes.c_int(0)
    if lib.sp_device_count(ctypes.byref(count)) != 0 or count.value < 1:
      raise _hip.HipError('the HIP tile backend needs an AMD GPU (no HIP device is visible); '
                          'there is no CPU fallback')
    self.device = 'hip'
    self._np_cache = collections.OrderedDict()   # bounded: iterative drivers pass a new array every step
    self._side_copies, self._pinned_free = None, {}     # to_numpy_later
    self._fixed_points = None                            # fixed_points()
    # inside a fit: what is known about label tiles -- {address of a tile: (the tile, the int64 labels it was cast
    # from or None, their np.bincount or None)}; the last few entries only, each holding its arrays alive (an
    # address cannot be re-used for something else while its entry is there)
    self._fit_labels = None
    self.launches = 0
    self.gemms = 0            # gemm_into launches (the K-split tests count them)
    self.host_round_trips = 0  # local functions that had to run on host copies of their tiles (call_local_fn)
    self._warned_host = set()
    self.gemm_events = None   # set to [] to record (start, stop) HIP events around every GEMM launch
    self._rng_seed = (int(time.time() * 100000) + os.getpid()) & (2**63 - 1)   # srandom.py:23-35: from the clock
    self._rng_offset = 0
    # lowered programs of fused operator trees seen before: (operator structure, operand types, tile shape) ->
    # program + operand order (see _lowering_key); a hit skips type inference and code emission, not the launch
    self._lowered = collections.OrderedDict()
    self.lowering_hits = 0
    lower.warm_result_dtypes()
    # the code objects that travel with the tree (csrc/jit_seed) are loaded while the host builds its first
    # expressions: ctypes drops the GIL for the call, a seeded program then starts specialised at once
    dev = ctypes.c_int32(0)
    if os.environ.get('SP_JIT_PRELOAD', '1') != '0' and lib.sp_get_device(ctypes.byref(dev)) == 0:
      import atexit
      import threading
      t = threading.Thread(target=lib.sp_jit_preload, args=(dev.value,), daemon=True, name='sp_jit_preload')
      t.start()
      atexit.register(t.join)     # (never inside hipModuleLoadData when the runtime is torn down)

  # -- memory -------------------------------------------------------------------
  def empty(self, shape, dtype):
    return D.empty(tuple(int(s) for s in shape), dtype)

  def zeros(self, shape, dtype):
    return D.zeros(tuple(int(s) for s in shape), dtype)

  def from_numpy(self, arr):
    return D.from_numpy(arr)

  def to_numpy(self, t):
    if isinstance(t, np.ndarray):
      return t
    if isinstance(t, tile.EmptyBlob):
      return np.zeros(t.shape, t.dtype)
    return t.numpy() if isinstance(t, D.DevArray) else np.asarray(t.cpu().numpy())

  def to_numpy_later(self, t):
    """A handle whose get() is the host copy of the (small) device tensor `t` AS IT IS ONCE EVERYTHING ENQUEUED SO FAR
    HAS RUN -- without making the host wait now: the copy is put on a side stream behind an event into a pinned
    buffer, the compute stream goes on (sp_copy_d2h_async).  For values a driver only CHECKS (the cluster counts of
    a k-means iteration), one iteration later."""
    t = self.contiguous(t)
    ready = D.Event().record()
    if self._side_copies is None:
      self._side_copies = D.Stream()
    side = self._side_copies
    side.wait_event(ready)
    # pinned landing buffers are kept (hipHostMalloc maps memory, hipHostFree waits for the device): free lists by size
    size = max(256, 1 << (max(t.nbytes, 1) - 1).bit_length())
    free = self._pinned_free.setdefault(size, [])
    if free:
      host = free.pop()
    else:
      host = ctypes.c_void_p()
      _hip.check(_hip.lib().sp_pinned_alloc(size, ctypes.byref(host)))
    _hip.check(_hip.lib().sp_copy_d2h_async(host, ctypes.c_void_p(t.data_ptr()), t.nbytes, side.ptr))
    return _HostCopyLater(t, host, free, D.Event().record(side))

  def release_pinned(self):
    """Give the pooled landing buffers of to_numpy_later back (waits for the device: hipHostFree)."""
    for free in self._pinned_free.values():
      while free:
        _hip.lib().sp_pinned_free(free.pop())

  def dtype_of(self, t):
    if isinstance(t, sparse_mod.CsrTile):
      return t.dtype
    if isinstance(t, (tile.EmptyBlob, distarray.Absent, np.ndarray, np.generic)):
      return np.dtype(t.dtype)
    if isinstance(t, D.DevArray):
      return t.dtype
    if hasattr(t, 'data_ptr'):
      return kernels.np_dtype_of(t)
    return np.asarray(t).dtype

  def same_dtype(self, t, dtype):
    return self.dtype_of(t) == np.dtype(dtype)

  def contiguous(self, t):
    return t if t.is_contiguous() else self.copy(t)

  def copy(self, t):
    out = D.empty(tuple(t.shape), self.dtype_of(t))
    if t.numel():
      self.paste(out, tuple(slice(0, n) for n in t.shape), t)
    return out

  def same_memory(self, a, b):
    """Do the two device arrays start at the same HBM address (one is the other, or a view of all of it)?"""
    return hasattr(a, 'data_ptr') and hasattr(b, 'data_ptr') and a.data_ptr() == b.data_ptr()

  def astype(self, t, dtype):
    dtype = np.dtype(dtype)
    if self.dtype_of(t) == dtype:
      return t
    v = lower.cast(lower.V('tensor', dtype=self.dtype_of(t), shape=tuple(t.shape), tensor=t), dtype)
    return self._run_map(v, tuple(t.shape))

  def _note_labels(self, tile_, origin=None, counts=None):
    known = self._fit_labels
    key = tile_.data_ptr()
    old = known.pop(key, None)
    if old is not None and self._same_view(old[0], tile_):
      origin = origin if origin is not None else old[1]
      counts = counts if counts is not None else old[2]
    known[key] = (tile_, origin, counts)
    # an iteration notes two tiles per worker (the int64 labels and their float32 image): two iterations' worth stay
    from . import context
    cap = 4 * (context.get().num_workers if context.initialized() else 1) + 4
    while len(known) > cap:
      known.popitem(last=False)

  @staticmethod
  def _same_view(a, b):
    return a.shape == b.shape and a.strides == b.strides and a.dtype == b.dtype and a.data_ptr() == b.data_ptr()

  def _known_labels(self, labels):
    """(int64 labels, their counts or None) of a label tile this fit produced, or (None, None)."""
    known = self._fit_labels
    if known is None or not isinstance(labels, D.DevArray):
      return None, None
    hit = known.get(labels.data_ptr())
    if hit is None or not self._same_view(hit[0], labels):
      return None, None
    tile_, origin, counts = hit
    if origin is None:
      return (tile_ if tile_.dtype == np.int64 else None), counts
    # counts noted under the origin's own entry (segment_sum sees the int64 array) count for the cast tile too
    if counts is None:
      o = known.get(origin.data_ptr())
      counts = o[2] if o is not None and self._same_view(o[0], origin) else None
    return origin, counts

  def _take_counts(self, labels):
    """The counts this fit's segment_sum noted for these labels, handed over ONCE (the caller owns the array: a
    target tile adopts it and may add to it in place)."""
    origin, counts = self._known_labels(labels)
    if counts is None:
      return None
    known = self._fit_labels
    for t in (labels, origin):
      if t is not None:
        e = known.get(t.data_ptr())
        if e is not None and e[2] is counts:
          known[t.data_ptr()] = (e[0], e[1], None)
    return counts

  def cached_numpy(self, arr, slices):
    """A (slice of a) driver-side NumPy operand in HBM (the reference pickles it into every RunKernelReq,
    dot.py:172-187).  All tiles of ONE top-level evaluation share one upload, found by object identity alone (the
    driver does not run while an evaluation does).  Across evaluations the driver may have updated the array in place
    (`w -= alpha * grad`), so the copy is re-used only while a content stamp says the bytes are the same; the stamp
    is taken when the same object comes back in a LATER evaluation, not at the first upload -- an iterative driver
    hands over a new array every step (`w = w - g * alpha`, `centers = sums / counts`) and never pays for a hash --
    and arrays above 4 MiB are not hashed at all: one upload per evaluation."""
    if type(arr) is D.DevArray:        # a driver loop that keeps its operand on the device (examples/lreg.py)
      return arr[tuple(slices)]
    from . import context
    ctx = context.get() if context.initialized() else None
    epoch = ctx.eval_epoch if ctx is not None and ctx.eval_depth > 0 else None     # (a direct backend call: no epoch)
    # (one key per box however it is spelt: missing trailing axes, None bounds)
    key = (id(arr), tuple([(s.start or 0, arr.shape[i] if s.stop is None else s.stop) for i, s in enumerate(slices)] +
                          [(0, n) for n in arr.shape[len(slices):]]))
    hit = self._np_cache.get(key)
    if hit is not None and hit[0] is arr:
      if epoch is not None and hit[3] == epoch:
        self._np_cache.move_to_end(key)
        return hit[1]
      stamp = _content_stamp(arr[slices]) if arr.nbytes <= (1 << 22) else None
      if stamp is not None and hit[2] == stamp:
        self._np_cache[key] = (arr, hit[1], stamp, epoch)
        self._np_cache.move_to_end(key)
        return hit[1]
    else:
      stamp = None
    hit = (arr, self.from_numpy(arr[slices]), stamp, epoch)
    self._np_cache[key] = hit
    self._np_cache.move_to_end(key)
    while len(self._np_cache) > 64:
      self._np_cache.popitem(last=False)
    return hit[1]

  def reducer_name(self, fn):
    try:
      return _REDUCER_NAMES[fn]
    except (KeyError, TypeError):
      raise lower.NotLowerable('reducer %r has no GPU combine kernel (supported: None, np.add, '
                               'np.multiply, np.maximum, np.minimum, np.logical_and, np.logical_or)' % (fn,))

  # -- strided views ----------------------------------------------------------------
  def paste(self, dst, dst_slices, src):
    """dst[dst_slices] = src (strided box copy in HBM)."""
    view = dst[dst_slices] if dst.dim() else dst
    if tuple(view.shape) != tuple(src.shape):
      src = src.reshape(view.shape)
    if view.numel() == 0:
      return
    if self.dtype_of(view) != self.dtype_of(src):
      raise _hip.HipError('paste: dtype mismatch %s vs %s' % (self.dtype_of(view), self.dtype_of(src)))
    nd = view.dim()
    base = view.storage_offset() - dst.storage_offset()
    if nd > _hip.SP_MAX_DIMS:
      # the copy kernel walks SP_MAX_DIMS dimensions: one launch per index of the leading ones
      lead = nd - _hip.SP_MAX_DIMS
      for index in np.ndindex(*view.shape[:lead]):
        self.launches += 1
        kernels.slice_copy(dst, base + sum(i * st for i, st in zip(index, view.stride()[:lead])), view.stride()[lead:],
                           src, sum(i * st for i, st in zip(index, src.stride()[:lead])), src.stride()[lead:],
                           view.shape[lead:])
      return
    self.launches += 1
    kernels.slice_copy(dst, base, view.stride(), src, 0, src.stride(), view.shape)

  # -- combine --------------------------------------------------------------------
  def update_box(self, dst, ul, lr, src, reducer, mask_mode, mask):
    self.launches += 1
    src = self.contiguous(src)
    if not dst.is_contiguous():
      raise _hip.HipError('update target must be a dense tile')
    kernels.update(dst, ul, lr, src, self.reducer_name(reducer), mask_mode, mask)
    known = self._fit_labels
    if known is not None:
      if (reducer is None and dst.dtype == np.float32 and src.dtype == np.int64 and tuple(src.shape) == tuple(dst.shape)
          and tuple(lr) == tuple(dst.shape) and not any(ul) and src.data_ptr() in known
          and self._same_view(known[src.data_ptr()][0], src)):
        # the int64 labels of a fit's assignment written over a whole float32 target tile (map2 targets take the
        # points' dtype, map.py:317-318): remember which labels these floats ARE -- exact, nearest_center registers
        # labels below 2^24 only -- so that the joins that read the target back need not convert them again
        self._note_labels(dst, origin=src)
      elif dst.data_ptr() in known:
        known.pop(dst.data_ptr(), None)        # anything else written into a known label tile: forget it

  def mask_all_set(self, mask, subslice):
    return bool(self.evaluate_reduce_tensor(mask[subslice], 'AND').item())

  def mask_first(self, mask):
    return bool(mask.reshape(-1)[0].item())

  # -- kernels from LocalExpr trees ---------------------------------------------------
  def _prepare(self, v):
    """Upload NumPy operands of a V tree."""
    if v.kind == 'tensor' and isinstance(v.tensor, np.ndarray):
      v.tensor = self.cached_numpy(v.tensor, tuple(slice(0, n) for n in v.tensor.shape))
    for a in v.args:
      self._prepare(a)

  def _carve(self, root):
    """The lowered tree does not fit one kernel (operands / registers / instructions, include/spartan_hip.h
    SP_MAX_INPUTS, SP_NREG, SP_MAX_INSTR): run the largest proper sub-expression that certainly fits as a launch
    of its own and put its result in its place.  Every call removes at least one operator, so repeated carving
    ends with a tree that fits.  Returns False when nothing can be carved (a single operator over leaves)."""
    best = [None, 0]            # (parent, index in parent.args), operators in the subtree

    def survey(v, parent, idx):
      """(operators, distinct tensor operands) of the subtree; remembers the biggest fitting proper subtree."""
      if v.kind != 'op':
        # (a position-dependent leaf belongs to the index space of the WHOLE program: never carved off)
        return 0, ({id(v.tensor)} if v.kind == 'tensor' else ({'iota'} if v.kind == 'iota' else set()))
      ops, tensors = 1, set()
      for i, a in enumerate(v.args):
        o, t = survey(a, v, i)
        ops += o
        tensors |= t
      # conservative bounds: every operator may need a result register and a dtype normalisation
      fits = 'iota' not in tensors and len(tensors) <= 4 and 2 * ops + len(tensors) <= 24
      if parent is not None and fits and v.op not in ('FILL',) and ops > best[1]:
        best[0], best[1] = (parent, idx), ops
      return ops, tensors
    survey(root, None, 0)
    if best[0] is None:
      return False
    parent, idx = best[0]
    sub = parent.args[idx]
    piece = self._run_map(sub, sub.shape, sub.dtype)
    parent.args[idx] = lower.V('tensor', dtype=sub.dtype, shape=sub.shape, tensor=piece)
    return True

  def _emit_map(self, root, out_shape, out_dtype, launches_allowed=True):
    """(program, operands, output dtype) of a lowered tree.  With launches_allowed (a tile is being evaluated) host
    operands are uploaded, views the kernels cannot address are copied and a tree too large for one program is carved
    into several launches; without (prelower_map: nothing may run yet) any of those raises NotLowerable."""
    if launches_allowed:
      self._prepare(root)
    if root.kind == 'const':
      root = lower.V('op', dtype=root.dtype, shape=(), op='FILL', args=[root])
    out_dtype = np.dtype(out_dtype or root.dtype)
    while True:
      cls = lower.choose_class(root, [class_of(out_dtype)] if out_dtype != np.bool_ else [])
      em = lower.Emitter(cls, out_shape, self.contiguous if launches_allowed else _no_copy)
      try:
        prog, tensors = em.finish(root, out_dtype)
        return prog, tensors, out_dtype
      except ProgramTooLarge:
        if not launches_allowed:
          raise lower.NotLowerable('needs more than one launch')
        if not self._carve(root):
          raise

  def _run_map(self, root, out_shape, out_dtype=None):
    prog, tensors, out_dtype = self._emit_map(root, out_shape, out_dtype)
    out = self.empty(out_shape, out_dtype)
    if out.numel():
      self.launches += 1
      kernels.map_fused(prog, tensors, out)
      self._last_map = (prog, tensors, tuple(out.shape), out_dtype)      # (evaluate_map may remember it)
    return out

  # -- lowered programs of operator trees seen before ---------------------------------------------------------
  def _op_structure(self, op):
    """(structure, variable names in order of appearance, reads the tile's position?) of an operator tree made of
    registered operators only -- None if it calls anything that is TRACED (a user's function: what it computes may
    depend on values it closes over, which no key can see).  Remembered on the operator object: trees of optimised
    DAGs answered from the plan table (expr/plan.py) are shared between evaluations."""
    memo = getattr(op, '_lowering_structure', False)
    if memo is not False:
      return memo
    names, positional = [], [False]

    def walk(o):
      if isinstance(o, LocalInput):
        if o.idx in ('extent', 'axis'):
          if o.idx == 'extent':
            positional[0] = True
          return o.idx
        if o.idx not in names:
          names.append(o.idx)
        return names.index(o.idx)
      if not isinstance(o, FnCallExpr) or (o.fn not in lower.MAP_RULES and o.fn not in lower.REDUCE_RULES):
        raise lower.NotLowerable('traced')
      if getattr(o.fn, '_sp_random', None) is not None or getattr(o.fn, '_sp_tile_fn', False):
        raise lower.NotLowerable('not a kernel')
      kw = ()
      if o.kw:
        kw = tuple(sorted((k, v if isinstance(v, (bool, int, float, str, type(None))) else np.dtype(v).str)
                          for k, v in o.kw.items()))
      if isinstance(o, LocalMapLocationExpr):
        positional[0] = True
      return (type(o).__name__, id(o.fn), kw, tuple([walk(d) for d in o.deps]))
    try:
      memo = (walk(op), tuple(names), positional[0])
      hash(memo)
    except (lower.NotLowerable, TypeError):
      memo = None
    try:
      op._lowering_structure = memo
    except AttributeError:
      pass
    return memo

  def _lowering_key(self, op, inputs, ex, extra):
    st = self._op_structure(op)
    if st is None:
      return None
    structure, names, positional = st
    described, ptrs = [], []
    for n in names:
      v = inputs.get(n)
      t = type(v)
      if t is D.DevArray:
        # (address: two names may read the same bytes -- the emitter then loads them once -- and alignment decides
        #  the kernel's vector width; both are functions of the low bits and of equality between operands)
        p = v.data_ptr()
        described.append((v.dtype, v.shape, v.strides, p & 15, ptrs.index(p) if p in ptrs else len(ptrs)))
        ptrs.append(p)
      elif t in (bool, int, float):
        described.append((t, v))
      elif isinstance(v, np.generic):
        described.append((v.dtype.str, v.item()))
      else:
        return None          # NumPy operands (uploaded through the driver-array cache), empty / sparse / masked tiles
    where = (ex.ul, ex.lr, ex.array_shape) if positional else ex.shape
    return (structure, tuple(described), where, extra)

  def _remember(self, key, launch, inputs, op):
    """Keep the program of a launch whose operands were exactly (some of) the device tiles handed in; operands are
    remembered by their POSITION among the operator tree's variables (two trees of one structure name them apart)."""
    prog, tensors, out_shape, out_dtype = launch[:4]
    names = op._lowering_structure[1]
    by_id = {id(inputs[n]): i for i, n in enumerate(names) if type(inputs.get(n)) is D.DevArray}
    order = [by_id.get(id(t)) for t in tensors]
    if None in order:
      return              # a carved sub-expression, a dense copy of a view, an uploaded operand: not replayable as is
    self._lowered[key] = (prog, order, out_shape, np.dtype(out_dtype)) + tuple(launch[4:])
    while len(self._lowered) > 512:
      self._lowered.popitem(last=False)

  def seed_random(self, seed):
    self._rng_seed = int(seed) & (2**63 - 1)
    self._rng_offset = 0

  def random_tile(self, kind, shape, dtype, low=0, high=10):
    """One srandom tile (srandom.py:38-50) from the counter-based generator; consecutive
    fills consume consecutive counters of this worker's stream."""
    out = self.empty(shape, dtype)
    n = out.numel()
    if n:
      self.launches += 1
      kernels.random_fill(out, kind, self._rng_seed, self._rng_offset, low, high)
      self._rng_offset += n + (n & 1)
    return out

  def evaluate_map(self, op, inputs, ex):
    """tile_mapper body: the fused map as ONE launch (map.py:74, local.py:115-127)."""
    rnd = getattr(getattr(op, 'fn', None), '_sp_random', None)
    if rnd is not None:
      return self.random_tile(rnd[0], ex.shape, rnd[1], **(op.kw or {}))
    if getattr(getattr(op, 'fn', None), '_sp_tile_fn', False):
      return self._call_tile_fn(op, inputs, ex)
    if any(tile.is_sparse_blob(v) for v in inputs.values()):
      return self._evaluate_sparse_map(op, inputs, ex)
    op, inputs = self._materialise_random(op, inputs, ex)
    key = self._lowering_key(op, inputs, ex, None)
    if key is not None:
      hit = self._lowered.get(key)
      if hit is not None:
        prog, order, out_shape, out_dtype = hit
        names = op._lowering_structure[1]
        out = D.empty(out_shape, out_dtype)
        self.launches += 1
        self.lowering_hits += 1
        kernels.map_fused(prog, [inputs[names[i]] for i in order], out)
        return out
    try:
      root = lower.infer(op, inputs, ex, self.dtype_of)
      self._last_map = None
      out = self._run_map(root, root.shape if root.kind != 'const' else ex.shape)
      if key is not None and self._last_map is not None:
        self._remember(key, self._last_map, inputs, op)
      return out
    except ProgramTooLarge:
      return self._evaluate_split(op, inputs, ex)
    except lower.NotLowerable:
      return self._evaluate_eager(op, inputs, ex)

  def prelower_map(self, op, inputs, ex):
    """Lower NOW what evaluate_map(op, inputs, ex) will launch, into the table of lowered programs, without running or
    allocating anything: the optimiser calls it for the maps of a DAG it has just optimised for the first time (the
    reference generates the code of its fused operators in the optimiser too, optimize.py:1023-1076), so that the
    first evaluation finds its program like every later one does.  Only what can be replayed from the table is
    prepared -- device tiles and scalars in, one launch; anything else is left to evaluate_map.  Returns whether the
    table now answers this key."""
    fn = getattr(op, 'fn', None)
    if getattr(fn, '_sp_random', None) is not None or getattr(fn, '_sp_tile_fn', False):
      return False
    if any(tile.is_sparse_blob(v) for v in inputs.values()) or self._materialise_random_needed(op):
      return False
    key = self._lowering_key(op, inputs, ex, None)
    if key is None:
      return False
    if key in self._lowered:
      return True
    try:
      root = lower.infer(op, inputs, ex, self.dtype_of)
      prog, tensors, out_dtype = self._emit_map(root, root.shape if root.kind != 'const' else ex.shape, None,
                                                launches_allowed=False)
    except (ProgramTooLarge, lower.NotLowerable):
      return False
    self._remember(key, (prog, tensors, tuple(root.shape if root.kind != 'const' else ex.shape), out_dtype), inputs, op)
    return key in self._lowered

  def map_result_meta(self, op, inputs, ex):
    """(dtype, is_sparse) of what evaluate_map(op, inputs, ex) will produce, from the operator tree and the operands'
    dtypes and shapes alone -- `inputs` may hold placeholders of tiles that live on other ranks -- or None when only
    running it can tell (sparse operands, whole-tile functions, a user function that cannot be traced).  The answer
    depends on nothing a rank holds alone, so every rank gets the same one."""
    fn = getattr(op, 'fn', None)
    rnd = getattr(fn, '_sp_random', None)
    if rnd is not None:
      return (np.dtype(rnd[1]), False)
    if getattr(fn, '_sp_tile_fn', False) or self._op_structure(op) is None:
      return None          # (a user's function would have to be RUN to be traced: never for a derivation)
    described = {}
    for name, v in inputs.items():
      if tile.is_sparse_blob(v) or isinstance(v, tile.MaskedBlob):
        return None
      if isinstance(v, (distarray.Absent, D.DevArray)):
        v = lower.V('tensor', dtype=v.dtype, shape=tuple(v.shape), tensor=None)
      described[name] = v
    try:
      if self._materialise_random_needed(op):
        return None
      root = lower.infer(op, described, ex, self.dtype_of)
      return (np.dtype(root.dtype), False) if root.dtype is not None else None
    except Exception:   # noqa: BLE001  (whatever stops the derivation stops it on every rank alike)
      return None

  def _materialise_random_needed(self, op):
    for d in getattr(op, 'deps', ()):
      if getattr(getattr(d, 'fn', None), '_sp_random', None) is not None:
        return True
      if isinstance(d, FnCallExpr) and self._materialise_random_needed(d):
        return True
    return False

  # -- local functions that are not element-wise kernels ---------------------------------------------------------
  def _evaluate_eager(self, op, inputs, ex):
    """The reference runs ANY Python callable on its NumPy tiles (FnCallExpr.evaluate, local.py:115-127).  A tree
    that cannot become one fused kernel is evaluated node by node, like there: every sub-tree that does lower still
    runs as a kernel; a function that does not is called on the device tiles themselves -- they answer the ndarray
    calls mappers make (devarray.py), each with its own kernel launch -- and, only if it asks for something they
    cannot do, on host copies (device -> host, the call, host -> device: the slow path, counted in
    `host_round_trips` and announced once per function).  Never the oracle, never silently."""
    def ev(node):
      if isinstance(node, LocalInput):
        if node.idx == 'extent':
          return ex.to_tuple()
        v = inputs[node.idx]
        return np.ndarray(v.shape, v.dtype) if isinstance(v, tile.EmptyBlob) else v      # (tile.pyx:72-79: uninitialised)
      if not isinstance(node, FnCallExpr):
        raise lower.NotLowerable('cannot evaluate local expression %r' % (node,))
      if node is not op:
        try:
          root = lower.infer(node, inputs, ex, self.dtype_of)
          return self._run_map(root, root.shape if root.kind != 'const' else ex.shape)
        except (lower.NotLowerable, ProgramTooLarge):
          pass
      return self.call_local_fn(node.fn, [ev(d) for d in node.deps], dict(node.kw or {}), node.fn_name())
    out = ev(op)
    if not isinstance(out, D.DevArray):
      out = self.from_numpy(np.broadcast_to(np.asarray(out), ex.shape))
    elif tuple(out.shape) != tuple(ex.shape) and out.size == 1:
      out = self.from_numpy(np.broadcast_to(out.numpy(), ex.shape))
    return out

  def call_local_fn(self, fn, args, kw, name='local function'):
    """fn(*args) on device tiles; on host copies if the device tiles say they cannot answer it -- and only then:
    D.DeviceTileCannot is raised by DevArray alone (lower.NotLowerable by the lowering of an operator it forwards),
    so an exception of the user function's own is the user's error and propagates from its FIRST run (a function
    with side effects is never run twice because of a bug in it)."""
    import warnings
    self.launches += 1
    try:
      return fn(*args, **kw)
    except (D.DeviceTileCannot, lower.NotLowerable):
      pass
    self.host_round_trips += 1
    if fn not in self._warned_host:
      self._warned_host.add(fn)
      warnings.warn('%s cannot run on device tiles: evaluated on host copies (device -> host -> device round trip '
                    'per tile)' % name, RuntimeWarning, stacklevel=3)
    host_args = [a.numpy() if isinstance(a, D.DevArray) else a for a in args]
    res = fn(*host_args, **kw)
    if isinstance(res, (np.ndarray, np.generic, bool, int, float)):
      return self.from_numpy(np.asarray(res))
    return res

  def _call_tile_fn(self, op, inputs, ex):
    """A local function that works on backend tensors itself (region_map, k-means bodies ...)."""
    args = []
    for d in op.deps:
      if not isinstance(d, LocalInput):
        raise lower.NotLowerable('%s takes whole tiles: it cannot be fused with other local ops' % op.fn_name())
      args.append(ex.to_tuple() if d.idx == 'extent' else inputs[d.idx])
    self.launches += 1
    return op.fn(*args, **(op.kw or {}))

  def assign_box(self, dst, slices, value):
    """dst[slices] = value; value: Python/NumPy scalar (fill), NumPy array or backend tensor."""
    view = dst[slices]
    dt = self.dtype_of(dst)
    if np.isscalar(value) or (isinstance(value, np.ndarray) and value.ndim == 0):
      src = self._run_map(lower.const(np.asarray(value).astype(dt)[()], dt), tuple(view.shape), dt)
    elif isinstance(value, np.ndarray):
      src = self.from_numpy(np.broadcast_to(value, tuple(view.shape)).astype(dt))
    else:
      src = self.astype(value, dt)
    self.paste(dst, slices, src)

  def _materialise_random(self, op, inputs, ex):
    """Random sources fused INTO a tree (the reference's fusion does that despite @not_idempotent,
    because the marker is keyed by id() and the optimiser clones nodes: `(r - r).optimized()` draws
    twice there too) become tensor inputs, one fill per occurrence, like the reference's evaluation."""
    if not isinstance(op, FnCallExpr) or getattr(op, '_no_random_below', False):
      return op, inputs
    new_deps, changed = [], False
    for i, d in enumerate(op.deps):
      rnd = getattr(getattr(d, 'fn', None), '_sp_random', None)
      if rnd is not None:
        if not changed:
          inputs = dict(inputs)
        name = '__random_%d_%d' % (id(op), i)
        inputs[name] = self.random_tile(rnd[0], ex.shape, rnd[1], **(d.kw or {}))
        new_deps.append(LocalInput(idx=name))
        changed = True
      elif isinstance(d, FnCallExpr):
        nd, inputs2 = self._materialise_random(d, inputs, ex)
        if nd is not d:
          changed = True
          inputs = inputs2
        new_deps.append(nd)
      else:
        new_deps.append(d)
    if not changed:
      op._no_random_below = True         # (operator trees are not mutated once built: remembered on the tree)
      return op, inputs
    return op.__class__(fn=op.fn, kw=op.kw, pretty_fn=op.pretty_fn, deps=new_deps), inputs

  def _evaluate_split(self, op, inputs, ex):
    """The tree does not fit one kernel: materialise its sub-expressions first."""
    if not isinstance(op, FnCallExpr):
      raise
    new_deps = []
    new_inputs = dict(inputs)
    progressed = False
    for i, d in enumerate(op.deps):
      if isinstance(d, FnCallExpr):
        name = '__split_%d_%d' % (id(op), i)
        new_inputs[name] = self.evaluate_map(d, inputs, ex)
        new_deps.append(LocalInput(idx=name))
        progressed = True
      else:
        new_deps.append(d)
    if not progressed:
      raise ProgramTooLarge('a single local function call does not fit one kernel')
    clone = op.__class__(fn=op.fn, kw=op.kw, pretty_fn=op.pretty_fn, deps=new_deps)
    root = lower.infer(clone, new_inputs, ex, self.dtype_of)
    return self._run_map(root, root.shape)

  def evaluate_fn(self, fn, args, kw, out_shape):
    """Apply one registered local function to backend tensors (one launch)."""
    rule = lower.MAP_RULES[fn]
    vals = []
    for a in args:
      v = lower.value_of_input(a)
      if v.kind == 'tensor' and v.dtype is None:
        v.dtype = self.dtype_of(a)
      vals.append(v)
    root = rule(vals, kw, None)
    return self._run_map(root, root.shape if out_shape is None else out_shape)

  def _axis_split(self, shape, axis):
    if axis is None:
      return 1, int(np.prod(shape, dtype=np.int64)), 1
    if axis < 0:
      axis += len(shape)
    return (int(np.prod(shape[:axis], dtype=np.int64)), int(shape[axis]),
            int(np.prod(shape[axis + 1:], dtype=np.int64)))

  def _build_reduce(self, data, red_op, nat_dtype, shape, axis):
    """The launch recipe of a fused map -> reduce over one tile: (prog, tensors, result shape, natural dtype,
    red_op, O, A, I).  Nothing runs and nothing is allocated."""
    self._prepare(data)
    if data.kind in ('const', 'shape'):
      raise lower.NotLowerable('reduction over a constant')
    nat_dtype = np.dtype(nat_dtype)
    extra = [class_of(nat_dtype)] if nat_dtype != np.bool_ else []
    cls = lower.choose_class(data, extra)
    full_shape = tuple(np.broadcast_shapes(data.shape, shape))
    em = lower.Emitter(cls, full_shape, self.contiguous)
    prog, tensors = em.finish(data, None)
    O, A, I = self._axis_split(full_shape, axis)
    if A == 0 or O * I == 0:
      raise _hip.HipError('reduction over an empty axis')
    if axis is None:
      shape = ()
    else:
      ax = axis if axis >= 0 else axis + len(full_shape)
      shape = full_shape[:ax] + full_shape[ax + 1:]
    return (prog, tensors, shape, nat_dtype, red_op, O, A, I)

  def _run_reduce(self, data, red_op, nat_dtype, shape, axis):
    recipe = self._build_reduce(data, red_op, nat_dtype, shape, axis)
    prog, tensors, shape, nat_dtype, red_op, O, A, I = recipe
    out = self.empty((O * I,), nat_dtype)
    self.launches += 1
    kernels.reduce(prog, tensors, red_op, O, A, I, out)
    self._last_reduce = recipe
    return out.reshape(shape)

  def prelower_reduce(self, op, inputs, ex, axis):
    """prelower_map for the local reduction of a ReduceExpr: the fused map -> reduce program of evaluate_reduce(op,
    inputs, ex, axis) lowered into the table now (the optimiser calls it for a DAG it sees for the first time), so
    that the first evaluation replays it like every later one, with the partials' workspace already at its size
    (`kernels.reduce_warm`)."""
    rule = lower.REDUCE_RULES.get(op.fn)
    if rule is None or any(tile.is_sparse_blob(v) for v in inputs.values()):
      return False
    data_deps = [d for d in op.deps if not (isinstance(d, LocalInput) and d.idx in ('extent', 'axis'))]
    if len(data_deps) != 1 or self._materialise_random_needed(op):
      return False
    key = self._lowering_key(op, inputs, ex, ('reduce', axis))
    if key is None:
      return False
    if key in self._lowered:
      return True
    try:
      data = lower.infer(data_deps[0], inputs, ex, self.dtype_of)
      red_op, data, nat = rule(data, axis, ex)
      recipe = self._build_reduce(data, red_op, nat, ex.shape, axis)
    except (ProgramTooLarge, lower.NotLowerable, _hip.HipError):
      return False
    self._remember(key, recipe, inputs, op)
    if key in self._lowered:
      prog, tensors, _, nat_dtype, red_op, O, A, I = recipe
      kernels.reduce_warm(prog, tensors, red_op, O, A, I, nat_dtype)
    return key in self._lowered

  def evaluate_reduce(self, op, inputs, ex, axis):
    """_reduce_mapper's local reduction (reduce.py:54) incl. the fused map prologue."""
    rule = lower.REDUCE_RULES.get(op.fn)
    if rule is None:
      # a user's local reduce function fn(extent, data, axis) (reduce.py:130-167): called on the device tile
      # (DevArray.sum / max / ... are kernels), on a host copy if that is not enough
      args = []
      for d in op.deps:
        if isinstance(d, LocalInput):
          args.append(ex if d.idx == 'extent' else (axis if d.idx == 'axis' else inputs[d.idx]))
        else:
          args.append(self.evaluate_map(d, inputs, ex))
      out = self.call_local_fn(op.fn, args, dict(op.kw or {}), op.fn_name())
      return out if isinstance(out, D.DevArray) else self.from_numpy(np.asarray(out))
    data_deps = [d for d in op.deps if not (isinstance(d, LocalInput) and d.idx in ('extent', 'axis'))]
    if len(data_deps) != 1:
      raise lower.NotLowerable('reduce over %d operands' % len(data_deps))
    if any(tile.is_sparse_blob(v) for v in inputs.values()):
      return self._evaluate_sparse_reduce(op, data_deps[0], inputs, ex, axis)
    key = self._lowering_key(op, inputs, ex, ('reduce', axis))
    if key is not None:
      hit = self._lowered.get(key)
      if hit is not None:
        prog, order, shape, nat, red_op, O, A, I = hit
        names = op._lowering_structure[1]
        out = D.empty((O * I,), nat)
        self.launches += 1
        self.lowering_hits += 1
        kernels.reduce(prog, [inputs[names[i]] for i in order], red_op, O, A, I, out)
        return out.reshape(shape)
    try:
      data = lower.infer(data_deps[0], inputs, ex, self.dtype_of)
    except ProgramTooLarge:
      raise
    red_op, data, nat = rule(data, axis, ex)
    try:
      self._last_reduce = None
      out = self._run_reduce(data, red_op, nat, ex.shape, axis)
      if key is not None and self._last_reduce is not None:
        self._remember(key, self._last_reduce, inputs, op)
      return out
    except ProgramTooLarge:
      # materialise the map, then reduce the dense result
      dense = self.evaluate_map(data_deps[0], inputs, ex)
      v = lower.V('tensor', dtype=self.dtype_of(dense), shape=tuple(dense.shape), tensor=dense)
      red_op, v, nat = rule(v, axis, ex)
      return self._run_reduce(v, red_op, nat, ex.shape, axis)

  def evaluate_reduce_tensor(self, t, red_op):
    v = lower.V('tensor', dtype=self.dtype_of(t), shape=tuple(t.shape), tensor=t)
    nat = np.bool_ if red_op in ('AND', 'OR') else self.dtype_of(t)
    return self._run_reduce(v, red_op, nat, tuple(t.shape), None)

  def evaluate_argreduce(self, data, ex, axis, which, index_offset, nan_index):
    """Fused (value, first index) reduction of one tile (sorting.py:67-123 in one pass)."""
    if isinstance(data, tile.EmptyBlob):
      raise lower.NotLowerable('argmax/argmin of an uninitialised array')
    v = lower.V('tensor', dtype=self.dtype_of(data), shape=tuple(data.shape), tensor=data)
    cls = lower.choose_class(v)
    em = lower.Emitter(cls, v.shape, self.contiguous)
    prog, tensors = em.finish(v, None)
    O, A, I = self._axis_split(v.shape, axis)
    out_idx = self.empty((O * I,), np.int64)
    val_dtype = {_hip.SP_F32: np.float32, _hip.SP_F64: np.float64, _hip.SP_I64: np.int64}[cls]
    out_val = self.empty((O * I,), val_dtype)
    self.launches += 1
    kernels.argreduce(prog, tensors, which, O, A, I, index_offset,
                      nan_index, out_idx, out_val)
    if np.dtype(val_dtype) != v.dtype:
      out_val = self.astype(out_val, v.dtype)
    return out_idx, out_val

  # -- contraction ------------------------------------------------------------------
  def rowdot_colsum(self, x, w, y):
    """(d,) partial of `sum(x * (dot(x, w) - y), axis=0)` for one row tile (expr/rowdot.py): one pass over x when
    the layout allows (sp_rowdot_colsum_f32), else the two launches the rewrite replaced."""
    d = int(x.shape[1])
    wd = self.cached_numpy(w, (slice(0, d),))
    wd = wd.reshape(d)
    yd = None
    if y is not None:
      yd = y.reshape(y.shape[0]) if y.dim() == 2 and y.shape[1] == 1 else y
    out = self.empty((d,), np.float32)
    self.launches += 1
    if wd.is_contiguous() and (yd is None or yd.dim() == 1) and kernels.rowdot_colsum(x, wd, yd, out):
      return out
    t = self.dot(x, wd.reshape(d, 1))
    r = t if y is None else t - y.reshape(t.shape)
    return (x * r).sum(0)

  def dot(self, a, b):
    """ndarray.dot for backend tensors: MFMA GEMM for fp32 matrix.matrix, fused
    multiply-reduce launches for everything else."""
    if tile.is_sparse_blob(a) or tile.is_sparse_blob(b):
      return self._sparse_dot(a, b)
    a_dt, b_dt = self.dtype_of(a), self.dtype_of(b)
    res_dt = np.result_type(a_dt, b_dt)
    if a.dim() == 2 and b.dim() == 2 and b.shape[1] == 1:
      # (M,K).(K,1), the lreg `X.w` (linear_regression.py:10-16): HBM-bound matrix.vector,
      # not a GEMM -- one fused multiply-reduce pass over A
      return self.dot(a, b.reshape(b.shape[0])).reshape(a.shape[0], 1)
    if a.dim() == 2 and b.dim() == 2 and a.shape[0] == 1:
      return self.dot(a.reshape(a.shape[1]), b).reshape(1, b.shape[1])
    if a.dim() == 2 and b.dim() == 2 and a_dt == b_dt and a_dt in (np.float32, np.float64):
      # fp32 / fp64 MFMA GEMM (sp_gemm_f32 / sp_gemm_f64)
      M, K = a.shape
      N = b.shape[1]
      c = self.empty((M, N), a_dt)
      if a.stride(1) != 1:
        a = self.copy(a)
      if b.stride(1) != 1:
        b = self.copy(b)
      self.launches += 1
      if self.gemm_events is not None:
        e0, e1 = kernels.Event(), kernels.Event()
        e0.record()
        kernels.gemm_f32(a, b, c, accumulate=False)
        e1.record()
        self.gemm_events.append((e0, e1, M, N, K))
      else:
        kernels.gemm_f32(a, b, c, accumulate=False)
      return c
    va = lower.V('tensor', dtype=a_dt, shape=tuple(a.shape), tensor=self.contiguous(a))
    vb = lower.V('tensor', dtype=b_dt, shape=tuple(b.shape), tensor=self.contiguous(b))
    if a.dim() == 2 and b.dim() == 1:      # (M,K).(K,) -> (M,): row kernels
      prod = lower.apply('MUL', np.multiply, [va, vb])
      return self._run_reduce(prod, 'SUM', res_dt, prod.shape, 1)
    if a.dim() == 1 and b.dim() == 1:      # (K,).(K,) -> scalar
      prod = lower.apply('MUL', np.multiply, [va, vb])
      return self._run_reduce(prod, 'SUM', res_dt, prod.shape, None)
    if a.dim() == 1 and b.dim() == 2:      # (K,).(K,N) -> (N,): column kernels
      va.shape = (a.shape[0], 1)
      prod = lower.apply('MUL', np.multiply, [va, vb])
      return self._run_reduce(prod, 'SUM', res_dt, prod.shape, 0)
    if a.dim() == 2 and b.dim() == 2:      # generic dtype: [M,K,1] * [1,K,N] summed over K
      M, K = a.shape
      N = b.shape[1]
      va.shape = (M, K, 1)
      vb.shape = (1, K, N)
      prod = lower.apply('MUL', np.multiply, [va, vb])
      return self._run_reduce(prod, 'SUM', res_dt, (M, K, N), 1)
    raise lower.NotLowerable('dot of %d-d and %d-d operands' % (a.dim(), b.dim()))

  def gemm_into(self, a, b, out, accumulate=False):
    """out (+)= a . b for 2-D fp32 / fp64 tensors that may be strided views (inner stride 1): the building block
    of the pipelined joins (dot.ksplit_plan), one sp_gemm launch, nothing allocated."""
    self.launches += 1
    self.gemms += 1
    if self.gemm_events is not None:
      e0, e1 = kernels.Event(), kernels.Event()
      e0.record()
      kernels.gemm_f32(a, b, out, accumulate=accumulate)
      e1.record()
      self.gemm_events.append((e0, e1, a.shape[0], b.shape[1], a.shape[1]))
    else:
      kernels.gemm_f32(a, b, out, accumulate=accumulate)
    return out

  def dot_chunked(self, a, rhs):
    """a . B with B arriving as column chunks (distarray.ChunkedWhole): one GEMM per chunk into the
    matching columns of C, each launched as soon as its chunk's gather has landed, so the remaining
    gathers (RCCL stream) overlap with the GEMMs (this stream)."""
    if not (a.dim() == 2 and self.dtype_of(a) == rhs.dtype and rhs.dtype in (np.float32, np.float64)):
      whole = self.empty(rhs.shape, rhs.dtype)
      for i in range(len(rhs.chunks)):
        c0, c1, t = rhs.ready(i)
        self.paste(whole, (slice(0, rhs.shape[0]), slice(c0, c1)), t)
      return self.dot(a, whole)
    M, K = a.shape
    N = rhs.shape[1]
    if a.stride(1) != 1:
      a = self.copy(a)
    c = self.empty((M, N), rhs.dtype)
    for i in range(len(rhs.chunks)):
      c0, c1, t = rhs.ready(i)
      self.launches += 1
      if self.gemm_events is not None:
        e0, e1 = kernels.Event(), kernels.Event()
        e0.record()
        kernels.gemm_f32(a, t, c[:, c0:c1], accumulate=False)
        e1.record()
        self.gemm_events.append((e0, e1, M, c1 - c0, K))
      else:
        kernels.gemm_f32(a, t, c[:, c0:c1], accumulate=False)
    return c

  # -- k-means tile bodies (examples/sklearn/cluster/k_means_.py) ---------------------
  def _as_device(self, t):
    if isinstance(t, np.ndarray):
      return self.cached_numpy(t, (slice(None),) * t.ndim)
    return t

  def _rows(self, t):
    """2-D view with inner stride 1 (a box copy if the tile is a strided view)."""
    if t.dim() != 2:
      raise lower.NotLowerable('expected a 2-D tile, got %d-d' % t.dim())
    if t.shape[1] > 1 and t.stride(1) != 1:
      t = self.copy(t)
    if self.dtype_of(t) not in (np.float32, np.float64):
      t = self.astype(t, np.float64)
    return t

  def _labels_i64(self, labels, n):
    origin, _ = self._known_labels(labels)
    if origin is not None and origin.numel() == n:
      return self.contiguous(origin).reshape(n)
    labels = self.astype(labels, np.int64) if self.dtype_of(labels) != np.int64 else labels
    return self.contiguous(labels).reshape(n)

  def nearest_center(self, points, centers, tier=_hip.NEAREST_AUTO):
    """np.argmin(cdist(points, centers), axis=1) -> int64 (n,)  (k_means_.py:61-66)."""
    points, centers = self._rows(self._as_device(points)), self._rows(self._as_device(centers))
    out = self.empty((points.shape[0],), np.int64)
    self.launches += 1
    prepared = None
    if self._fixed_points is not None and self.dtype_of(points) == np.float32 and points.shape[0] >= 1024:
      # inside a fit (fixed_points): the points' bf16 images are made on their first use and kept to its end
      key = (points.data_ptr(), tuple(points.shape), tuple(points.strides))
      prepared = self._fixed_points.get(key)
      if prepared is None:
        if len(self._fixed_points) >= 64:      # (points that are re-made every iteration: keep nothing of them)
          self._fixed_points.clear()
        prepared = self._fixed_points[key] = (kernels.prepare_points(points), points)     # (the tile stays alive)
      prepared = prepared[0]
    out = kernels.nearest_center(points, centers, out, tier, prepared)
    if self._fit_labels is not None and centers.shape[0] <= (1 << 24):
      self._note_labels(out)           # (labels below 2^24: their float32 image is exact)
    return out

  def fixed_points(self):
    """with be.fixed_points(): -- the caller promises that the point tiles it passes to nearest_center are not
    written inside the block (the iterations of one k-means fit): what the kernels derive from the points alone is
    then derived once per tile.  Nothing outlives the block."""
    return _FixedPoints(self)

  def bincount(self, labels, k):
    """np.bincount(labels.astype(int), minlength=k) -> int64 (k,)  (k_means_.py:69-72)."""
    counts = self._take_counts(labels) if self._fit_labels is not None else None
    if counts is not None and counts.numel() == int(k):
      return counts                    # counted by this fit's segment_sum of the same labels (sp_segment_sum_counts)
    labels = self._labels_i64(labels, int(np.prod(labels.shape)))
    self.launches += 1
    return kernels.bincount(labels, int(k), self.empty((int(k),), np.int64))

  def segment_sum(self, points, labels, k):
    """out[c] = points[labels == c].sum(axis=0), in the points' dtype (k_means_.py:75-97)."""
    points = self._rows(points)
    given = labels
    labels = self._labels_i64(labels, points.shape[0])
    out = self.empty((int(k), points.shape[1]), self.dtype_of(points))
    self.launches += 1
    if self._fit_labels is None:
      return kernels.segment_sum(points, labels, int(k), out)
    # inside a fit the counts of the same labels are wanted too (kmeans_count_mapper): the counting sort has them
    counts = self.empty((int(k),), np.int64)
    kernels.segment_sum(points, labels, int(k), out, counts)
    if isinstance(given, D.DevArray):
      self._note_labels(given, counts=counts)
    if labels is not given:
      self._note_labels(labels, counts=counts)
    return out

  def concat(self, a, b, axis=0):
    """np.concatenate((a, b), axis) as two box copies (manipulation.py:51)."""
    dt = np.result_type(self.dtype_of(a), self.dtype_of(b))
    shape = list(a.shape)
    shape[axis] += b.shape[axis]
    out = self.empty(shape, dt)
    lo = [slice(0, n) for n in a.shape]
    self.paste(out, tuple(lo), self.astype(a, dt))
    hi = [slice(0, n) for n in b.shape]
    hi[axis] = slice(a.shape[axis], shape[axis])
    self.paste(out, tuple(hi), self.astype(b, dt))
    return out

  def reduce_axis(self, t, red_op, axis):
    """np.sum / np.prod of a tile along one axis (the scan operator's per-tile totals, scan.py:25)."""
    t = self.contiguous(t)
    v = lower.V('tensor', dtype=self.dtype_of(t), shape=tuple(t.shape), tensor=t)
    return self._run_reduce(v, red_op, self.dtype_of(t), tuple(t.shape), axis)

  def sort_axis(self, t, axis, indices=False):
    """np.sort(t, axis) / np.argsort(t, axis, kind='stable') of a tile (sort.py:68-69, :137-138): the axis is
    brought last by a strided copy, sp_sort_rows sorts every line, and the result is copied back."""
    t = self.contiguous(self._as_device(t))
    shape = tuple(t.shape)
    nd = len(shape)
    axis = axis if axis >= 0 else axis + nd
    n = shape[axis]
    outer = int(np.prod(shape[:axis], dtype=np.int64))
    inner = int(np.prod(shape[axis + 1:], dtype=np.int64))
    self.launches += 1
    if inner == 1:
      vals, idx = kernels.sort_rows(t.reshape(outer, n), values=not indices, indices=indices)
      return (idx if indices else vals).reshape(shape)
    # [outer, n, inner] -> [outer, inner, n] (a 3-d strided copy whatever the rank of the tile), sort, and back
    lines = self.contiguous(t.reshape(outer, n, inner).movedim(1, 2))
    vals, idx = kernels.sort_rows(lines.reshape(outer * inner, n), values=not indices, indices=indices)
    out = (idx if indices else vals).reshape(outer, inner, n)
    return self.contiguous(out.movedim(2, 1)).reshape(shape)

  def convolve(self, image, filters):
    """stencil.py:29-45 as a GEMM: P[(n, x, y), (c, i, j)] = image[n, c, x+i, y+j] (0 beyond the edge) by one strided
    box copy per (c, i, j); P . filters[(c, i, j), f] on the MFMA GEMM; back to [n, f, x, y]."""
    image = self.contiguous(self._as_device(image))
    filters = self.contiguous(self._as_device(filters))
    n, c, w, h = image.shape
    f, fc, fw, fh = filters.shape
    assert c == fc
    dt = np.result_type(self.dtype_of(image), self.dtype_of(filters))
    image, filters = self.astype(image, dt), self.astype(filters, dt)
    k = c * fw * fh
    patches = self.zeros((n, w, h, k), dt)
    for ci in range(c):
      for i in range(min(fw, w)):
        for j in range(min(fh, h)):
          col = (ci * fw + i) * fh + j
          self.paste(patches, (slice(0, n), slice(0, w - i), slice(0, h - j), slice(col, col + 1)),
                     image[:, ci, i:, j:].reshape(n, w - i, h - j, 1))
    fm = self.copy(filters.reshape(f, k).t())                     # [(c, i, j), f]
    out = self.dot(patches.reshape(n * w * h, k), fm)             # [(n, x, y), f]
    return self.copy(out.reshape(n, w, h, f).permute(0, 3, 1, 2))

  def maxpool(self, region, pool_size, stride, out_shape):
    """out[n, c, X, Y] = max of the pixels (a, b) with a // stride == X, b // stride == Y whose offset inside the
    window is below pool_size -- the indexing of the reference's (disabled) loop, stencil.py:61-70: one strided copy +
    one max-merge per window offset, starting from its -1e12 (stencil.py:57-59)."""
    region = self.contiguous(self._as_device(region))
    n, c, w, h = region.shape
    dt = self.dtype_of(region)
    out = self.evaluate_fn(np.add, [self.zeros(out_shape, dt), dt.type(-1e12)], {}, tuple(out_shape))
    span = pool_size if pool_size < stride else stride      # pixel a belongs to window a // stride (stencil.py:68-70)
    for i in range(span):
      for j in range(span):
        part = region[:, :, i::stride, j::stride]
        if part.numel() == 0:
          continue
        part = self.copy(part)
        self.update_box(out, (0, 0, 0, 0), tuple(part.shape), part, np.maximum, tile.MASK_ALL_SET, None)
    return out

  def gather_rows(self, block, rows):
    """block[rows] along axis 0 (filter.py:50-75); `rows` is a host int64 vector relative to the block."""
    self.launches += 1
    return kernels.gather_rows(self.contiguous(self._as_device(block)), self.from_numpy(np.ascontiguousarray(rows, dtype=np.int64)))

  def cumscan(self, t, axis, product=False):
    """np.cumsum / np.cumprod along `axis` (sp_cumscan)."""
    t = self.contiguous(t)
    if self.dtype_of(t) == np.bool_:
      t = self.astype(t, np.int64)
    out = self.empty(tuple(t.shape), self.dtype_of(t))
    if out.numel():
      self.launches += 1
      kernels.cumscan(t, out, axis, product)
    return out

  # -- sparse tiles: canonical CSR in HBM, spartan_amd/sparse.py over csrc/sparse.hip -----------
  def is_sparse(self, x):
    return isinstance(x, sparse_mod.CsrTile)

  def sparse_blob(self, mat, dtype=None):
    """What a mapper yielded (scipy.sparse, any format) or a device tile -> device tile of `dtype`."""
    if isinstance(mat, sparse_mod.CsrTile):
      if dtype is None or mat.dtype == np.dtype(dtype):
        return mat
      return sparse_mod.CsrTile(mat.shape, dtype, mat.indptr, mat.indices, self.astype(mat.data, dtype))
    self.launches += 1
    return sparse_mod.from_scipy(mat, self.device, dtype)

  def sparse_to_host(self, b):
    return sparse_mod.to_scipy(b)

  def sparse_parts(self, b):
    """The three arrays of a sparse tile as they travel between ranks: int64 row pointers, int32 columns, values."""
    return b.indptr.contiguous(), b.indices.contiguous(), b.data.contiguous()

  def sparse_parts_empty(self, shape, dtype, nnz):
    """Receive buffers for sparse_parts of a tile of `shape` with `nnz` stored entries."""
    return (D.empty((int(shape[0]) + 1,), np.int64), D.empty((int(nnz),), np.int32), D.empty((int(nnz),), dtype))

  def sparse_from_parts(self, shape, dtype, parts):
    return sparse_mod.CsrTile(shape, dtype, *parts)

  def sparse_empty(self, shape, dtype):
    return sparse_mod.empty(shape, dtype, self.device)

  def sparse_slice(self, b, slices):
    r0, r1, _ = slices[0].indices(b.shape[0])
    c0, c1 = slices[1].indices(b.shape[1])[:2] if len(slices) > 1 else (0, b.shape[1])
    self.launches += 1
    return sparse_mod.slice_box(b, r0, r1, c0, c1)

  def sparse_paste(self, shape, dtype, pieces):
    self.launches += 1
    return sparse_mod.paste(shape, dtype, [(ul[0], ul[1], p) for ul, p in pieces])

  def sparse_reduce(self, old, upd, reducer):
    if reducer is not np.add:
      raise NotImplementedError('sparse tiles combine with np.add only (got %r)' % (reducer,))
    self.launches += 1
    return sparse_mod.add(old, upd)

  def sparse_update(self, old, ul, lr, upd, reducer):
    if reducer is not None and reducer is not np.add:
      raise NotImplementedError('sparse tiles combine with np.add only (got %r)' % (reducer,))
    self.launches += 1
    return sparse_mod.update_box(old, ul[0], lr[0], ul[1], lr[1], upd, add_to_old=reducer is not None)

  def sparse_scatter(self, dst, ul, blob, mode, mask):
    self.launches += 1
    blob = self.sparse_blob(blob, self.dtype_of(dst))
    sparse_mod.scatter(blob, dst, ul[0], ul[1], mode, mask)

  def sparse_to_dense(self, b):
    self.launches += 1
    return sparse_mod.to_dense(b)

  def dense_to_sparse(self, t, dtype=None):
    """The non-zero cells of a dense block as a sparse tile (a dense update of a sparse array, tile.pyx:284-297)."""
    self.launches += 4
    return sparse_mod.from_dense(t, dtype)

  def sparse_transpose(self, b):
    self.launches += 1
    return sparse_mod.transpose(b)

  def sparse_reshape(self, b, offset, shape):
    self.launches += 1
    return sparse_mod.reshape_rect(b, offset, shape)

  def sparse_random(self, shape, density, dtype):
    """scipy.sparse.rand's role (srandom.py:57-65) from the counter-based generator: int(density * size)
    uniformly drawn positions (colliding ones merge), uniform [0, 1) values."""
    m, n = int(shape[0]), int(shape[1])
    k = int(density * m * n)
    if k <= 0 or m == 0 or n == 0:
      return self.sparse_empty((m, n), dtype)
    rows = self.random_tile('randint', (k,), np.int32, 0, m)
    cols = self.random_tile('randint', (k,), np.int32, 0, n)
    t = sparse_mod.from_coo((m, n), dtype, rows, cols, self.zeros((k,), dtype))
    # one uniform value per KEPT position (positions drawn twice were merged), so every value is in [0, 1)
    return sparse_mod.CsrTile(t.shape, t.dtype, t.indptr, t.indices, self.random_tile('uniform', (t.nnz,), dtype))

  def _sparse_dot(self, a, b):
    """tile_a.dot(tile_b) with a sparse operand (dot.py:212-240)."""
    self.launches += 1
    if isinstance(a, sparse_mod.CsrTile):
      if isinstance(b, sparse_mod.CsrTile):
        return sparse_mod.spgemm(a, b)
      return sparse_mod.spmm(a, self._as_device(b))
    # dense x sparse = (sparse^T x dense^T)^T
    a = self._as_device(a)
    vec = a.dim() == 1
    at = a.reshape(1, -1) if vec else a
    out = sparse_mod.spmm(sparse_mod.transpose(b), self.copy(at.t()))
    out = self.copy(out.t())
    return out.reshape(-1) if vec else out

  def _evaluate_sparse_map(self, op, inputs, ex):
    """A local map tree with sparse operands, node by node (the reference hands the scipy matrices to the
    NumPy function, local.py:115-127): sparse (+|-) sparse and scalings stay sparse; a ufunc over one sparse
    and one dense operand sees the sparse one densified (local.py:120-126); dense sub-trees are fused maps."""
    from .expr import builtins as B

    def ev(node):
      if isinstance(node, LocalInput):
        return ex.to_tuple() if node.idx == 'extent' else inputs[node.idx]
      if not isinstance(node, FnCallExpr):
        raise lower.NotLowerable('cannot evaluate local expression %r' % (node,))
      args = [ev(d) for d in node.deps]
      flags = [tile.is_sparse_blob(a) for a in args]
      fn = node.fn
      if getattr(fn, '_sp_tile_fn', False):      # a builder that works on blobs itself (sparse_diagonal ...), fused in
        self.launches += 1
        return fn(*args, **(node.kw or {}))
      if isinstance(fn, np.ufunc) and len(args) == 2 and (flags[0] ^ flags[1]):
        args = [self.sparse_to_dense(a) if f else a for a, f in zip(args, flags)]
        flags = [False, False]
      if not any(flags):
        if fn not in lower.MAP_RULES:
          raise lower.NotLowerable('%s next to sparse operands has no GPU lowering' % node.fn_name())
        return self.evaluate_fn(fn, args, dict(node.kw or {}), ex.shape)
      self.launches += 1
      if fn is np.add and all(flags) and len(args) == 2:
        return sparse_mod.add(args[0], args[1])
      if fn is np.subtract and all(flags) and len(args) == 2:
        return sparse_mod.add(args[0], args[1], -1)
      if fn is np.negative:
        return sparse_mod.scaled(args[0], -1.0)
      if fn is np.multiply and len(args) == 2 and sum(flags) == 1:
        other = args[1] if flags[0] else args[0]
        if isinstance(other, np.ndarray) and other.ndim == 0:
          other = other[()]
        if isinstance(other, (int, float, np.generic)):
          return sparse_mod.scaled(args[0] if flags[0] else args[1], float(other))
      raise NotImplementedError('%s on sparse tiles is not supported on the GPU backend' % node.fn_name())

    return ev(op)

  def _evaluate_sparse_reduce(self, op, data_dep, inputs, ex, axis):
    """Local reduction of a sparse tile: sums only (scipy's .sum(axis), dense result; reduce.py:57-66)."""
    from .expr import builtins as B
    if op.fn is not B._sum_local:
      raise NotImplementedError('%s of a sparse tile is not supported on the GPU backend' % op.fn_name())
    t = inputs[data_dep.idx] if isinstance(data_dep, LocalInput) else self._evaluate_sparse_map(data_dep, inputs, ex)
    if not tile.is_sparse_blob(t):
      return self.evaluate_reduce_tensor(t, 'SUM') if axis is None else self.reduce_axis(t, 'SUM', axis)
    self.launches += 1
    if axis is None:
      if t.nnz == 0:
        return self.zeros((), t.dtype)
      return self.evaluate_reduce_tensor(t.data, 'SUM')
    if axis in (1, -1):
      return sparse_mod.row_sums(t)
    return sparse_mod.row_sums(sparse_mod.transpose(t))

  def synchronize(self):
    D.synchronize()

  def liveness_probe(self, timeout_s=2.0):
    """One round trip through the device on a stream of its own (heartbeat.py): False if it does not come back."""
    import time
    if getattr(self, '_probe_stream', None) is None:
      self._probe_stream = D.Stream()
    ev = D.Event().record(self._probe_stream)
    deadline = time.time() + timeout_s
    while not D._event_done(ev):
      if time.time() > deadline:
        return False
      time.sleep(0.001)
    return True
