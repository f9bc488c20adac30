"""TEST INFRASTRUCTURE -- not part of the product path.

A NumPy tile-kernel backend for the spartan_amd host framework: it evaluates
LocalExpr trees the way the reference's worker does (call the local function on
NumPy tiles, spartan/expr/operator/local.py:115-127) and applies Tile.merge with
NumPy (spartan/array/tile.pyx:200-297).  It exists so that

  * `-m "not gpu"` tests can run the host logic (tiling, extents, DAG, fusion,
    fetch/update plans, the gloo world_size-2 exchange) without a GPU, and
  * bench.py can time a CPU baseline of the same workload on the host cores.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  `spartan_amd.initialize()` never selects it by itself.

Tile payloads are NumPy arrays, exactly what the reference's workers hold (spartan/worker.py:70); across gloo
ranks the transport wraps them as CPU tensors without copying (spartan_amd/comm.py TorchTransport).
"""
import numpy as np
import scipy.sparse as sps

from spartan_amd.array import distarray, tile
from spartan_amd.expr.local import FnCallExpr, LocalInput, LocalMapLocationExpr

_REDUCERS = {None: 'NONE', np.add: 'ADD', np.multiply: 'MUL', np.maximum: 'MAX', np.minimum: 'MIN',
             np.logical_and: 'AND', np.logical_or: 'OR'}
_DTYPES = set(np.dtype(t) for t in (np.float32, np.float64, np.int32, np.int64, np.bool_, np.uint8))


def _np(x):
  """NumPy view of a local value."""
  if isinstance(x, tile.EmptyBlob):
    return np.ndarray(x.shape, x.dtype)  # uninitialised, like tile.pyx:72-79
  return x


class NumpyBackend(object):
  name = 'numpy-oracle'

  def __init__(self):
    self.launches = 0
    self._np_cache = {}

  # -- memory
  def empty(self, shape, dtype):
    return np.empty(tuple(int(s) for s in shape), dtype=self._known(dtype))

  def zeros(self, shape, dtype):
    return np.zeros(tuple(int(s) for s in shape), dtype=self._known(dtype))

  @staticmethod
  def _known(dtype):
    dtype = np.dtype(dtype)
    if dtype not in _DTYPES:
      raise TypeError('unsupported dtype %s' % dtype)
    return dtype

  def from_numpy(self, arr):
    arr = np.asarray(arr)
    arr = arr if arr.flags['C_CONTIGUOUS'] else arr.copy(order='C')  # (ascontiguousarray would make 0-d -> 1-d)
    self._known(arr.dtype)
    return arr.copy()

  def to_numpy(self, t):
    if isinstance(t, np.ndarray):
      return t
    if isinstance(t, tile.EmptyBlob):
      return np.zeros(t.shape, t.dtype)
    return np.array(t, copy=True)

  def dtype_of(self, t):
    if sps.issparse(t):
      return np.dtype(t.dtype)
    if isinstance(t, (tile.EmptyBlob, distarray.Absent, np.ndarray, np.generic)):
      return np.dtype(t.dtype)
    return np.asarray(t).dtype

  def same_dtype(self, t, dtype):
    return self.dtype_of(t) == np.dtype(dtype)

  def contiguous(self, t):
    return t if t.flags['C_CONTIGUOUS'] else np.ascontiguousarray(t).reshape(t.shape)

  def copy(self, t):
    return np.array(t, copy=True, order='C')

  def same_memory(self, a, b):
    return isinstance(a, np.ndarray) and isinstance(b, np.ndarray) and np.shares_memory(a, b)

  def astype(self, t, dtype):
    if self.dtype_of(t) == np.dtype(dtype):
      return t
    return self.from_numpy(_np(t).astype(dtype))

  def cached_numpy(self, arr, slices):
    return self.from_numpy(arr[slices])

  def reducer_name(self, fn):
    return _REDUCERS.get(fn, 'CALLABLE')

  def gemm_into(self, a, b, out, accumulate=False):
    self.launches += 1
    self.gemms = getattr(self, 'gemms', 0) + 1
    prod = _np(a).dot(_np(b))
    out[...] = out + prod if accumulate else prod
    return out

  def paste(self, dst, dst_slices, src):
    view = dst[dst_slices] if dst.ndim else dst
    view[...] = src.reshape(view.shape)

  # -- Tile.merge (tile.pyx:250-283), on the box [ul, lr)
  def update_box(self, dst, ul, lr, src, reducer, mask_mode, mask):
    self.launches += 1
    d = dst
    s = _np(src)
    if d.ndim == 0:
      # (cast as the reference's merge does, tile.pyx:267 `update.astype(old.dtype)`: an int32 target WRAPS)
      d[...] = np.asarray(reducer(d, s) if reducer is not None else s).astype(d.dtype)
      return
    box = tuple(slice(u, l) for u, l in zip(ul, lr))
    s = s.reshape(d[box].shape)
    if mask_mode == 2:
      m = mask[box].astype(bool)
    else:
      m = np.full(d[box].shape, mask_mode == tile.MASK_ALL_SET, dtype=bool)
    region = d[box]
    replaced = ~m
    if np.any(replaced):
      region[replaced] = s[replaced]
    if np.any(m):
      if reducer is not None:
        region[m] = reducer(region[m], s[m])
      else:
        region[m] = s[m]
    if mask is not None:
      mask[box] = 1

  def mask_all_set(self, mask, subslice):
    return bool(np.all(mask[subslice]))

  def mask_first(self, mask):
    return bool(mask.reshape(-1)[0])

  # -- LocalExpr evaluation exactly as the reference worker does it
  def _eval(self, op, inputs, ex):
    if isinstance(op, LocalInput):
      if op.idx == 'extent':
        return ex
      v = inputs[op.idx]
      return _np(v)
    assert isinstance(op, FnCallExpr), op
    if isinstance(op, LocalMapLocationExpr):
      deps = []
      for d in op.deps:
        if isinstance(d, LocalInput) and d.idx == 'extent':
          deps.append(ex.to_tuple())  # local.py:137-149
        else:
          deps.append(self._eval(d, inputs, ex))
    else:
      deps = [self._eval(d, inputs, ex) for d in op.deps]
    # local.py:120-126: a ufunc over one sparse and one dense operand sees the sparse one densified
    if isinstance(op.fn, np.ufunc) and len(deps) == 2 and (sps.issparse(deps[0]) ^ sps.issparse(deps[1])):
      deps = [np.asarray(d.todense()) if sps.issparse(d) else d for d in deps]
    with np.errstate(all='ignore'):
      return op.fn(*deps, **op.kw)

  def _wrap(self, result, shape=None):
    if sps.issparse(result):
      return result          # a sparse local result stays a scipy matrix (tile.pyx:145-152 from_data)
    result = np.asarray(result)
    if shape is not None and result.shape != tuple(shape):
      result = np.broadcast_to(result, shape)
    return self.from_numpy(result)

  def evaluate_map(self, op, inputs, ex):
    self.launches += 1
    if getattr(getattr(op, 'fn', None), '_sp_tile_fn', False):
      # a local function written against backend tensors (region_map ...): same call as on the GPU
      args = [ex.to_tuple() if d.idx == 'extent' else inputs[d.idx] for d in op.deps]
      return op.fn(*args, **(op.kw or {}))
    return self._wrap(self._eval(op, inputs, ex), ex.shape)

  def assign_box(self, dst, slices, value):
    dst[slices] = np.broadcast_to(np.asarray(value), dst[slices].shape).astype(dst.dtype)

  def evaluate_fn(self, fn, args, kw, out_shape):
    self.launches += 1
    return self._wrap(fn(*[_np(a) for a in args], **kw), out_shape)

  def evaluate_reduce(self, op, inputs, ex, axis):
    self.launches += 1
    return self._wrap(self._eval(op, inputs, ex))

  def evaluate_argreduce(self, data, ex, axis, which, index_offset, nan_index):
    """Per-tile part of sorting.py:67-123 (value + first index)."""
    self.launches += 1
    x = _np(data)
    val = x.max(axis) if which == 0 else x.min(axis)
    idx = (np.argmax(x, axis) if which == 0 else np.argmin(x, axis)).astype(np.int64) + index_offset
    idx = np.where(np.isnan(val), nan_index, idx) if val.dtype.kind == 'f' else idx
    return self._wrap(np.asarray(idx, dtype=np.int64)), self._wrap(val)

  def dot(self, a, b):
    self.launches += 1
    a, b = _np(a), _np(b)
    if sps.issparse(a):
      a = a.tocsr()          # dot.py:212-216
    if sps.issparse(b):
      b = b.tocsr()
      if not sps.issparse(a):
        # ndarray.dot(scipy matrix) builds an object array; the product is (B^T A^T)^T
        return self._wrap(np.ascontiguousarray(np.asarray(b.T.dot(np.asarray(a).T)).T))
    return self._wrap(a.dot(b))

  def dot_chunked(self, a, rhs):
    """Host-framework counterpart of HipBackend.dot_chunked (column chunks of a gathered B)."""
    parts = [None] * len(rhs.chunks)
    for i in range(len(rhs.chunks)):
      c0, c1, t = rhs.ready(i)
      parts[i] = _np(t)
    return self.dot(a, self._wrap(np.concatenate(parts, axis=1)))

  # -- k-means tile bodies: the reference's own NumPy/SciPy statements
  def nearest_center(self, points, centers, tier=0):
    """k_means_.py:61-66: np.argmin(cdist(points, centers), axis=1)."""
    from scipy.spatial.distance import cdist
    self.launches += 1
    return self._wrap(np.argmin(cdist(_np(points), _np(centers)), axis=1).astype(np.int64))

  def bincount(self, labels, k):
    """k_means_.py:69-72."""
    self.launches += 1
    return self._wrap(np.bincount(_np(labels).reshape(-1).astype(np.int64), minlength=int(k))[:int(k)].astype(np.int64))

  def segment_sum(self, points, labels, k):
    """k_means_.py:91-95: per-cluster masked row sums (NumPy axis-0 order)."""
    self.launches += 1
    pts, lab = _np(points), _np(labels).reshape(-1)
    out = np.zeros((int(k), pts.shape[1]), pts.dtype)
    for i in range(int(k)):
      out[i] = pts[lab == i].sum(axis=0)
    return self._wrap(out)

  def concat(self, a, b, axis=0):
    """manipulation.py:51."""
    self.launches += 1
    return self._wrap(np.concatenate((_np(a), _np(b)), axis=axis))

  def reduce_axis(self, t, red_op, axis):
    """scan.py:25."""
    self.launches += 1
    return self._wrap({'SUM': np.sum, 'PROD': np.prod}[red_op](_np(t), axis=axis))

  def sort_axis(self, t, axis, indices=False):
    """np.sort / np.argsort of a tile (sort.py:68-69, :137-138), kind='stable' (the documented tie order)."""
    self.launches += 1
    x = _np(t)
    return self._wrap(np.argsort(x, axis, kind='stable') if indices else np.sort(x, axis, kind='stable'))

  def convolve(self, image, filters):
    """stencil.py:29-45, the loops as written (vectorised over images / filters / positions)."""
    self.launches += 1
    img, flt = _np(image), _np(filters)
    n, c, w, h = img.shape
    f, fc, fw, fh = flt.shape
    out = np.zeros((n, f, w, h), dtype=np.result_type(img.dtype, flt.dtype))
    for ci in range(c):
      for i in range(min(fw, w)):
        for j in range(min(fh, h)):
          out[:, :, :w - i, :h - j] += img[:, ci, i:, j:][:, None] * flt[:, ci, i, j][None, :, None, None]
    return self._wrap(out)

  def maxpool(self, region, pool_size, stride, out_shape):
    self.launches += 1
    x = _np(region)
    out = np.full(tuple(out_shape), -1e12, dtype=x.dtype)
    span = pool_size if pool_size < stride else stride      # pixel a belongs to window a // stride (stencil.py:68-70)
    for i in range(span):
      for j in range(span):
        part = x[:, :, i::stride, j::stride]
        out[:, :, :part.shape[2], :part.shape[3]] = np.maximum(out[:, :, :part.shape[2], :part.shape[3]], part)
    return self._wrap(out)

  def gather_rows(self, block, rows):
    self.launches += 1
    return self._wrap(_np(block)[np.asarray(rows, dtype=np.int64)])

  def cumscan(self, t, axis, product=False):
    """scan.py:63."""
    self.launches += 1
    x = _np(t)
    return self._wrap((np.cumprod if product else np.cumsum)(x, axis=axis).astype(x.dtype).reshape(x.shape))

  # -- sparse tiles: scipy.sparse, the library the reference's sparse tile bodies are written in
  # (spartan/array/sparse.pyx, tile.pyx:226-252, dot.py:212-240)
  def is_sparse(self, x):
    return sps.issparse(x)

  def sparse_blob(self, mat, dtype=None):
    if dtype is not None and mat.dtype != np.dtype(dtype):
      mat = mat.astype(dtype)
    return mat

  def sparse_to_host(self, b):
    return b

  def sparse_parts(self, b):
    c = sps.csr_matrix(b)
    c.sum_duplicates()
    c.sort_indices()
    return c.indptr.astype(np.int64), c.indices.astype(np.int32), np.ascontiguousarray(c.data)

  def sparse_parts_empty(self, shape, dtype, nnz):
    return np.empty(int(shape[0]) + 1, np.int64), np.empty(int(nnz), np.int32), np.empty(int(nnz), np.dtype(dtype))

  def sparse_from_parts(self, shape, dtype, parts):
    indptr, indices, data = parts
    return sps.csr_matrix((data.astype(dtype), indices, indptr), shape=tuple(shape))

  def sparse_empty(self, shape, dtype):
    return sps.coo_matrix(tuple(shape), dtype=dtype)     # tile.pyx:74-77

  def sparse_slice(self, b, slices):
    return b.tocsr()[slices]                              # tile.pyx:98 / sparse.pyx:198-213

  def sparse_paste(self, shape, dtype, pieces):
    """distarray.py:338-353: the pieces are written into a lil matrix of the region."""
    tgt = sps.lil_matrix(tuple(shape), dtype=dtype)
    for ul, p in pieces:
      if p.shape[0] and p.shape[1]:
        tgt[ul[0]:ul[0] + p.shape[0], ul[1]:ul[1] + p.shape[1]] = p
    return tgt.tocsr()

  def sparse_reduce(self, old, upd, reducer):
    return reducer(old, upd)                              # tile.pyx:246-247

  def sparse_update(self, old, ul, lr, upd, reducer):
    """sparse.pyx:246-286 compute_sparse_update (out of place: the box is replaced / reduced)."""
    data = old.tolil(copy=True)
    box = (slice(ul[0], lr[0]), slice(ul[1], lr[1]))
    data[box] = reducer(data[box].tocsr(), upd.tocsr()) if reducer is not None else upd
    return data.tocsr()

  def sparse_scatter(self, dst, ul, blob, mode, mask):
    """sparse.pyx:21-38 sparse_to_dense_update with REDUCE_ADD, entry by entry in COO order."""
    self.launches += 1
    d = dst
    m = mask if mask is not None else None
    coo = blob.tocoo()
    for r, c, v in zip(coo.row + ul[0], coo.col + ul[1], coo.data):
      if mode == 0 or (mode == 2 and not m[r, c]):
        d[r, c] = v
        if m is not None:
          m[r, c] = 1
      else:
        d[r, c] = d[r, c] + v

  def sparse_to_dense(self, b):
    return self.from_numpy(np.asarray(b.todense()))

  def dense_to_sparse(self, t, dtype=None):
    return sps.csr_matrix(np.asarray(t, dtype=dtype))

  def sparse_transpose(self, b):
    return b.transpose()

  def sparse_reshape(self, b, offset, shape):
    """reshape.py:181-193 on the COO triplets (linear position - offset, re-split by the new row length)."""
    coo = b.tocoo()
    lin = coo.row.astype(np.int64) * b.shape[1] + coo.col - offset
    keep = (lin >= 0) & (lin < shape[0] * shape[1])
    return sps.coo_matrix((coo.data[keep], (lin[keep] // shape[1], lin[keep] % shape[1])), shape=tuple(shape)).tocsr()

  def sparse_random(self, shape, density, dtype):
    return sps.rand(shape[0], shape[1], density=density, format='csr', dtype=dtype)

  def _as_device(self, t):
    return np.asarray(t)

  def synchronize(self):
    pass
