"""Sparse tiles in HBM.

The reference stores a sparse tile as a scipy.sparse matrix of whatever format the producing mapper
chose and converts it on every use (spartan/array/tile.pyx:149-156, sparse.pyx:232-242, dot.py:212-216).
Here a sparse tile is a `CsrTile`: canonical CSR held as three device arrays in HBM (int64 indptr, int32
column indices ascending inside a row without duplicates, f32 / f64 values).  scipy appears only at the
host boundary -- what a user mapper yields (upload) and what `glom()` hands back (download).

Every structural operation is "edit a COO list on the device, then sp_coo_to_csr"
(spartan_amd/csrc/sparse.hip); the numeric hot op is sp_csr_spmm.
"""
import ctypes as C
import os

import numpy as np

from . import _hip, kernels
from . import devarray as D
from ._hip import check
from .kernels import _stream, _ws, np_dtype_of, _ld

_VALUE_DTYPES = (np.dtype(np.float32), np.dtype(np.float64))
_backend = None


def _be():
  """Value conversions / scaling go through the fused map kernel of the HIP backend."""
  global _backend
  if _backend is None:
    from .backend_hip import HipBackend
    _backend = HipBackend()
  return _backend


def _cast(vals, dtype):
  if np_dtype_of(vals) == np.dtype(dtype):
    return vals
  return _be().astype(vals, dtype)


def _p(t):
  return C.c_void_p(t.data_ptr() if t is not None and t.numel() else 0)


def _cat(parts):
  """The 1-D arrays of `parts` (one dtype) one after the other, as contiguous copies into a new array."""
  parts = [p for p in parts if p.numel()] or parts[:1]
  dt = np_dtype_of(parts[0])
  out = D.empty((sum(int(p.numel()) for p in parts),), dt)
  at = 0
  for p in parts:
    n = int(p.numel())
    if n:
      kernels.slice_copy(out, at, (1,), p, 0, (p.stride(0),), (n,))
      at += n
  return out


class CsrTile(object):
  """Device blob of a sparse tile."""
  __slots__ = ('shape', 'dtype', 'indptr', 'indices', 'data', '_plan', '_block_plan', '_transposed')
  is_sparse_tile = True

  def __init__(self, shape, dtype, indptr, indices, data):
    self.shape = (int(shape[0]), int(shape[1]))
    self.dtype = np.dtype(dtype)
    self.indptr = indptr
    self.indices = indices
    self.data = data
    self._plan = None      # sp_csr_spmv_plan output, made on the first matrix x vector product and kept
    self._block_plan = None   # spmv_block_plan's: the column-blocked copy of the entries, or False (none for this tile)
    self._transposed = None   # the transposed tile, made on first use and kept (tiles are immutable)

  @property
  def nnz(self):
    return int(self.indices.numel())

  @property
  def device(self):
    return self.indptr.device

  @property
  def size(self):
    return self.shape[0] * self.shape[1]

  def __repr__(self):
    return 'CsrTile(%s, %s, nnz=%d)' % (self.shape, self.dtype, self.nnz)


def spmv_plan(t):
  """First row starting in each 2048-entry chunk (csrc/sparse.hip): an analysis of the STRUCTURE, computed once
  per tile -- tiles are immutable -- and reused by every product with a vector."""
  if t._plan is None:
    lib = _hip.lib()
    plan = D.empty((int(lib.sp_csr_spmv_plan_entries(t.nnz)),), np.int64)
    check(lib.sp_csr_spmv_plan(t.shape[0], t.nnz, _p(t.indptr), _p(plan), _stream()))
    t._plan = plan
  return t._plan


def spmv_block_plan(t):
  """The column-blocked plan of csrc/spmv_blocked.hip (fp32 tiles with sorted rows, large enough to pay): built on
  the tile's first product with a vector and kept; False when the tile has none and sp_csr_spmm's kernels stay."""
  if t._block_plan is None:
    lib = _hip.lib()
    dt = _hip.sp_dtype(t.dtype) if t.dtype in (np.float32, np.float64) else -1
    need = int(lib.sp_csr_spmv_blockplan_bytes(dt, t.shape[0], t.shape[1], t.nnz)) if dt >= 0 and os.environ.get('SP_SPMV_BLOCKED', '1') != '0' else 0
    t._block_plan = False
    if need:
      plan = D.empty((need,), np.uint8)
      check(lib.sp_csr_spmv_blockplan(dt, t.shape[0], t.shape[1], t.nnz, _p(t.indptr), _p(t.indices), _p(t.data), _p(plan),
                                      need, _stream()))
      if int(plan[:8].numpy().view(np.int64)[0]) == 1:      # (0: a row not sorted by column)
        t._block_plan = plan
  return t._block_plan


def _check_dtype(dtype):
  dtype = np.dtype(dtype)
  if dtype not in _VALUE_DTYPES:
    raise NotImplementedError('sparse tiles hold float32 / float64 values on the device (got %s)' % dtype)
  return dtype


def empty(shape, dtype, device):
  dtype = _check_dtype(dtype)
  return CsrTile(shape, dtype, D.zeros((int(shape[0]) + 1,), np.int64), D.empty((0,), np.int32), D.empty((0,), dtype))


def from_coo(shape, dtype, rows, cols, vals):
  """Device COO list (int32 rows / cols, values) -> canonical CsrTile; rows < 0 are dropped and equal
  coordinates added in list order."""
  dtype = _check_dtype(dtype)
  m, n = int(shape[0]), int(shape[1])
  nnz = int(rows.numel())
  dev = rows.device
  if nnz == 0:
    return empty((m, n), dtype, dev)
  assert np_dtype_of(rows) == np.int32 and np_dtype_of(cols) == np.int32 and rows.is_contiguous() and cols.is_contiguous()
  vals = _cast(vals.contiguous(), dtype)
  lib = _hip.lib()
  indptr = D.empty((m + 1,), np.int64)
  indices = D.empty((nnz,), np.int32)
  out = D.empty((nnz,), dtype)
  need = lib.sp_coo_to_csr_workspace_bytes(nnz)
  ws = _ws.get(need, dev)
  check(lib.sp_coo_to_csr(_hip.sp_dtype(dtype), m, n, nnz, _p(rows), _p(cols), _p(vals), _p(indptr), _p(indices),
                          _p(out), _p(ws), ws.numel(), _stream()))
  kept = int(indptr[m].item())
  return CsrTile((m, n), dtype, indptr, indices[:kept], out[:kept])


def from_scipy(mat, device, dtype=None):
  """Upload what a mapper produced (any scipy.sparse format): the COO triplets go to the device as they
  are and sp_coo_to_csr canonicalises them there."""
  coo = mat.tocoo()
  dtype = _check_dtype(coo.dtype if dtype is None else dtype)
  if coo.shape[0] >= 2 ** 31 or coo.shape[1] >= 2 ** 31:
    raise NotImplementedError('sparse tile dimension exceeds the int32 index range')
  rows = D.from_numpy(np.ascontiguousarray(coo.row, dtype=np.int32))
  cols = D.from_numpy(np.ascontiguousarray(coo.col, dtype=np.int32))
  vals = D.from_numpy(np.ascontiguousarray(coo.data, dtype=dtype))
  return from_coo(coo.shape, dtype, rows, cols, vals)


def from_dense(x, dtype=None):
  """The non-zero cells of a dense (m, n) device array as a tile (what scipy makes of `csr_matrix(dense)`): row and
  column numbers are built on the device by broadcasting two small index vectors, cells equal to zero get row -1,
  which sp_coo_to_csr drops."""
  m, n = int(x.shape[0]), int(x.shape[1])
  dtype = _check_dtype(np_dtype_of(x) if dtype is None else dtype)
  if m == 0 or n == 0:
    return empty((m, n), dtype, None)
  be = _be()
  x = _cast(x.contiguous(), dtype)
  row_of = D.from_numpy(np.arange(m, dtype=np.int32).reshape(m, 1))
  col_of = D.from_numpy(np.arange(n, dtype=np.int32).reshape(1, n))
  stored = be.evaluate_fn(np.not_equal, [x, dtype.type(0)], {}, (m, n))
  rows = be.evaluate_fn(np.where, [stored, row_of, np.int32(-1)], {}, (m, n))
  cols = be.evaluate_fn(np.add, [col_of, D.zeros((m, 1), np.int32)], {}, (m, n))
  return from_coo((m, n), dtype, rows.reshape(m * n).contiguous(), cols.reshape(m * n).contiguous(), x.reshape(m * n))


def to_scipy(t):
  import scipy.sparse
  return scipy.sparse.csr_matrix((t.data.numpy(), t.indices.numpy(), t.indptr.numpy()),
                                 shape=t.shape)


def rows_of(t):
  """int32 row index of every stored entry (CSR -> COO)."""
  rows = D.empty((t.nnz,), np.int32)
  if t.nnz:
    check(_hip.lib().sp_csr_rows(t.shape[0], t.nnz, _p(t.indptr), _p(rows), _stream()))
  return rows


def _box(rows, cols, r0, r1, c0, c1, dr, dc, drop_inside):
  if rows.numel():
    check(_hip.lib().sp_coo_box(rows.numel(), _p(rows), _p(cols), r0, r1, c0, c1, dr, dc, 1 if drop_inside else 0,
                                _stream()))


def slice_box(t, r0, r1, c0, c1):
  """t[r0:r1, c0:c1] (sparse.pyx:198-230 slice / slice_coo)."""
  if (r0, r1, c0, c1) == (0, t.shape[0], 0, t.shape[1]):
    return t
  rows, cols = rows_of(t), t.indices.clone()
  _box(rows, cols, r0, r1, c0, c1, -r0, -c0, False)
  return from_coo((r1 - r0, c1 - c0), t.dtype, rows, cols, t.data)


def reshape_rect(t, offset, shape):
  """The entries of `t` at linear positions [offset, offset + prod(shape)) as a tile of `shape`
  (reshape.py:181-193)."""
  rows, cols = rows_of(t), t.indices.clone()
  if t.nnz:
    check(_hip.lib().sp_coo_reshape(t.nnz, _p(rows), _p(cols), t.shape[1], int(offset), int(shape[0]), int(shape[1]),
                                    _stream()))
  return from_coo(shape, t.dtype, rows, cols, t.data)


def transpose(t):
  """The transposed tile (column sums, dense x sparse and Transpose views ask for it repeatedly: kept on the tile)."""
  if t._transposed is None:
    tt = from_coo((t.shape[1], t.shape[0]), t.dtype, t.indices.clone(), rows_of(t), t.data)
    t._transposed = tt        # (one direction only: a cycle would keep both alive until the cyclic GC runs)
  return t._transposed


def add(a, b, sign=1):
  """a + sign * b (scipy's sparse + / - behind np.add / np.subtract on two sparse tiles)."""
  assert a.shape == b.shape
  dtype = np.result_type(a.dtype, b.dtype)
  rows = _cat([rows_of(a), rows_of(b)])
  cols = _cat([a.indices, b.indices])
  bv = _cast(b.data, dtype)
  if sign < 0:
    bv = scale_values(bv, -1.0)
  vals = _cat([_cast(a.data, dtype), bv])
  return from_coo(a.shape, dtype, rows, cols, vals)


def scale_values(vals, alpha):
  """alpha * vals (same dtype) through the fused map kernel."""
  if vals.numel() == 0:
    return vals
  return _be().evaluate_fn(np.multiply, [vals, np_dtype_of(vals).type(alpha)], {}, tuple(vals.shape))


def scaled(t, alpha):
  return CsrTile(t.shape, t.dtype, t.indptr, t.indices, scale_values(t.data, alpha))


def paste(shape, dtype, pieces):
  """Assemble a sparse tile of `shape` from [(row0, col0, CsrTile)] (disjoint boxes)."""
  rows, cols, vals = [], [], []
  dev = None
  for r0, c0, p in pieces:
    dev = p.device
    if p.nnz == 0:
      continue
    r, c = rows_of(p), p.indices.clone()
    _box(r, c, 0, p.shape[0], 0, p.shape[1], r0, c0, False)
    rows.append(r)
    cols.append(c)
    vals.append(_cast(p.data, dtype))
  if not rows:
    return empty(shape, dtype, dev)
  return from_coo(shape, dtype, _cat(rows), _cat(cols), _cat(vals))


def update_box(old, r0, r1, c0, c1, upd, add_to_old):
  """compute_sparse_update (sparse.pyx:246-286): the box [r0:r1, c0:c1] of `old` becomes `upd`
  (add_to_old False: the old entries of the box are dropped) or old[box] + upd."""
  dtype = old.dtype
  orow, ocol = rows_of(old), old.indices.clone()
  if not add_to_old:
    _box(orow, ocol, r0, r1, c0, c1, 0, 0, True)
  urow, ucol = rows_of(upd), upd.indices.clone()
  _box(urow, ucol, 0, upd.shape[0], 0, upd.shape[1], r0, c0, False)
  return from_coo(old.shape, dtype, _cat([orow, urow]), _cat([ocol, ucol]), _cat([old.data, _cast(upd.data, dtype)]))


def spmm(a, b, out=None, accumulate=False, plan=True):
  """a (CsrTile [m, k]) x b (dense [k] / [k, n]) -> dense [m] / [m, n]."""
  vec = b.dim() == 1
  b2 = b.reshape(-1, 1) if vec else b
  if b2.stride(-1) != 1:
    b2 = b2.contiguous()
  if b2.shape[0] != a.shape[1]:
    raise ValueError('objects are not aligned')
  dtype = np.result_type(a.dtype, np_dtype_of(b2))
  av = _cast(a.data, dtype)
  b2 = _cast(b2, dtype)
  m, n = a.shape[0], int(b2.shape[1])
  # both kernels write rows of n consecutive values: a caller's strided `out` (a column of a wider matrix, a view
  # with a step) is computed into a packed buffer and pasted over itself afterwards
  strided_out = None
  if out is None:
    out = D.empty((m, n), dtype)
  elif not out.is_contiguous():
    strided_out, out = out, (out.contiguous() if accumulate else D.empty((m, n), dtype))
  if m and n:
    lib = _hip.lib()
    if n == 1 and plan and dtype == a.dtype:
      bp = spmv_block_plan(a)
      if bp is not False:
        # the blocked kernel reads x as ONE packed vector (include/spartan_hip.h: d_x contiguous): a (k, 1) view of
        # a wider matrix -- X[:, 0:1], what Tile.get hands out -- or a vector with a step is packed first
        xv = b2 if b2.is_contiguous() else b2.copy()
        if xv.data_ptr() % 16:
          xv = xv.copy()
        check(lib.sp_csr_spmv_blocked(_hip.sp_dtype(dtype), m, a.shape[1], a.nnz, _p(a.indptr), _p(bp),
                                      C.c_void_p(xv.data_ptr()), C.c_void_p(out.data_ptr()), 1, 1 if accumulate else 0,
                                      _stream()))
        return _spmm_result(out, strided_out, m, vec)
    ws = _ws.get(lib.sp_csr_spmm_workspace_bytes(a.nnz, n), a.device)
    check(lib.sp_csr_spmm(_hip.sp_dtype(dtype), m, a.shape[1], n, a.nnz, _p(a.indptr), _p(a.indices), _p(av),
                          C.c_void_p(b2.data_ptr()), _ld(b2) if b2.shape[0] > 1 else max(n, 1),
                          C.c_void_p(out.data_ptr()), n, 1 if accumulate else 0,
                          _p(spmv_plan(a)) if plan and n == 1 and a.nnz else C.c_void_p(0), _p(ws), ws.numel(), _stream()))
  return _spmm_result(out, strided_out, m, vec)


def _spmm_result(out, strided_out, m, vec):
  if strided_out is not None:
    D._be().paste(strided_out, tuple(slice(0, k) for k in strided_out.shape), out.reshape(strided_out.shape))
    out = strided_out
  return out.reshape(m) if vec else out


def row_sums(t):
  out = D.empty((t.shape[0], 1), np_dtype_of(t.data))
  if t.shape[0]:
    lib = _hip.lib()
    ws = _ws.get(lib.sp_csr_spmm_workspace_bytes(t.nnz, 1), t.device)
    check(lib.sp_csr_spmm(_hip.sp_dtype(t.dtype), t.shape[0], t.shape[1], 1, t.nnz, _p(t.indptr), _p(t.indices),
                          _p(t.data), C.c_void_p(0), 1, C.c_void_p(out.data_ptr()), 1, 0,
                          _p(spmv_plan(t)) if t.nnz else C.c_void_p(0), _p(ws), ws.numel(), _stream()))
  return out.reshape(-1)


def scatter(t, out, row0=0, col0=0, mode=0, mask=None):
  """Write `t` into the box of the dense 2-D tensor `out` at (row0, col0): mode 0 assign, 1 add,
  2 the reference's masked first-write rule (sparse.pyx:21-38)."""
  assert out.dim() == 2 and out.stride(1) == 1 and np_dtype_of(out) == t.dtype
  if t.nnz:
    check(_hip.lib().sp_csr_scatter(_hip.sp_dtype(t.dtype), t.shape[0], t.nnz, _p(t.indptr), _p(t.indices), _p(t.data),
                                    C.c_void_p(out.data_ptr()), out.stride(0) if out.shape[0] > 1 else out.shape[1],
                                    row0, col0, C.c_void_p(mask.data_ptr() if mask is not None else 0),
                                    (mask.stride(0) if mask.shape[0] > 1 else mask.shape[1]) if mask is not None else 0,
                                    mode, _stream()))
  return out


def to_dense(t):
  out = D.zeros(t.shape, t.dtype)
  return scatter(t, out)


def spgemm(a, b):
  """a x b, both sparse -> sparse (expand, sort, compress)."""
  if a.shape[1] != b.shape[0]:
    raise ValueError('objects are not aligned')
  dtype = np.result_type(a.dtype, b.dtype)
  dev = a.device
  shape = (a.shape[0], b.shape[1])
  if a.nnz == 0 or b.nnz == 0:
    return empty(shape, dtype, dev)
  lib = _hip.lib()
  offs = D.empty((a.nnz + 1,), np.int32)
  total = D.empty((1,), np.int64)
  need = lib.sp_spgemm_count_workspace_bytes(a.nnz)
  ws = _ws.get(need, dev)
  check(lib.sp_spgemm_count(a.nnz, _p(a.indices), _p(b.indptr), _p(offs), _p(total), _p(ws), ws.numel(), _stream()))
  n_prod = int(total.item())
  if n_prod >= 2 ** 31 - 4096:
    raise NotImplementedError('sparse x sparse: %d products do not fit one tile expansion' % n_prod)
  if n_prod == 0:
    return empty(shape, dtype, dev)
  rows = D.empty((n_prod,), np.int32)
  cols = D.empty((n_prod,), np.int32)
  vals = D.empty((n_prod,), dtype)
  av, bv = _cast(a.data, dtype), _cast(b.data, dtype)
  check(lib.sp_spgemm_expand(_hip.sp_dtype(dtype), a.shape[0], a.nnz, _p(a.indptr), _p(a.indices), _p(av), _p(b.indptr),
                             _p(b.indices), _p(bv), _p(offs), _p(rows), _p(cols), _p(vals), _stream()))
  return from_coo(shape, dtype, rows, cols, vals)
