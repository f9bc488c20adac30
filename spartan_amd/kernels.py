"""Pythonic launchers over the C-ABI (spartan_amd/_hip.py) for tile blobs in HBM.

Operands are device arrays (spartan_amd/devarray.py: views of blobs the library's tile store owns); anything else
that can say where its bytes live in HBM -- `data_ptr()`, `shape`, `stride()`, `dtype` -- is accepted as well, so
a test may hand in memory from another allocator.  Launches go to devarray.current_stream().  No torch here.
"""
import ctypes as C

import numpy as np

from . import _hip, devarray
from ._hip import check
from .devarray import Event  # noqa: F401  (bench.py / tools time launches with it)

_KNOWN = {n: np.dtype(n) for n in ('float32', 'float64', 'int32', 'int64', 'bool', 'uint8')}


def np_dtype_of(t):
  """NumPy dtype of a device operand (a DevArray carries one; foreign tensors name theirs, e.g. 'torch.float32')."""
  dt = t.dtype
  if isinstance(dt, np.dtype):
    return dt
  name = str(dt).split('.')[-1]
  if name not in _KNOWN:
    raise TypeError('dtype %s is not supported by the HIP tile backend' % (dt,))
  return _KNOWN[name]


def _stream():
  return devarray.current_stream().ptr


def _require_device(*tensors):
  for t in tensors:
    if t is not None and not getattr(t, 'is_cuda', False):
      raise _hip.HipError('HIP tile kernels need device (HBM) operands; got %s' % (type(t).__name__,))


def _elsize(t):
  return t.element_size() if hasattr(t, 'element_size') else t.itemsize


class Workspace(object):
  """Grow-only scratch blob for reduction partials (one per device)."""

  def __init__(self):
    self.buf = None

  def get(self, nbytes, device=None):
    if self.buf is None or self.buf.numel() < nbytes:
      self.buf = devarray.empty((max(int(nbytes), 1 << 20),), np.uint8)
    return self.buf


_ws = Workspace()


def map_fused(prog, inputs, out):
  """Launch the fused map `prog` over dense tensors; `out` is written."""
  _require_device(out, *inputs)
  ptrs = _hip.ptr_array([t.data_ptr() for t in inputs])
  check(_hip.lib().sp_map_fused(C.byref(prog), ptrs, C.c_void_p(out.data_ptr()), _stream()))
  return out


_reduce_ws_bytes = {}


def reduce(prog, inputs, red_op, outer, axis_len, inner, out):
  _require_device(out, *inputs)
  lib = _hip.lib()
  key = (prog.cls, outer, axis_len, inner)
  need = _reduce_ws_bytes.get(key)
  if need is None:                         # (a pure function of the plan's geometry: asked once per shape)
    if len(_reduce_ws_bytes) > 4096:
      _reduce_ws_bytes.clear()
    need = _reduce_ws_bytes[key] = lib.sp_reduce_workspace_bytes(prog.cls, outer, axis_len, inner)
  ws = _ws.get(need, out.device)
  ptrs = _hip.ptr_array([t.data_ptr() for t in inputs])
  check(lib.sp_reduce(C.byref(prog), ptrs, _hip.RED[red_op] if isinstance(red_op, str) else red_op,
                      outer, axis_len, inner, C.c_void_p(out.data_ptr()),
                      _hip.sp_dtype(np_dtype_of(out)), C.c_void_p(ws.data_ptr()), ws.numel(), _stream()))
  return out


def reduce_warm(prog, inputs, red_op, outer, axis_len, inner, out_dtype):
  """What the first reduce() of this program would set up besides the launch: the partials' workspace at its size."""
  need = _hip.lib().sp_reduce_workspace_bytes(prog.cls, outer, axis_len, inner)
  _ws.get(need)


def argreduce(prog, inputs, which, outer, axis_len, inner, index_offset, nan_index, out_idx, out_val=None):
  _require_device(out_idx, out_val, *inputs)
  lib = _hip.lib()
  need = lib.sp_argreduce_workspace_bytes(prog.cls, outer, axis_len, inner)
  ws = _ws.get(need, out_idx.device)
  ptrs = _hip.ptr_array([t.data_ptr() for t in inputs])
  assert np_dtype_of(out_idx) == np.int64
  check(lib.sp_argreduce(C.byref(prog), ptrs, which, outer, axis_len, inner, int(index_offset),
                         int(nan_index), C.c_void_p(out_idx.data_ptr()),
                         C.c_void_p(out_val.data_ptr() if out_val is not None else 0),
                         C.c_void_p(ws.data_ptr()), ws.numel(), _stream()))
  return out_idx


def update(dst, ul, lr, src, reducer, mask_mode, mask=None):
  """Tile.merge: dst[ul:lr] = merge(dst[ul:lr], src) with the tile's mask state."""
  _require_device(dst, src, mask)
  nd = dst.dim()
  check(_hip.lib().sp_update(
      C.c_void_p(dst.data_ptr()), _hip.sp_dtype(np_dtype_of(dst)), _hip.i64_array(dst.shape), nd,
      _hip.i64_array(ul), _hip.i64_array(lr), C.c_void_p(src.data_ptr()), _hip.sp_dtype(np_dtype_of(src)),
      _hip.REDUCER[reducer] if isinstance(reducer, str) else reducer, mask_mode,
      C.c_void_p(mask.data_ptr() if mask is not None else 0), _stream()))
  return dst


def slice_copy(dst, dst_offset, dst_strides, src, src_offset, src_strides, shape):
  """Strided box copy between two blobs of the same element size (strides/offsets in elements)."""
  _require_device(dst, src)
  es = _elsize(dst)
  assert es == _elsize(src)
  nd = len(shape)
  check(_hip.lib().sp_slice_copy(
      C.c_void_p(dst.data_ptr() + int(dst_offset) * es), _hip.i64_array(dst_strides),
      C.c_void_p(src.data_ptr() + int(src_offset) * es), _hip.i64_array(src_strides),
      _hip.i64_array(shape), nd, es, _stream()))
  return dst


def gemm_f32(a, b, c, accumulate=False):
  """c (+)= a . b for 2-D row-major fp32 (or, all three, fp64) tensors (inner stride 1)."""
  _require_device(a, b, c)
  adt = np_dtype_of(a)
  assert adt == np_dtype_of(b) == np_dtype_of(c) and adt in (np.float32, np.float64)
  lib = _hip.lib()
  dt = _hip.SP_F32 if adt == np.float32 else _hip.SP_F64
  M, K = a.shape
  K2, N = b.shape
  assert K == K2 and tuple(c.shape) == (M, N), (a.shape, b.shape, c.shape)
  assert a.stride(1) == 1 and b.stride(1) == 1 and c.stride(1) == 1
  need = lib.sp_gemm_workspace_bytes(dt, M, N, K)     # > 0: few output tiles, long contraction -> split-K
  ws = _ws.get(need, a.device) if need else None
  check(lib.sp_gemm_ws(dt, C.c_void_p(a.data_ptr()), a.stride(0) if M > 1 else max(K, 1),
                       C.c_void_p(b.data_ptr()), b.stride(0) if K > 1 else max(N, 1),
                       C.c_void_p(c.data_ptr()), c.stride(0) if M > 1 else max(N, 1),
                       M, N, K, 1 if accumulate else 0, C.c_void_p(ws.data_ptr() if need else 0),
                       ws.numel() if need else 0, _stream()))
  return c


def rowdot_colsum(x, w, y, out, accumulate=False):
  """out[c] (+)= sum_i x[i, c] * (x[i, :] . w - y[i]) in one pass over the fp32 row tile x [n, d] (y may be None);
  False when the operands do not meet sp_rowdot_colsum_f32's layout (the caller then takes the two-launch form)."""
  _require_device(x, w, out)
  n, d = x.shape
  lib = _hip.lib()
  need = lib.sp_rowdot_colsum_workspace_bytes(n, d) if n else 256
  ldx = x.stride(0) if n > 1 else d
  if (not need or x.stride(1) != 1 or ldx % 4 or (x.data_ptr() | w.data_ptr() | out.data_ptr()) % 16
      or any(np_dtype_of(t) != np.float32 for t in (x, w, out) + ((y,) if y is not None else ()))):
    return False
  ws = _ws.get(need, x.device)
  check(lib.sp_rowdot_colsum_f32(C.c_void_p(x.data_ptr()), ldx, n, d, C.c_void_p(w.data_ptr()),
                                 C.c_void_p(y.data_ptr() if y is not None else 0),
                                 (y.stride(0) if y is not None and n > 1 else 1), C.c_void_p(out.data_ptr()),
                                 1 if accumulate else 0, C.c_void_p(ws.data_ptr()), ws.numel(), _stream()))
  return True


gemm = gemm_f32   # dtype-dispatching: fp32 -> sp_gemm_f32, fp64 -> sp_gemm_f64


def _ld(t):
  """Row stride (elements) of a 2-D tensor whose inner stride is 1."""
  assert t.dim() == 2 and (t.shape[1] <= 1 or t.stride(1) == 1), (t.shape, t.stride())
  return t.stride(0) if t.shape[0] > 1 else max(t.shape[1], 1)


def prepare_points(points):
  """What the split tier of nearest_center reads of fp32 points -- two bf16 images and |x|^2 per point -- as a
  buffer to hand to nearest_center(prepared=...) for as long as the points are not written (one k-means fit)."""
  _require_device(points)
  n, d = points.shape
  assert np_dtype_of(points) == np.float32
  lib = _hip.lib()
  out = devarray.empty((max(int(lib.sp_kmeans_points_prepared_bytes(n, d)), 1),), np.uint8)
  check(lib.sp_kmeans_points_prepare(C.c_void_p(points.data_ptr()), _ld(points), n, d, C.c_void_p(out.data_ptr()),
                                     out.numel(), _stream()))
  return out


def nearest_center(points, centers, labels, tier=_hip.NEAREST_AUTO, prepared=None):
  """labels[i] = argmin_c |points[i] - centers[c]| (cdist + argmin; k_means_.py:61-66)."""
  _require_device(points, centers, labels)
  n, d = points.shape
  k, d2 = centers.shape
  assert d == d2 and np_dtype_of(labels) == np.int64 and labels.numel() == n and labels.is_contiguous()
  lib = _hip.lib()
  if prepared is not None:
    ws = _ws.get(lib.sp_nearest_center_prepared_workspace_bytes(n, k, d), points.device)
    check(lib.sp_nearest_center_prepared(C.c_void_p(points.data_ptr()), _hip.sp_dtype(np_dtype_of(points)), _ld(points),
                                         C.c_void_p(prepared.data_ptr()),
                                         C.c_void_p(centers.data_ptr()), _hip.sp_dtype(np_dtype_of(centers)), _ld(centers),
                                         n, k, d, C.c_void_p(labels.data_ptr()), tier, C.c_void_p(ws.data_ptr()),
                                         ws.numel(), _stream()))
    return labels
  need = lib.sp_nearest_center_workspace_bytes(n, k, d)
  ws = _ws.get(need, points.device)
  check(lib.sp_nearest_center(C.c_void_p(points.data_ptr()), _hip.sp_dtype(np_dtype_of(points)), _ld(points),
                              C.c_void_p(centers.data_ptr()), _hip.sp_dtype(np_dtype_of(centers)), _ld(centers),
                              n, k, d, C.c_void_p(labels.data_ptr()), tier, C.c_void_p(ws.data_ptr()),
                              ws.numel(), _stream()))
  return labels


def bincount(labels, k, counts):
  """counts[:k] = np.bincount(labels, minlength=k) (k_means_.py:69-72)."""
  _require_device(labels, counts)
  assert np_dtype_of(labels) == np.int64 and np_dtype_of(counts) == np.int64 and counts.numel() == k
  assert labels.is_contiguous() and counts.is_contiguous()
  check(_hip.lib().sp_bincount_i64(C.c_void_p(labels.data_ptr()), labels.numel(), k,
                                   C.c_void_p(counts.data_ptr()), _stream()))
  return counts


def segment_sum(points, labels, k, out, counts=None):
  """out[c] = points[labels == c].sum(axis=0) (k_means_.py:75-97); counts (int64 [k], optional) = np.bincount(labels,
  minlength=k): the counting sort inside has the number anyway."""
  _require_device(points, labels, out, counts)
  n, d = points.shape
  assert np_dtype_of(labels) == np.int64 and labels.numel() == n and labels.is_contiguous()
  assert np_dtype_of(out) == np_dtype_of(points) and tuple(out.shape) == (k, d) and out.is_contiguous()
  lib = _hip.lib()
  need = lib.sp_segment_sum_workspace_bytes(n, k, d)
  ws = _ws.get(need, points.device)
  if counts is not None:
    assert np_dtype_of(counts) == np.int64 and counts.numel() == k and counts.is_contiguous()
  check(lib.sp_segment_sum_counts(C.c_void_p(points.data_ptr()), _hip.sp_dtype(np_dtype_of(points)), _ld(points),
                                  C.c_void_p(labels.data_ptr()), n, k, d, C.c_void_p(out.data_ptr()),
                                  C.c_void_p(counts.data_ptr() if counts is not None else 0),
                                  C.c_void_p(ws.data_ptr()), ws.numel(), _stream()))
  return out


RANDOM_KINDS = {'uniform': 0, 'normal': 1, 'randint': 2}


def random_fill(out, kind, seed, offset, lo=0, hi=1):
  """Philox fill of a dense tensor: element i = f(seed, offset + i) (srandom.py:38-55)."""
  _require_device(out)
  assert out.is_contiguous()
  check(_hip.lib().sp_random_fill(C.c_void_p(out.data_ptr()), _hip.sp_dtype(np_dtype_of(out)), out.numel(),
                                  RANDOM_KINDS[kind], int(seed) & (2**64 - 1), int(offset) & (2**64 - 1),
                                  int(lo), int(hi), _stream()))
  return out


def cumscan(src, out, axis, product=False):
  """out = np.cumsum / np.cumprod(src, axis) for dense tensors of one dtype (scan.py:42-63)."""
  _require_device(src, out)
  assert src.is_contiguous() and out.is_contiguous() and np_dtype_of(src) == np_dtype_of(out) and tuple(src.shape) == tuple(out.shape)
  shape = tuple(src.shape)
  outer = int(np.prod(shape[:axis], dtype=np.int64))
  inner = int(np.prod(shape[axis + 1:], dtype=np.int64))
  check(_hip.lib().sp_cumscan(C.c_void_p(src.data_ptr()), C.c_void_p(out.data_ptr()), _hip.sp_dtype(np_dtype_of(src)),
                              outer, shape[axis], inner, 1 if product else 0, _stream()))
  return out


def sort_rows(src, values=True, indices=False):
  """Stable sort of every row of a contiguous [rows, cols] tensor along its last axis (sort.py:68-69, :137-138):
  returns (sorted values or None, int64 argsort or None)."""
  _require_device(src)
  assert src.dim() == 2 and src.is_contiguous()
  rows, cols = src.shape
  vals = devarray.empty((rows, cols), np_dtype_of(src)) if values else None
  idx = devarray.empty((rows, cols), np.int64) if indices else None
  if rows and cols:
    lib = _hip.extras()        # (sort is outside the tile path: libspartan_hip_extras.so)
    dt = _hip.sp_dtype(np_dtype_of(src))
    ws = _ws.get(lib.sp_sort_rows_workspace_bytes(dt, rows, cols), src.device)
    check(lib.sp_sort_rows(C.c_void_p(src.data_ptr()), dt, rows, cols, C.c_void_p(vals.data_ptr() if values else 0),
                           C.c_void_p(idx.data_ptr() if indices else 0), C.c_void_p(ws.data_ptr()), ws.numel(),
                           _stream()))
  return vals, idx


def gather_rows(src, idx):
  """src[idx] along axis 0 for a contiguous tensor and a device int64 index vector (filter.py:50-75)."""
  _require_device(src, idx)
  assert src.is_contiguous() and np_dtype_of(idx) == np.int64 and idx.is_contiguous()
  n = int(idx.numel())
  row = int(np.prod(src.shape[1:], dtype=np.int64)) * _elsize(src)
  out = devarray.empty((n,) + tuple(src.shape[1:]), np_dtype_of(src))
  if row % 4:
    raise _hip.HipError('gather_rows: rows of %d bytes (need a multiple of 4)' % row)
  check(_hip.lib().sp_gather_rows(C.c_void_p(src.data_ptr()), row, int(src.shape[0]), C.c_void_p(idx.data_ptr()), n, row,
                                  C.c_void_p(out.data_ptr()), _stream()))
  return out


def stream_copy(dst, src, nbytes=None, max_workgroups=0, stream=None):
  """dst <- src (contiguous bytes); max_workgroups > 0 bounds the grid (a transfer that takes a limited share of
  the CUs, see sp_stream_copy_wg)."""
  _require_device(dst, src)
  n = src.numel() * _elsize(src) if nbytes is None else int(nbytes)
  check(_hip.lib().sp_stream_copy_wg(C.c_void_p(dst.data_ptr()), C.c_void_p(src.data_ptr()), n, int(max_workgroups),
                                     (stream or devarray.current_stream()).ptr))
  return dst
