"""ctypes binding of libspartan_hip.so (see include/spartan_hip.h).

This is the thin host side of the drop-in boundary: plain pointers and sizes
only.  Loading FAILS LOUDLY when the library is missing -- there is no CPU
fallback in the product path (the NumPy restatement under oracle/ is test
infrastructure and is never imported from here).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SPARTAN_HIP_LIB: load another build of the same library (kernel A/B experiments, tools/)
LIB_PATH = os.environ.get('SPARTAN_HIP_LIB') or os.path.join(_HERE, 'csrc', 'libspartan_hip.so')

# ---- enums (mirror include/spartan_hip.h) ---------------------------------
SP_F32, SP_F64, SP_I32, SP_I64, SP_BOOL, SP_U8 = range(6)
SP_MAX_INPUTS, SP_MAX_INSTR, SP_MAX_CONSTS, SP_MAX_DIMS, SP_NREG = 8, 64, 16, 4, 8
SP_BLOB_MAX_DIMS, SP_COMM_UID_BYTES = 8, 128

OP = dict(
    NOP=0, CONST=1, IOTA=2, MOV=3,
    ADD=10, SUB=11, MUL=12, DIV=13, FLOORDIV=14, MOD=15, FMOD=16, POW=17, MAX=18, MIN=19,
    EQ=20, NE=21, LT=22, LE=23, GT=24, GE=25, LAND=26, LOR=27, LXOR=28, LNOT=29,
    NEG=30, ABS=31, SQRT=32, SQUARE=33, EXP=34, LOG=35, RECIP=36, SIGN=37, FLOOR=38, CEIL=39,
    TANH=40, NORM_CDF=41, WHERE=45, TO_F32=50, TO_I32=51, TO_I64=52, TO_BOOL=53, TO_U8=54,
    ADDC=60, SUBC=61, RSUBC=62, MULC=63, DIVC=64, RDIVC=65, MAXC=66, MINC=67)

RED = dict(SUM=0, PROD=1, MAX=2, MIN=3, AND=4, OR=5)
REDUCER = dict(NONE=0, ADD=1, MUL=2, MAX=3, MIN=4, AND=5, OR=6)
MASK_ALL_CLEAR, MASK_ALL_SET, MASK_ARRAY = 0, 1, 2
NEAREST_AUTO, NEAREST_EXACT, NEAREST_FUSED, NEAREST_FUSED_UNCHECKED, NEAREST_SPLIT, NEAREST_SPLIT_UNCHECKED = 0, 1, 2, 3, 4, 5

_NP2SP = {
    np.dtype(np.float32): SP_F32, np.dtype(np.float64): SP_F64, np.dtype(np.int32): SP_I32,
    np.dtype(np.int64): SP_I64, np.dtype(np.bool_): SP_BOOL, np.dtype(np.uint8): SP_U8,
}
_SP2NP = {v: k for k, v in _NP2SP.items()}


def sp_dtype(dt):
  dt = np.dtype(dt)
  if dt not in _NP2SP:
    raise TypeError('dtype %s is not supported by the HIP tile backend '
                    '(supported: float32 float64 int32 int64 bool uint8)' % dt)
  return _NP2SP[dt]


def np_dtype(code):
  return _SP2NP[code]


class sp_instr(C.Structure):
  _fields_ = [('op', C.c_uint8), ('dst', C.c_uint8), ('a', C.c_uint8), ('b', C.c_uint8),
              ('c', C.c_uint8), ('pad0', C.c_uint8), ('pad1', C.c_uint8), ('pad2', C.c_uint8)]


class sp_program(C.Structure):
  _fields_ = [
      ('cls', C.c_int32), ('n_inputs', C.c_int32), ('n_instr', C.c_int32), ('result_reg', C.c_int32),
      ('ndim', C.c_int32), ('out_dtype', C.c_int32), ('linear', C.c_int32), ('pad', C.c_int32),
      ('shape', C.c_int64 * SP_MAX_DIMS),
      ('in_stride', (C.c_int64 * SP_MAX_DIMS) * SP_MAX_INPUTS),
      ('in_dtype', C.c_int32 * SP_MAX_INPUTS),
      ('consts', C.c_double * SP_MAX_CONSTS),
      ('iconsts', C.c_int64 * SP_MAX_CONSTS),
      ('instr', sp_instr * SP_MAX_INSTR),
  ]


class HipError(RuntimeError):
  """A libspartan_hip.so call failed (text from sp_last_error())."""


class HipLibraryMissing(ImportError):
  pass


_lib = None


def _declare(lib):
  vp, i32, i64, sz = C.c_void_p, C.c_int32, C.c_int64, C.c_size_t
  pp = C.POINTER(C.c_void_p)
  p64 = C.POINTER(C.c_int64)
  lib.sp_abi_version.restype = C.c_int
  lib.sp_last_error.restype = C.c_char_p
  lib.sp_device_count.argtypes = [C.POINTER(C.c_int)]
  lib.sp_device_info.argtypes = [C.c_int, C.POINTER(C.c_int), p64, C.c_char_p, sz]
  lib.sp_map_fused.argtypes = [C.POINTER(sp_program), pp, vp, vp]
  lib.sp_program_static_id.argtypes = [C.POINTER(sp_program), i32]
  lib.sp_jit_configure.argtypes = [C.c_int, C.c_longlong]
  lib.sp_jit_compiled_count.argtypes = []
  lib.sp_jit_wait.argtypes = []
  lib.sp_jit_wait.restype = None
  lib.sp_jit_compile_check.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(sp_program)]
  lib.sp_jit_seed_begin.argtypes = [C.c_char_p]
  lib.sp_jit_seed_end.argtypes = []
  lib.sp_reduce_workspace_bytes.argtypes = [i32, i64, i64, i64]
  lib.sp_reduce_workspace_bytes.restype = sz
  lib.sp_reduce.argtypes = [C.POINTER(sp_program), pp, i32, i64, i64, i64, vp, i32, vp, sz, vp]
  lib.sp_argreduce_workspace_bytes.argtypes = [i32, i64, i64, i64]
  lib.sp_argreduce_workspace_bytes.restype = sz
  lib.sp_argreduce.argtypes = [C.POINTER(sp_program), pp, i32, i64, i64, i64, i64, i64, vp, vp, vp, sz, vp]
  lib.sp_update.argtypes = [vp, i32, p64, i32, p64, p64, vp, i32, i32, i32, vp, vp]
  lib.sp_slice_copy.argtypes = [vp, p64, vp, p64, p64, i32, i32, vp]
  lib.sp_gemm_f32.argtypes = [vp, i64, vp, i64, vp, i64, i64, i64, i64, i32, vp]
  lib.sp_gemm_f64.argtypes = [vp, i64, vp, i64, vp, i64, i64, i64, i64, i32, vp]
  lib.sp_gemm_workspace_bytes.argtypes = [i32, i64, i64, i64]
  lib.sp_gemm_workspace_bytes.restype = sz
  lib.sp_gemm_ws.argtypes = [i32, vp, i64, vp, i64, vp, i64, i64, i64, i64, i32, vp, sz, vp]
  lib.sp_rowdot_colsum_workspace_bytes.argtypes = [i64, i64]
  lib.sp_rowdot_colsum_workspace_bytes.restype = sz
  lib.sp_rowdot_colsum_f32.argtypes = [vp, i64, i64, i64, vp, vp, i64, vp, i32, vp, sz, vp]
  lib.sp_nearest_center_workspace_bytes.argtypes = [i64, i64, i64]
  lib.sp_nearest_center_workspace_bytes.restype = sz
  lib.sp_nearest_center.argtypes = [vp, i32, i64, vp, i32, i64, i64, i64, i64, vp, i32, vp, sz, vp]
  lib.sp_kmeans_points_prepared_bytes.argtypes = [i64, i64]
  lib.sp_kmeans_points_prepared_bytes.restype = sz
  lib.sp_kmeans_points_prepare.argtypes = [vp, i64, i64, i64, vp, sz, vp]
  lib.sp_nearest_center_prepared_workspace_bytes.argtypes = [i64, i64, i64]
  lib.sp_nearest_center_prepared_workspace_bytes.restype = sz
  lib.sp_nearest_center_prepared.argtypes = [vp, i32, i64, vp, vp, i32, i64, i64, i64, i64, vp, i32, vp, sz, vp]
  lib.sp_bincount_i64.argtypes = [vp, i64, i64, vp, vp]
  lib.sp_segment_sum_workspace_bytes.argtypes = [i64, i64, i64]
  lib.sp_segment_sum_workspace_bytes.restype = sz
  lib.sp_segment_sum.argtypes = [vp, i32, i64, vp, i64, i64, i64, vp, vp, sz, vp]
  lib.sp_segment_sum_counts.argtypes = [vp, i32, i64, vp, i64, i64, i64, vp, vp, vp, sz, vp]
  lib.sp_random_fill.argtypes = [vp, i32, i64, i32, C.c_uint64, C.c_uint64, i64, i64, vp]
  lib.sp_cumscan.argtypes = [vp, vp, i32, i64, i64, i64, i32, vp]
  lib.sp_coo_to_csr_workspace_bytes.argtypes = [i64]
  lib.sp_coo_to_csr_workspace_bytes.restype = sz
  lib.sp_coo_to_csr.argtypes = [i32, i64, i64, i64, vp, vp, vp, vp, vp, vp, vp, sz, vp]
  lib.sp_csr_rows.argtypes = [i64, i64, vp, vp, vp]
  lib.sp_coo_box.argtypes = [i64, vp, vp, i64, i64, i64, i64, i64, i64, i32, vp]
  lib.sp_coo_reshape.argtypes = [i64, vp, vp, i64, i64, i64, i64, vp]
  lib.sp_csr_spmm_workspace_bytes.argtypes = [i64, i64]
  lib.sp_csr_spmm_workspace_bytes.restype = sz
  lib.sp_csr_spmv_plan_entries.argtypes = [i64]
  lib.sp_csr_spmv_plan_entries.restype = i64
  lib.sp_csr_spmv_plan.argtypes = [i64, i64, vp, vp, vp]
  lib.sp_csr_spmv_blockplan_bytes.argtypes = [i32, i64, i64, i64]
  lib.sp_csr_spmv_blockplan_bytes.restype = sz
  lib.sp_csr_spmv_blockplan.argtypes = [i32, i64, i64, i64, vp, vp, vp, vp, sz, vp]
  lib.sp_csr_spmv_blocked.argtypes = [i32, i64, i64, i64, vp, vp, vp, vp, i64, i32, vp]
  lib.sp_csr_spmm.argtypes = [i32, i64, i64, i64, i64, vp, vp, vp, vp, i64, vp, i64, i32, vp, vp, sz, vp]
  lib.sp_csr_scatter.argtypes = [i32, i64, i64, vp, vp, vp, vp, i64, i64, i64, vp, i64, i32, vp]
  lib.sp_spgemm_count_workspace_bytes.argtypes = [i64]
  lib.sp_spgemm_count_workspace_bytes.restype = sz
  lib.sp_spgemm_count.argtypes = [i64, vp, vp, vp, vp, vp, sz, vp]
  lib.sp_spgemm_expand.argtypes = [i32, i64, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
  lib.sp_tiling_solve.argtypes = [i32, i64, vp, vp, vp, i32, vp, vp, vp, vp]
  lib.sp_gather_rows.argtypes = [vp, i64, i64, vp, i64, i64, vp, vp]
  lib.sp_stream_copy.argtypes = [vp, vp, sz, vp]
  lib.sp_stream_copy_wg.argtypes = [vp, vp, sz, i32, vp]
  lib.sp_memset.argtypes = [vp, i32, sz, vp]
  lib.sp_device_synchronize.argtypes = []
  lib.sp_stream_create_priority.argtypes = [pp, i32]
  lib.sp_event_query.argtypes = [vp, C.POINTER(i32)]
  u64 = C.c_uint64
  lib.sp_blob_create.argtypes = [p64, i32, i32, C.POINTER(u64)]
  lib.sp_blob_destroy.argtypes = [u64]
  lib.sp_blob_trim.argtypes = []
  lib.sp_blob_info.argtypes = [u64, pp, p64, C.POINTER(i32), C.POINTER(i32)]
  lib.sp_blob_stats.argtypes = [p64, p64]
  lib.sp_blob_h2d.argtypes = [u64, vp, p64, p64, vp]
  lib.sp_blob_d2h.argtypes = [u64, vp, p64, p64, vp]
  lib.sp_blob_h2d_staged.argtypes = [u64, vp, p64, p64, vp, C.POINTER(i32)]
  lib.sp_blob_d2h_staged.argtypes = [u64, vp, p64, p64, vp]
  lib.sp_pinned_alloc.argtypes = [sz, pp]
  lib.sp_pinned_free.argtypes = [vp]
  lib.sp_copy_d2h_async.argtypes = [vp, vp, sz, vp]
  lib.sp_blob_slice_copy.argtypes = [u64, p64, u64, p64, p64, vp]
  lib.sp_comm_available.argtypes = []
  lib.sp_comm_version.argtypes = [C.POINTER(C.c_int)]
  lib.sp_comm_paths.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, sz]
  lib.sp_comm_unique_id.argtypes = [vp, sz]
  lib.sp_comm_init.argtypes = [i32, i32, vp, pp]
  lib.sp_comm_destroy.argtypes = [vp]
  lib.sp_comm_abort.argtypes = [vp]
  lib.sp_comm_async_error.argtypes = [vp]
  lib.sp_comm_all_reduce.argtypes = [vp, vp, vp, i64, i32, i32, vp]
  lib.sp_comm_reduce_scatter.argtypes = [vp, vp, vp, i64, i32, i32, vp]
  lib.sp_comm_reduce.argtypes = [vp, vp, vp, i64, i32, i32, i32, vp]
  lib.sp_comm_all_gather.argtypes = [vp, vp, vp, i64, i32, vp]
  lib.sp_comm_bcast.argtypes = [vp, vp, i64, i32, i32, vp]
  lib.sp_comm_all_to_all_blocks.argtypes = [vp, i32, C.POINTER(i32), pp, p64, i32, C.POINTER(i32), pp, p64, vp]
  lib.sp_set_device.argtypes = [i32]
  lib.sp_get_device.argtypes = [vp]
  lib.sp_jit_preload.argtypes = [C.c_int]
  lib.sp_jit_shutdown.argtypes = []
  lib.sp_jit_shutdown.restype = None
  lib.sp_stream_create.argtypes = [pp]
  lib.sp_stream_destroy.argtypes = [vp]
  lib.sp_stream_synchronize.argtypes = [vp]
  lib.sp_stream_query.argtypes = [vp, C.POINTER(i32)]
  lib.sp_stream_wait_event.argtypes = [vp, vp]
  lib.sp_event_create.argtypes = [pp]
  lib.sp_event_destroy.argtypes = [vp]
  lib.sp_event_record.argtypes = [vp, vp]
  lib.sp_event_synchronize.argtypes = [vp]
  lib.sp_event_elapsed_ms.argtypes = [vp, vp, C.POINTER(C.c_float)]
  return lib


# every symbol include/spartan_hip.h declares
EXPORTS = [
    'sp_abi_version', 'sp_last_error', 'sp_device_count', 'sp_device_info', 'sp_map_fused',
    'sp_program_static_id', 'sp_jit_configure', 'sp_jit_wait', 'sp_jit_compiled_count', 'sp_jit_compile_check', 'sp_jit_seed_begin', 'sp_jit_seed_end',
    'sp_reduce_workspace_bytes', 'sp_reduce', 'sp_argreduce_workspace_bytes', 'sp_argreduce',
    'sp_update', 'sp_slice_copy', 'sp_gemm_f32', 'sp_gemm_f64', 'sp_gemm_workspace_bytes', 'sp_gemm_ws', 'sp_rowdot_colsum_workspace_bytes', 'sp_rowdot_colsum_f32', 'sp_nearest_center_workspace_bytes', 'sp_nearest_center', 'sp_kmeans_points_prepared_bytes', 'sp_kmeans_points_prepare', 'sp_nearest_center_prepared_workspace_bytes', 'sp_nearest_center_prepared',
    'sp_bincount_i64', 'sp_segment_sum_workspace_bytes', 'sp_segment_sum', 'sp_segment_sum_counts', 'sp_random_fill', 'sp_cumscan',
    'sp_coo_to_csr_workspace_bytes', 'sp_coo_to_csr', 'sp_csr_rows', 'sp_coo_box', 'sp_coo_reshape', 'sp_csr_spmm_workspace_bytes', 'sp_csr_spmv_plan_entries', 'sp_csr_spmv_plan', 'sp_csr_spmv_blockplan_bytes', 'sp_csr_spmv_blockplan', 'sp_csr_spmv_blocked', 'sp_csr_spmm', 'sp_csr_scatter',
    'sp_spgemm_count_workspace_bytes', 'sp_spgemm_count', 'sp_spgemm_expand', 'sp_tiling_solve', 'sp_gather_rows', 'sp_stream_copy', 'sp_event_create',
    'sp_event_destroy', 'sp_event_record', 'sp_event_synchronize', 'sp_event_elapsed_ms',
    'sp_blob_create', 'sp_blob_destroy', 'sp_blob_trim', 'sp_blob_info', 'sp_blob_stats', 'sp_blob_h2d', 'sp_blob_h2d_staged', 'sp_blob_d2h', 'sp_blob_d2h_staged', 'sp_pinned_alloc', 'sp_pinned_free', 'sp_copy_d2h_async',
    'sp_blob_slice_copy', 'sp_comm_available', 'sp_comm_version', 'sp_comm_paths', 'sp_comm_unique_id', 'sp_comm_init',
    'sp_comm_destroy', 'sp_comm_abort', 'sp_comm_async_error', 'sp_comm_all_reduce', 'sp_comm_reduce_scatter',
    'sp_comm_reduce', 'sp_comm_all_gather', 'sp_comm_bcast', 'sp_comm_all_to_all_blocks', 'sp_set_device', 'sp_get_device', 'sp_jit_preload', 'sp_jit_shutdown',
    'sp_stream_create', 'sp_stream_create_priority', 'sp_device_synchronize', 'sp_memset', 'sp_stream_copy_wg',
    'sp_event_query', 'sp_stream_destroy', 'sp_stream_synchronize', 'sp_stream_query', 'sp_stream_wait_event',
]


def source_sha():
  """First 16 hex digits of the sha256 over the kernel sources (csrc/*.hip, *.hpp, the C-ABI header): profile
  summaries under profiles/ are stamped with it, and bench.py quotes a counter measurement only for the tree it
  was taken on."""
  import glob
  import hashlib
  h = hashlib.sha256()
  csrc = os.path.join(_HERE, 'csrc')
  files = sorted(glob.glob(os.path.join(csrc, '*.hip')) + glob.glob(os.path.join(csrc, '*.hpp')))
  files.append(os.path.join(os.path.dirname(_HERE), 'include', 'spartan_hip.h'))
  for f in files:
    h.update(os.path.basename(f).encode())
    with open(f, 'rb') as fh:
      h.update(fh.read())
  return h.hexdigest()[:16]


# every symbol include/spartan_hip_extras.h declares (libspartan_hip_extras.so: `make extras`)
EXTRAS_LIB_PATH = os.path.join(os.path.dirname(LIB_PATH), 'libspartan_hip_extras.so')
EXPORTS_EXTRAS = ['sp_sort_rows_workspace_bytes', 'sp_sort_rows']
_extras = None


def extras():
  """The library of kernels outside the tile path (sort); raises if it has not been built."""
  global _extras
  if _extras is None:
    lib()
    if not os.path.exists(EXTRAS_LIB_PATH):
      raise HipLibraryMissing('%s not found: build it with `make -C spartan_amd/csrc extras` (or '
                              '__graft_entry__.build())' % EXTRAS_LIB_PATH)
    x = C.CDLL(EXTRAS_LIB_PATH)
    vp, i32, i64, sz = C.c_void_p, C.c_int32, C.c_int64, C.c_size_t
    x.sp_sort_rows_workspace_bytes.argtypes = [i32, i64, i64]
    x.sp_sort_rows_workspace_bytes.restype = sz
    x.sp_sort_rows.argtypes = [vp, i32, i64, i64, vp, vp, vp, sz, vp]
    _extras = x
  return _extras


def lib():
  """Load (once) and return the C-ABI library; raises if it has not been built."""
  global _lib
  if _lib is None:
    if not os.path.exists(LIB_PATH):
      raise HipLibraryMissing(
          '%s not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
          '(or `make -C spartan_amd/csrc`). The HIP tile backend has no CPU fallback.' % LIB_PATH)
    # dmabuf IPC between the per-GPU processes of a job (RCCL over xGMI): the host driver supports nothing else, and
    # the HSA runtime reads the variable when it comes up -- i.e. with the first HIP call this library makes
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    _lib = _declare(C.CDLL(LIB_PATH))
    if _lib.sp_abi_version() != 1:
      raise HipError('libspartan_hip.so ABI version mismatch')
    # the run-time compile thread is stopped (after the compile it may be in) while the interpreter is still whole,
    # ahead of every C exit handler -- a short-lived process must not tear the compiler down under it
    import atexit
    atexit.register(_lib.sp_jit_shutdown)
  return _lib


def check(rc):
  if rc != 0:
    raise HipError(lib().sp_last_error().decode('utf-8', 'replace'))


def i64_array(vals):
  return (C.c_int64 * max(1, len(vals)))(*[int(v) for v in vals])


def ptr_array(ptrs):
  return (C.c_void_p * max(1, len(ptrs)))(*[C.c_void_p(int(p)) for p in ptrs])
