"""A host that brings neither torch nor a device allocator: ctypes + NumPy over libspartan_hip.so only.

Runs the tile path of INTEGRATION.md on library-owned blobs -- fused map, fp32 MFMA GEMM, the merge kernel, box
transfers -- and the sp_comm_* collectives on a one-rank communicator.  tests/test_runtime_gpu.py runs it in a
fresh interpreter on the GPU box and checks that `torch` was never imported.
"""
import ctypes as C
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# Only the ctypes binding and the program builder are used -- not the repository's host framework (which, since round
# 3, is itself such a ctypes + NumPy host: spartan_amd/devarray.py, kernels.py): this file stays the minimal binding
# INTEGRATION.md describes, so the package's __init__ is not run.
pkg = types.ModuleType('spartan_amd')
pkg.__path__ = [os.path.join(ROOT, 'spartan_amd')]
sys.modules['spartan_amd'] = pkg
from spartan_amd import _hip, program  # noqa: E402

lib = _hip.lib()
check = _hip.check


class Blob(object):
  """sp_blob_* handle with NumPy transfers."""

  def __init__(self, shape, dtype):
    self.shape, self.dtype = tuple(shape), np.dtype(dtype)
    h = C.c_uint64()
    check(lib.sp_blob_create(_hip.i64_array(self.shape), len(self.shape), _hip.sp_dtype(dtype), C.byref(h)))
    self.h = h
    p = C.c_void_p()
    check(lib.sp_blob_info(self.h, C.byref(p), None, None, None))
    self.ptr = p

  def put(self, a, stream, ul=None, lr=None):
    a = np.ascontiguousarray(a, dtype=self.dtype)
    check(lib.sp_blob_h2d(self.h, a.ctypes.data_as(C.c_void_p), None if ul is None else _hip.i64_array(ul),
                          None if lr is None else _hip.i64_array(lr), stream))
    check(lib.sp_stream_synchronize(stream))       # `a` is pageable host memory: finish before it goes away

  def get(self, stream, ul=None, lr=None):
    shape = self.shape if ul is None else tuple(b - a for a, b in zip(ul, lr))
    out = np.empty(shape, dtype=self.dtype)
    check(lib.sp_blob_d2h(self.h, out.ctypes.data_as(C.c_void_p), None if ul is None else _hip.i64_array(ul),
                          None if lr is None else _hip.i64_array(lr), stream))
    check(lib.sp_stream_synchronize(stream))
    return out

  def free(self):
    check(lib.sp_blob_destroy(self.h))


def main():
  check(lib.sp_set_device(0))
  stream = C.c_void_p()
  check(lib.sp_stream_create(C.byref(stream)))
  rng = np.random.RandomState(20150708)

  # ---- fused map x*x + x: register 0 holds the input, r1 = r0*r0, r1 = r1 + r0
  x = rng.randint(-9, 10, size=(1024, 512)).astype(np.float32)
  X, Y = Blob(x.shape, np.float32), Blob(x.shape, np.float32)
  X.put(x, stream)
  pb = program.Program()
  pb.add_input(np.float32, (1,))               # one collapsed, dense dimension
  pb.emit('MUL', 1, 0, 0)
  pb.emit('ADD', 1, 1, 0)
  pb.result_reg = 1
  prog = pb.finish(_hip.SP_F32, (x.size,), np.float32, linear=True)
  check(lib.sp_map_fused(C.byref(prog), _hip.ptr_array([X.ptr.value]), Y.ptr, stream))
  np.testing.assert_array_equal(Y.get(stream), x * x + x)

  # ---- spartan.dot's tile body, then the np.add merge of a second partial (accumulate = 1)
  a = rng.randint(-3, 4, size=(256, 384)).astype(np.float32)
  b = rng.randint(-3, 4, size=(384, 128)).astype(np.float32)
  A, B, Cm = Blob(a.shape, np.float32), Blob(b.shape, np.float32), Blob((256, 128), np.float32)
  A.put(a, stream)
  B.put(b, stream)
  check(lib.sp_gemm_f32(A.ptr, 384, B.ptr, 128, Cm.ptr, 128, 256, 128, 384, 0, stream))
  check(lib.sp_gemm_f32(A.ptr, 384, B.ptr, 128, Cm.ptr, 128, 256, 128, 384, 1, stream))
  np.testing.assert_array_equal(Cm.get(stream), 2 * a.dot(b))

  # ---- boxes: host -> box of a blob, box -> host, blob box -> blob box
  T = Blob((64, 96), np.int64)
  T.put(np.zeros((64, 96), np.int64), stream)
  patch = np.arange(20 * 30, dtype=np.int64).reshape(20, 30)
  T.put(patch, stream, ul=(5, 7), lr=(25, 37))
  want = np.zeros((64, 96), np.int64)
  want[5:25, 7:37] = patch
  np.testing.assert_array_equal(T.get(stream), want)
  np.testing.assert_array_equal(T.get(stream, ul=(10, 0), lr=(30, 50)), want[10:30, 0:50])
  U = Blob((40, 40), np.int64)
  U.put(np.full((40, 40), -1, np.int64), stream)
  check(lib.sp_blob_slice_copy(U.h, _hip.i64_array((3, 4)), T.h, _hip.i64_array((5, 7)), _hip.i64_array((20, 30)), stream))
  wu = np.full((40, 40), -1, np.int64)
  wu[3:23, 4:34] = patch
  np.testing.assert_array_equal(U.get(stream), wu)

  # ---- freed blobs are kept for re-use
  live, pooled = C.c_int64(), C.c_int64()
  p_old = U.ptr.value
  U.free()
  check(lib.sp_blob_stats(C.byref(live), C.byref(pooled)))
  assert pooled.value >= 40 * 40 * 8
  U2 = Blob((40, 40), np.int64)
  assert U2.ptr.value == p_old
  check(lib.sp_blob_trim())

  # ---- collectives over RCCL on a one-rank communicator (the N-rank schedules degenerate to copies)
  if lib.sp_comm_available():
    uid = C.create_string_buffer(_hip.SP_COMM_UID_BYTES)
    check(lib.sp_comm_unique_id(uid, _hip.SP_COMM_UID_BYTES))
    comm = C.c_void_p()
    check(lib.sp_comm_init(1, 0, uid, C.byref(comm)))
    v = rng.randint(-5, 6, size=4096).astype(np.float32)
    V, W = Blob(v.shape, np.float32), Blob(v.shape, np.float32)
    V.put(v, stream)
    check(lib.sp_comm_all_reduce(comm, V.ptr, W.ptr, v.size, _hip.SP_F32, _hip.REDUCER['ADD'], stream))
    np.testing.assert_array_equal(W.get(stream), v)
    check(lib.sp_comm_reduce_scatter(comm, V.ptr, W.ptr, v.size, _hip.SP_F32, _hip.REDUCER['MAX'], stream))
    np.testing.assert_array_equal(W.get(stream), v)
    check(lib.sp_comm_all_gather(comm, V.ptr, W.ptr, v.size, _hip.SP_F32, stream))
    check(lib.sp_comm_reduce(comm, V.ptr, W.ptr, v.size, _hip.SP_F32, _hip.REDUCER['ADD'], 0, stream))
    check(lib.sp_comm_bcast(comm, W.ptr, v.size, _hip.SP_F32, 0, stream))
    np.testing.assert_array_equal(W.get(stream), v)
    Z = Blob(v.shape, np.float32)
    peers = (C.c_int32 * 1)(0)
    nbytes = _hip.i64_array([v.nbytes])
    check(lib.sp_comm_all_to_all_blocks(comm, 1, peers, _hip.ptr_array([V.ptr.value]), nbytes,
                                        1, peers, _hip.ptr_array([Z.ptr.value]), nbytes, stream))
    np.testing.assert_array_equal(Z.get(stream), v)
    check(lib.sp_comm_async_error(comm))
    check(lib.sp_comm_destroy(comm))
    ver = C.c_int()
    check(lib.sp_comm_version(C.byref(ver)))
    print('collectives OK (RCCL %d)' % ver.value)
  else:
    print('collectives skipped: %s' % lib.sp_last_error().decode())
  check(lib.sp_stream_destroy(stream))
  assert 'torch' not in sys.modules, 'this host must not need torch'
  print('torch-free host OK')


if __name__ == '__main__':
  main()
