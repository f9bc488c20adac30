"""C-ABI level parity: every libspartan_hip.so kernel against NumPy on the same
seeded inputs (bit-exact for integer / index / comparison work, stated
tolerances for fp32 sums and GEMM)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from spartan_amd import _hip, kernels  # noqa: E402
from spartan_amd import devarray as D  # noqa: E402
from spartan_amd.program import Program, broadcast_strides, collapse, dense_strides  # noqa: E402

RNG = np.random.RandomState(20150708)


def dev(a):
  return D.from_numpy(a)


def host(t):
  return t.numpy()


def build(cls, shape, inputs, body, out_dtype):
  """inputs: list of (np dtype, in_shape); body(p) emits instrs and returns the result reg."""
  p = Program()
  strides = [broadcast_strides(s, shape) for _, s in inputs]
  cshape, cstrides = collapse(shape, strides)
  linear = all(st == dense_strides(cshape) or all(x == 0 for x in st) for st in cstrides)
  for (dt, _), st in zip(inputs, cstrides):
    p.add_input(dt, st)
  p.result_reg = body(p)
  return p.finish(cls, cshape, out_dtype, linear)


def run_map(cls, shape, arrays, body, out_dtype):
  prog = build(cls, shape, [(a.dtype, a.shape) for a in arrays], body, out_dtype)
  out = D.empty(shape, out_dtype)
  kernels.map_fused(prog, [dev(a) for a in arrays], out)
  D.synchronize()
  return host(out)


@pytest.mark.parametrize('shape', [(1000, 1000), (1003,), (7, 13), (4, 8, 16), (1,), (2, 3, 5, 7)])
def test_map_add_scalar_f32(shape):
  x = RNG.rand(*shape).astype(np.float32)

  def body(p):
    c = p.add_const(1.0)
    p.emit('CONST', 1, c)
    p.emit('ADD', 2, 0, 1)
    return 2
  got = run_map(_hip.SP_F32, shape, [x], body, np.float32)
  np.testing.assert_array_equal(got, x + np.float32(1))


def test_map_xx_plus_x_large():
  x = RNG.rand(2048, 4096).astype(np.float32)

  def body(p):
    p.emit('MUL', 1, 0, 0)
    p.emit('ADD', 1, 1, 0)
    return 1
  got = run_map(_hip.SP_F32, x.shape, [x], body, np.float32)
  np.testing.assert_array_equal(got, x * x + x)


@pytest.mark.parametrize('bshape', [(64, 1), (1, 48), (48,), (1, 1)])
def test_map_broadcast(bshape):
  x = RNG.rand(64, 48).astype(np.float32)
  y = RNG.rand(*bshape).astype(np.float32)

  def body(p):
    p.emit('SUB', 2, 0, 1)
    p.emit('MUL', 2, 2, 0)
    return 2
  got = run_map(_hip.SP_F32, x.shape, [x, y], body, np.float32)
  np.testing.assert_array_equal(got, (x - y) * x)


def test_map_broadcast_odd_inner():
  x = RNG.rand(33, 7).astype(np.float32)
  y = RNG.rand(33, 1).astype(np.float32)
  z = RNG.rand(7).astype(np.float32)

  def body(p):
    p.emit('ADD', 3, 0, 1)
    p.emit('MUL', 3, 3, 2)
    return 3
  got = run_map(_hip.SP_F32, x.shape, [x, y, z], body, np.float32)
  np.testing.assert_array_equal(got, (x + y) * z)


def test_map_int64_and_compare():
  a = RNG.randint(-50, 50, size=(129, 17)).astype(np.int64)
  b = RNG.randint(1, 9, size=(129, 17)).astype(np.int32)

  def body_mod(p):
    p.emit('MOD', 2, 0, 1)
    return 2
  np.testing.assert_array_equal(run_map(_hip.SP_I64, a.shape, [a, b], body_mod, np.int64), np.mod(a, b))

  def body_fd(p):
    p.emit('FLOORDIV', 2, 0, 1)
    return 2
  np.testing.assert_array_equal(run_map(_hip.SP_I64, a.shape, [a, b], body_fd, np.int64), a // b)

  def body_lt(p):
    p.emit('LT', 2, 0, 1)
    return 2
  np.testing.assert_array_equal(run_map(_hip.SP_I64, a.shape, [a, b], body_lt, np.bool_), a < b)


def test_map_f64_mixed_and_transcendental():
  x = (RNG.rand(300, 5) + 0.5).astype(np.float32)
  k = RNG.randint(0, 5, size=(300, 5)).astype(np.int64)

  def body(p):
    p.emit('ADD', 2, 0, 1)
    p.emit('SQRT', 2, 2)
    return 2
  got = run_map(_hip.SP_F64, x.shape, [x, k], body, np.float64)
  np.testing.assert_allclose(got, np.sqrt(x + k), rtol=1e-15)

  def body2(p):
    p.emit('LOG', 1, 0)
    p.emit('EXP', 1, 1)
    return 1
  got = run_map(_hip.SP_F32, x.shape, [x], body2, np.float32)
  np.testing.assert_allclose(got, np.exp(np.log(x)), rtol=3e-7)


def test_map_iota_arange():
  shape = (37, 11)

  def body(p):
    p.emit('IOTA', 0)
    c = p.add_const(2.0)
    p.emit('CONST', 1, c)
    p.emit('MUL', 0, 0, 1)
    return 0
  got = run_map(_hip.SP_F64, shape, [], body, np.float64)
  np.testing.assert_array_equal(got, (np.arange(37 * 11) * 2.0).reshape(shape))


# ---------------------------------------------------------------- reductions
def run_reduce(x, axis, op, cls, out_dtype, body=None, extra=()):
  shape = x.shape
  arrays = [x] + list(extra)
  if axis is None:
    O, A, I = 1, x.size, 1
  else:
    O = int(np.prod(shape[:axis], dtype=np.int64))
    A = shape[axis]
    I = int(np.prod(shape[axis + 1:], dtype=np.int64))
  p = Program()
  # the program's index space only has to enumerate the same elements in the
  # same row-major order as [O, A, I]; it keeps its own (collapsed) shape
  cshape, cstrides = collapse(shape, [broadcast_strides(a.shape, shape) for a in arrays])
  linear = all(tuple(st) == dense_strides(cshape) for st in cstrides)
  for a, st in zip(arrays, cstrides):
    p.add_input(a.dtype, st)
  p.result_reg = body(p) if body else 0
  prog = p.finish(cls, cshape, None, linear)
  out = D.empty((max(O * I, 1),), out_dtype)
  kernels.reduce(prog, [dev(a) for a in arrays], op, O, A, I, out)
  D.synchronize()
  res = host(out)
  if axis is None:
    return res[0]
  return res.reshape(shape[:axis] + shape[axis + 1:])


SHAPES = [(64, 64), (1000, 333), (5, 70000), (70000, 5), (17,), (3, 5, 7), (2048, 2048), (1, 1)]


@pytest.mark.parametrize('shape', SHAPES)
def test_reduce_sum_all_axes(shape):
  x = RNG.rand(*shape).astype(np.float32)
  for axis in [None] + list(range(len(shape))):
    got = run_reduce(x, axis, 'SUM', _hip.SP_F32, np.float32)
    ref = x.astype(np.float64).sum(axis)
    tol = 1e-6 * np.abs(x).astype(np.float64).sum(axis)
    assert np.all(np.abs(got - ref) <= tol + 1e-30), (shape, axis, np.abs(got - ref).max())


@pytest.mark.parametrize('shape', SHAPES)
def test_reduce_max_min_exact(shape):
  x = RNG.randn(*shape).astype(np.float32)
  for axis in [None] + list(range(len(shape))):
    np.testing.assert_array_equal(run_reduce(x, axis, 'MAX', _hip.SP_F32, np.float32), x.max(axis))
    np.testing.assert_array_equal(run_reduce(x, axis, 'MIN', _hip.SP_F32, np.float32), x.min(axis))


def test_reduce_integer_exact():
  x = RNG.randint(-1000, 1000, size=(513, 129)).astype(np.int32)
  for axis in (None, 0, 1):
    np.testing.assert_array_equal(run_reduce(x, axis, 'SUM', _hip.SP_I64, np.int64), x.sum(axis, dtype=np.int64))
  b = RNG.rand(100, 40) > 0.01
  for axis in (None, 0, 1):
    np.testing.assert_array_equal(run_reduce(b, axis, 'AND', _hip.SP_I64, np.bool_), np.all(b, axis))
    np.testing.assert_array_equal(run_reduce(b, axis, 'OR', _hip.SP_I64, np.bool_), np.any(b, axis))
  small = RNG.randint(1, 3, size=(3, 12)).astype(np.int64)
  np.testing.assert_array_equal(run_reduce(small, 1, 'PROD', _hip.SP_I64, np.int64), small.prod(1))


def test_reduce_fused_map_prologue():
  # sum(x * (yp - y), axis=0): the lreg gradient (sgd.py:34-39) as ONE launch
  x = RNG.rand(3000, 64).astype(np.float32)
  yp = RNG.rand(3000, 1).astype(np.float32)
  y = RNG.rand(3000, 1).astype(np.float32)

  def body(p):
    p.emit('SUB', 3, 1, 2)
    p.emit('MUL', 3, 0, 3)
    return 3
  for axis in (0, 1, None):
    got = run_reduce(x, axis, 'SUM', _hip.SP_F32, np.float32, body, extra=[yp, y])
    ref = (x.astype(np.float64) * (yp.astype(np.float64) - y)).sum(axis)
    np.testing.assert_allclose(got, ref, rtol=2e-5, atol=1e-3)


def run_arg(x, axis, which, offset=0, sentinel=-7):
  shape = x.shape
  if axis is None:
    O, A, I = 1, x.size, 1
  else:
    O = int(np.prod(shape[:axis], dtype=np.int64))
    A = shape[axis]
    I = int(np.prod(shape[axis + 1:], dtype=np.int64))
  p = Program()
  p.add_input(x.dtype, dense_strides((O, A, I)))
  prog = p.finish(_hip.SP_F32 if x.dtype == np.float32 else _hip.SP_I64, (O, A, I), None, True)
  oi = D.empty((max(O * I, 1),), np.int64)
  ov = D.empty((max(O * I, 1),), np.float32 if x.dtype == np.float32 else np.int64)
  kernels.argreduce(prog, [dev(x)], which, O, A, I, offset, sentinel, oi, ov)
  D.synchronize()
  idx, val = host(oi), host(ov)
  if axis is None:
    return idx[0], val[0]
  rs = shape[:axis] + shape[axis + 1:]
  return idx.reshape(rs), val.reshape(rs)


@pytest.mark.parametrize('shape', [(64, 64), (1000, 333), (5, 70000), (70000, 5), (17,), (3, 5, 7), (1024, 4096)])
def test_argreduce_first_occurrence(shape):
  # few distinct values => many ties: first occurrence must win (sorting.py:67-123)
  x = RNG.randint(0, 4, size=shape).astype(np.float32)
  for axis in [None] + list(range(len(shape))):
    idx, val = run_arg(x, axis, 0, offset=100)
    np.testing.assert_array_equal(idx, np.argmax(x, axis) + 100)
    np.testing.assert_array_equal(val, x.max(axis))
    idx, val = run_arg(x, axis, 1)
    np.testing.assert_array_equal(idx, np.argmin(x, axis))
    np.testing.assert_array_equal(val, x.min(axis))


def test_argreduce_nan_sentinel():
  x = RNG.rand(8, 16).astype(np.float32)
  x[3, 5] = np.nan
  idx, val = run_arg(x, 1, 0, sentinel=128)
  ref = np.argmax(np.where(np.isnan(x), -np.inf, x), 1)
  ref[3] = 128
  np.testing.assert_array_equal(idx, ref)
  assert np.isnan(val[3])


# --------------------------------------------------------------------- merge
def test_update_truth_table():
  # tile.pyx:200-297, dense->dense branch (truth table captured in SURVEY 8c)
  tile_shape = (6, 8)
  t = D.zeros(tile_shape, np.float32)
  mask = D.zeros(tile_shape, np.uint8)
  u1 = RNG.rand(*tile_shape).astype(np.float32)
  # full-tile first write on an empty tile: replace
  kernels.update(t, (0, 0), tile_shape, dev(u1), 'ADD', _hip.MASK_ALL_CLEAR, None)
  np.testing.assert_array_equal(host(t), u1)
  # second full-tile write: accumulate
  u2 = RNG.rand(*tile_shape).astype(np.float32)
  kernels.update(t, (0, 0), tile_shape, dev(u2), 'ADD', _hip.MASK_ALL_SET, None)
  np.testing.assert_array_equal(host(t), u1 + u2)
  # reducer None: replace
  kernels.update(t, (0, 0), tile_shape, dev(u2), 'NONE', _hip.MASK_ALL_SET, None)
  np.testing.assert_array_equal(host(t), u2)
  # sub-slice writes into an empty (zero-initialised) tile with an explicit mask
  t.fill(0)
  s1 = RNG.rand(3, 4).astype(np.float32)
  kernels.update(t, (1, 2), (4, 6), dev(s1), 'ADD', _hip.MASK_ARRAY, mask)
  exp = np.zeros(tile_shape, np.float32)
  exp[1:4, 2:6] = s1
  expm = np.zeros(tile_shape, np.uint8)
  expm[1:4, 2:6] = 1
  np.testing.assert_array_equal(host(t), exp)
  np.testing.assert_array_equal(host(mask), expm)
  # overlapping second sub-slice: written cells reduced, new cells copied
  s2 = RNG.rand(4, 4).astype(np.float32)
  kernels.update(t, (2, 4), (6, 8), dev(s2), 'ADD', _hip.MASK_ARRAY, mask)
  exp2 = exp.copy()
  region = exp2[2:6, 4:8]
  m = expm[2:6, 4:8].astype(bool)
  region[m] = region[m] + s2[m]
  region[~m] = s2[~m]
  expm[2:6, 4:8] = 1
  np.testing.assert_array_equal(host(t), exp2)
  np.testing.assert_array_equal(host(mask), expm)


def test_update_reducers_and_dtypes():
  a = RNG.randint(-5, 5, size=(33, 17)).astype(np.int64)
  b = RNG.randint(-5, 5, size=(33, 17)).astype(np.int64)
  for name, fn in (('MAX', np.maximum), ('MIN', np.minimum), ('MUL', np.multiply), ('ADD', np.add)):
    t = dev(a.copy())
    kernels.update(t, (0, 0), a.shape, dev(b), name, _hip.MASK_ALL_SET, None)
    np.testing.assert_array_equal(host(t), fn(a, b))
  ba, bb = a > 0, b > 0
  t = dev(ba.copy())
  kernels.update(t, (0, 0), a.shape, dev(bb), 'AND', _hip.MASK_ALL_SET, None)
  np.testing.assert_array_equal(host(t), np.logical_and(ba, bb))
  # update.astype(old.dtype): float64 update into a float32 tile
  t = D.zeros((33, 17), np.float32)
  upd = RNG.rand(33, 17)
  kernels.update(t, (0, 0), (33, 17), dev(upd), 'ADD', _hip.MASK_ALL_CLEAR, None)
  np.testing.assert_array_equal(host(t), upd.astype(np.float32))
  # 1-d and 3-d boxes
  t = D.zeros((4, 6, 8), np.float64)
  upd = RNG.rand(2, 3, 4)
  kernels.update(t, (1, 2, 4), (3, 5, 8), dev(upd), 'NONE', _hip.MASK_ALL_CLEAR, None)
  exp = np.zeros((4, 6, 8))
  exp[1:3, 2:5, 4:8] = upd
  np.testing.assert_array_equal(host(t), exp)


def test_slice_copy():
  src = RNG.rand(40, 48).astype(np.float32)
  dst = D.zeros((20, 16), np.float32)
  kernels.slice_copy(dst, 0, (16, 1), dev(src), 5 * 48 + 8, (48, 1), (20, 16))
  np.testing.assert_array_equal(host(dst), src[5:25, 8:24])
  src3 = RNG.randint(0, 100, size=(5, 7, 9)).astype(np.int64)
  dst3 = D.zeros((5, 7, 9), np.int64)
  kernels.slice_copy(dst3, 1 * 63 + 2 * 9 + 3, (63, 9, 1), dev(src3), 0, (63, 9, 1), (3, 4, 5))
  exp = np.zeros((5, 7, 9), np.int64)
  exp[1:4, 2:6, 3:8] = src3[0:3, 0:4, 0:5]
  np.testing.assert_array_equal(host(dst3), exp)
  b = RNG.rand(13, 7) > 0.5
  db = D.zeros((13, 7), np.bool_)
  kernels.slice_copy(db, 0, (7, 1), dev(b), 0, (7, 1), (13, 7))
  np.testing.assert_array_equal(host(db), b)


# ---------------------------------------------------------------------- GEMM
@pytest.mark.parametrize('mnk', [(256, 128, 16), (512, 512, 512), (100, 60, 30), (1000, 24, 37),
                                 (257, 129, 17), (64, 2000, 2000), (1250, 1024, 256), (1, 1, 1),
                                 # aligned rows, K not a multiple of 16: the direct-to-LDS kernels' register tail,
                                 # 128 x 128 and 256 x 128 macro-tiles
                                 (300, 256, 100), (2048, 2048, 40), (4096, 4096, 20), (513, 260, 1000)])
def test_gemm_integer_valued_exact(mnk):
  # small-integer operands: every partial sum is exact in fp32, so the result
  # must be bit-identical to NumPy (the reference tests use arange/ones inputs,
  # tests/test_dot.py:8-103)
  M, N, K = mnk
  a = RNG.randint(-3, 4, size=(M, K)).astype(np.float32)
  b = RNG.randint(-3, 4, size=(K, N)).astype(np.float32)
  c = D.full((M, N), 7.0, np.float32)
  kernels.gemm_f32(dev(a), dev(b), c, accumulate=False)
  np.testing.assert_array_equal(host(c), a.dot(b))
  kernels.gemm_f32(dev(a), dev(b), c, accumulate=True)
  np.testing.assert_array_equal(host(c), 2 * a.dot(b))


def test_gemm_random_tolerance_and_transpose_detect():
  M, N, K = 768, 640, 1024
  a = (RNG.rand(M, K) * 2 - 1).astype(np.float32)
  b = (RNG.rand(K, N) * 2 - 1).astype(np.float32)
  c = D.empty((M, N), np.float32)
  kernels.gemm_f32(dev(a), dev(b), c)
  ref = a.astype(np.float64).dot(b.astype(np.float64))
  # |dC| <= 2 K eps max|a| max|b|
  assert np.abs(host(c) - ref).max() <= 2 * K * np.finfo(np.float32).eps
  # A = I with an asymmetric B catches a row/col swap in the C write
  eye = np.eye(256, dtype=np.float32)
  bb = np.arange(256 * 128, dtype=np.float32).reshape(256, 128)
  c2 = D.empty((256, 128), np.float32)
  kernels.gemm_f32(dev(eye), dev(bb), c2)
  np.testing.assert_array_equal(host(c2), bb)


def test_gemm_strided_views():
  # K-slab of A and row-slab of B taken as views (the map2 join of dot.py:195-217)
  a = RNG.randint(-2, 3, size=(128, 512)).astype(np.float32)
  b = RNG.randint(-2, 3, size=(512, 256)).astype(np.float32)
  da, db = dev(a), dev(b)
  c = D.zeros((128, 256), np.float32)
  for k0 in range(0, 512, 128):
    kernels.gemm_f32(da[:, k0:k0 + 128], db[k0:k0 + 128, :], c, accumulate=True)
  np.testing.assert_array_equal(host(c), a.dot(b))


def test_errors_are_loud():
  with pytest.raises(_hip.HipError):
    kernels.gemm_f32(np.zeros((4, 4), np.float32), np.zeros((4, 4), np.float32), np.zeros((4, 4), np.float32))   # host memory


def test_static_program_library():
  """The streams the host emits for the hot expressions are the ones the kernel
  library was specialised for (so they do not silently fall back to the
  interpreter), and both paths agree bit for bit."""
  import ctypes as C
  import spartan_amd as sp
  from spartan_amd import lower
  from spartan_amd.backend_hip import HipBackend
  be = HipBackend()
  x = dev(RNG.rand(256, 512).astype(np.float32))
  y = dev(RNG.rand(256, 512).astype(np.float32))
  yp = dev(RNG.rand(256, 1).astype(np.float32))
  yy = dev(RNG.rand(256, 1).astype(np.float32))
  T = lambda t: lower.V('tensor', dtype=np.float32, shape=tuple(t.shape), tensor=t)
  ap = lower.apply
  cases = {
      1: ap('ADD', np.add, [T(x), lower.const(1)]),
      2: ap('SUB', np.subtract, [T(x), lower.const(1.5)]),
      3: ap('MUL', np.multiply, [T(x), lower.const(2.0)]),
      4: ap('DIV', np.divide, [T(x), lower.const(7)]),
      5: ap('ADD', np.add, [T(x), T(y)]),
      6: ap('SUB', np.subtract, [T(x), T(y)]),
      7: ap('MUL', np.multiply, [T(x), T(y)]),
      8: ap('DIV', np.divide, [T(x), T(y)]),
      9: ap('ADD', np.add, [ap('MUL', np.multiply, [T(x), T(x)]), T(x)]),
      10: ap('MUL', np.multiply, [T(x), ap('SUB', np.subtract, [T(yp), T(yy)])]),
      11: ap('MUL', np.multiply, [T(x), T(x)]),
      12: ap('SUB', np.subtract, [lower.const(3.0), T(x)]),
  }
  also = [(3, ap('MUL', np.multiply, [lower.const(3.0), T(x)])), (1, ap('ADD', np.add, [lower.const(0.25), T(x)]))]
  for sid, root in also:
    prog, _ = lower.Emitter(_hip.SP_F32, root.shape).finish(root, np.float32)
    assert _hip.lib().sp_program_static_id(C.byref(prog), _hip.SP_F32) == sid, sid
  for sid, root in cases.items():
    em = lower.Emitter(_hip.SP_F32, root.shape)
    prog, tensors = em.finish(root, np.float32)
    assert _hip.lib().sp_program_static_id(C.byref(prog), _hip.SP_F32) == sid, sid
  ident = lower.Emitter(_hip.SP_F32, x.shape).finish(T(x), None)[0]
  assert _hip.lib().sp_program_static_id(C.byref(ident), -1) == 0
  # a shape outside the library runs on the interpreter and still agrees with NumPy
  root = ap('ADD', np.add, [ap('MUL', np.multiply, [T(x), T(y)]), ap('SUB', np.subtract, [T(x), lower.const(2)])])
  prog, _ = lower.Emitter(_hip.SP_F32, root.shape).finish(root, np.float32)
  assert _hip.lib().sp_program_static_id(C.byref(prog), _hip.SP_F32) == -1
  np.testing.assert_array_equal(host(be._run_map(root, x.shape)), host(x) * host(y) + (host(x) - 2))


@pytest.mark.parametrize('tier', ['specialised', 'interpreted'])
@pytest.mark.parametrize('dt', [np.float32, np.float64, np.int64, np.int32])
def test_operators_with_a_constant_operand(dt, tier):
  """`x op c` / `c op x` lower to ONE instruction (SP_OP_ADDC .. SP_OP_MINC: reg[b] op consts[a]); every form, every
  arithmetic class, interpreter and specialised code, against NumPy -- and a chain of them, so the constant table
  holds several entries and registers are reused."""
  import ctypes as C
  from spartan_amd import lower
  from spartan_amd.backend_hip import HipBackend
  be = HipBackend()
  rng = np.random.RandomState(11)
  a = (rng.rand(129, 515) * 20 - 10).astype(dt)
  if np.dtype(dt).kind == 'i':
    a[a == 0] = 3
  x = dev(a)
  T = lambda: lower.V('tensor', dtype=np.dtype(dt), shape=a.shape, tensor=x)
  ap = lower.apply
  c = 3 if np.dtype(dt).kind == 'i' else 2.5
  forms = [('ADD', np.add), ('SUB', np.subtract), ('MUL', np.multiply), ('MAX', np.maximum), ('MIN', np.minimum)]
  if np.dtype(dt).kind == 'f':
    forms.append(('DIV', np.divide))
  cls = lower.class_of(np.dtype(dt))
  seen = set()
  with (_interpreted() if tier == 'interpreted' else _nothing()):
    for name, fn in forms:
      for const_first in (False, True):
        args = [lower.const(c), T()] if const_first else [T(), lower.const(c)]
        root = ap(name, fn, args)
        prog, _ = lower.Emitter(cls, root.shape).finish(root, np.dtype(dt))
        ops = [prog.instr[i].op for i in range(prog.n_instr)]
        assert _hip.OP['CONST'] not in ops and 60 <= ops[0] <= 67, ops
        seen.add(ops[0])
        want = fn(np.asarray(c, dt), a) if const_first else fn(a, np.asarray(c, dt))
        np.testing.assert_array_equal(host(be._run_map(root, a.shape)), want.astype(dt))
    # ((x * c1 + c2) max c3) - then c4 - that: four constants, one input
    root = ap('SUB', np.subtract, [lower.const(c + 4), ap('MAX', np.maximum, [
        ap('ADD', np.add, [ap('MUL', np.multiply, [T(), lower.const(c)]), lower.const(c + 1)]), lower.const(c + 2)])])
    prog, _ = lower.Emitter(cls, root.shape).finish(root, np.dtype(dt))
    assert _hip.OP['CONST'] not in [prog.instr[i].op for i in range(prog.n_instr)]
    k = lambda v: np.asarray(v, dt)
    want = k(c + 4) - np.maximum(a * k(c) + k(c + 1), k(c + 2))
    np.testing.assert_array_equal(host(be._run_map(root, a.shape)), want.astype(dt))
  assert len(seen) == (8 if np.dtype(dt).kind == 'f' else 6)


def test_constant_operand_instructions_are_validated():
  """A const index past the table and an unwritten source register are refused by the library."""
  x = RNG.rand(64).astype(np.float32)
  for a, b, why in ((16, 0, 'const index'), (0, 3, 'nothing has written')):
    def body(p, a=a, b=b):
      p.add_const(1.0)
      p.emit('ADDC', 1, a, b)
      return 1
    with pytest.raises(Exception, match=why):
      run_map(_hip.SP_F32, (64,), [x], body, np.float32)


class _nothing(object):
  def __enter__(self):
    pass

  def __exit__(self, *exc):
    pass


@pytest.mark.parametrize('shape', [(64, 64), (100, 37), (1000, 513), (17, 4096)])
def test_transposing_slice_copy(shape):
  """dst = src.T through sp_slice_copy (the LDS-tiled path) for 4- and 8-byte elements."""
  for dt in (np.float32, np.int64):
    a = RNG.randint(-1000, 1000, size=shape).astype(dt)
    da = dev(a)
    v = da.t()
    out = D.empty((shape[1], shape[0]), da.dtype)
    kernels.slice_copy(out, 0, out.stride(), da, 0, v.stride(), v.shape)
    np.testing.assert_array_equal(host(out), a.T)


# ---- interpreter tier: dispatch from the lane-held program, software-pipelined trips ----------------------------
class _interpreted(object):
  """Run-time specialisation off (none of the programs below is in the prebuilt library)."""

  def __enter__(self):
    _hip.lib().sp_jit_configure(0, -1)

  def __exit__(self, *exc):
    _hip.lib().sp_jit_configure(1, 1 << 22)


def _long_body(n_inputs, rounds):
  """((x0 * c + x1) - x2 ...) `rounds` times over the operands: 1 + 2 * rounds instructions (<= 64)."""
  def body(p):
    c = p.add_const(0.75)
    p.emit('CONST', 6, c)
    p.emit('MUL', 7, 0, 6)
    for k in range(rounds - 1):
      p.emit('ADD' if k % 3 else 'SUB', 7, 7, (k + 1) % n_inputs)
      p.emit('MUL', 7, 7, 6)
    return 7
  return body


def _long_numpy(xs, rounds):
  c = np.float32(0.75)
  acc = xs[0] * c
  for k in range(rounds - 1):
    y = xs[(k + 1) % len(xs)]
    acc = (acc + y) if k % 3 else (acc - y)
    acc = acc * c
  return acc


@pytest.mark.parametrize('n', [1, 5, 1023, 262144, 262144 + 4, 262144 * 2 + 1024 + 3, 3 * 1000 * 1000 + 1,
                               8192 * 256 * 2 * 4 * 2 + 4096 + 8])
@pytest.mark.parametrize('n_inputs', [1, 2, 3])
def test_interpreter_dense_sizes_and_operand_counts(n, n_inputs):
  """Every trip structure of the interpreted dense kernel: one lane, a partial first trip, exactly one trip per lane,
  the second group of a lane cut by the end, several trips with the last one partial, more vectors than the capped
  grid covers in one trip; one and two operands take the pipelined loop, three the plain one."""
  xs = [RNG.rand(n).astype(np.float32) for _ in range(n_inputs)]
  with _interpreted():
    got = run_map(_hip.SP_F32, (n,), xs, _long_body(n_inputs, 4), np.float32)
  np.testing.assert_array_equal(got, _long_numpy(xs, 4))


@pytest.mark.parametrize('rounds', [1, 16, 31])
def test_interpreter_long_programs(rounds):
  """Up to 63 instructions: the program is held one instruction per lane of a wave."""
  n = 300 * 1000
  xs = [RNG.rand(n).astype(np.float32) for _ in range(2)]
  with _interpreted():
    got = run_map(_hip.SP_F32, (n,), xs, _long_body(2, rounds), np.float32)
  np.testing.assert_array_equal(got, _long_numpy(xs, rounds))


def test_interpreter_mixed_operand_dtypes_take_the_plain_loop():
  n = 700 * 1000 + 2
  x = RNG.rand(n).astype(np.float32)
  y = RNG.randint(-5, 5, n).astype(np.int32)
  with _interpreted():
    got = run_map(_hip.SP_F32, (n,), [x, y], _long_body(2, 3), np.float32)
  np.testing.assert_array_equal(got, _long_numpy([x, y.astype(np.float32)], 3))


def test_program_that_reads_an_unwritten_register_is_refused():
  x = RNG.rand(64).astype(np.float32)

  def body(p):
    p.emit('ADD', 2, 0, 5)        # register 5: not an operand, never written
    return 2
  with pytest.raises(Exception, match='nothing has written'):
    run_map(_hip.SP_F32, (64,), [x], body, np.float32)

  def body2(p):
    p.emit('SQRT', 2, 0, 7)       # a unary operator does not read its `b`
    return 2
  np.testing.assert_array_equal(run_map(_hip.SP_F32, (64,), [x], body2, np.float32), np.sqrt(x))


# ---- run-time specialised tier (sp_jit.hip) --------------------------------------
def _both_tiers(fn):
  """fn() evaluated on the interpreter kernels and on run-time specialised ones."""
  lib = _hip.lib()
  try:
    assert lib.sp_jit_configure(0, -1) == 0
    want = fn()
    if lib.sp_jit_configure(1, 0) != 1:
      pytest.skip('libhiprtc not loadable on this box')
    fn()             # requests the specialisation (compiled on the background thread) ...
    lib.sp_jit_wait()
    got = fn()       # ... which this launch uses
    # (the stream may already be cached from another shape: the cache key is the program, not the tile)
    assert lib.sp_jit_compiled_count() > 0, 'program was not specialised at run time'
    got2 = fn()   # second call: served from the cache
  finally:
    lib.sp_jit_configure(1, 1 << 22)
  return want, got, got2


def _chain(ops):
  def body(p):
    c = p.add_const(1.25)
    p.emit('CONST', 3, c)
    p.emit(ops[0], 2, 0, 1)
    for op in ops[1:]:
      p.emit(op, 2, 2, 3 if op in ('ADD', 'MUL') else 1)
    return 2
  return body


@pytest.mark.parametrize('cls,dt,ops', [
    (_hip.SP_F32, np.float32, ['MUL', 'ADD', 'SUB', 'MUL', 'MAX']),
    (_hip.SP_F64, np.float64, ['ADD', 'MUL', 'DIV', 'SUB']),
    (_hip.SP_I64, np.int64, ['ADD', 'MUL', 'SUB', 'MIN']),
])
@pytest.mark.parametrize('shape,bshape', [((257, 1024), (257, 1024)), ((256, 512), (256, 1)), ((256, 512), (1, 512)),
                                          ((31, 7, 12), (31, 1, 12)), ((1003,), (1003,))])
def test_jit_map_bit_identical(cls, dt, ops, shape, bshape):
  a = (RNG.rand(*shape) * 8 + 1).astype(dt)
  b = (RNG.rand(*bshape) * 8 + 1).astype(dt)
  body = _chain(ops)
  want, got, got2 = _both_tiers(lambda: run_map(cls, shape, [a, b], body, dt))
  np.testing.assert_array_equal(got, want)
  np.testing.assert_array_equal(got2, want)


@pytest.mark.parametrize('cls,dt', [(_hip.SP_F32, np.float32), (_hip.SP_F64, np.float64), (_hip.SP_I64, np.int64)])
@pytest.mark.parametrize('shape,bshape', [((512, 1024), (512, 1024)), ((512, 1024), (512, 1)), ((8, 65536), (1, 65536)),
                                          ((70000, 8), (70000, 1))])
@pytest.mark.parametrize('op', ['SUM', 'MAX', 'PROD'])
def test_jit_reduce_bit_identical(cls, dt, shape, bshape, op):
  a = (RNG.rand(*shape) + 0.5).astype(dt) if op != 'PROD' else np.ones(shape, dt)
  b = (RNG.rand(*bshape) * 3 + 1).astype(dt)

  def body(p):
    p.emit('MUL', 2, 0, 1)
    p.emit('ADD', 2, 2, 0)
    p.emit('SUB', 2, 2, 1)
    return 2
  for axis in (0, 1, None):
    want, got, got2 = _both_tiers(lambda: run_reduce(a, axis, op, cls, dt, body, extra=[b]))
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(got2, want)


def test_jit_argreduce_bit_identical():
  x = RNG.rand(300, 4096).astype(np.float32)
  x[7, 100] = np.nan
  y = RNG.rand(300, 4096).astype(np.float32)

  def run(axis):
    O, A, I = (1, x.size, 1) if axis is None else ((1, 300, 4096) if axis == 0 else (300, 4096, 1))
    p = Program()
    p.add_input(np.float32, dense_strides((O, A, I)))
    p.add_input(np.float32, dense_strides((O, A, I)))
    p.emit('SUB', 2, 0, 1)
    p.emit('ABS', 2, 2)
    p.result_reg = 2
    prog = p.finish(_hip.SP_F32, (O, A, I), None, True)
    oi = D.empty((O * I,), np.int64)
    ov = D.empty((O * I,), np.float32)
    kernels.argreduce(prog, [dev(x), dev(y)], 0, O, A, I, 0, -7, oi, ov)
    D.synchronize()
    return np.stack([host(oi).astype(np.float64), host(ov).astype(np.float64)])
  for axis in (0, 1, None):
    want, got, got2 = _both_tiers(lambda: run(axis))
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(got2, want)
  idx = run(1)[0].astype(np.int64)
  keep = np.arange(300) != 7
  np.testing.assert_array_equal(idx[keep], np.argmax(np.abs(x - y)[keep], axis=1))
  assert idx[7] == -7


# ---- k-means tile kernels (kmeans.hip) --------------------------------------------
def _nearest(x, c, tier):
  labels = D.empty((x.shape[0],), np.int64)
  kernels.nearest_center(dev(x), dev(c), labels, tier)
  D.synchronize()
  return host(labels)


@pytest.mark.parametrize('n,k,d', [(1, 1, 1), (100, 10, 5), (257, 70, 33), (1000, 130, 64), (64, 200, 7)])
@pytest.mark.parametrize('xdt,cdt', [(np.float32, np.float64), (np.float32, np.float32), (np.float64, np.float64)])
def test_nearest_center_exact_is_cdist_argmin(n, k, d, xdt, cdt):
  from scipy.spatial.distance import cdist
  x = RNG.rand(n, d).astype(xdt)
  c = RNG.rand(k, d).astype(cdt)
  want = np.argmin(cdist(x, c), axis=1)
  np.testing.assert_array_equal(_nearest(x, c, _hip.NEAREST_EXACT), want)


@pytest.mark.parametrize('n,k,d', [(5000, 300, 64), (4097, 129, 50), (3000, 1024, 256), (2048, 16, 8), (130, 5, 3),
                                   (6000, 520, 100), (3000, 260, 36)])   # aligned rows, partial last k-tile
@pytest.mark.parametrize('tier', ['FUSED', 'SPLIT'])
def test_nearest_center_fused_equals_exact(n, k, d, tier):
  """An MFMA tier (fp32, or bf16-split) + exact re-check of near ties gives the exact tier's labels, bit for bit."""
  from scipy.spatial.distance import cdist
  x = RNG.rand(n, d).astype(np.float32)
  c = RNG.rand(k, d)
  c[k // 2] = c[0]                       # exact duplicate centre: ties must go to the lower index
  x[: min(n, k)] = c[: min(n, k)].astype(np.float32)   # points sitting (almost) on centres
  fused = _nearest(x, c, getattr(_hip, 'NEAREST_' + tier))
  exact = _nearest(x, c, _hip.NEAREST_EXACT)
  np.testing.assert_array_equal(fused, exact)
  np.testing.assert_array_equal(exact, np.argmin(cdist(x, c), axis=1))
  assert not np.any(fused == k // 2) or not np.array_equal(c[k // 2], c[0])


@pytest.mark.parametrize('n,k,d', [(9000, 40, 24), (6000, 600, 64), (300, 3000, 16), (4000, 64, 32)])
@pytest.mark.parametrize('tier', ['FUSED', 'SPLIT'])
def test_nearest_center_everything_undecided(n, k, d, tier):
  """Every centre exists twice, so no point can be decided by the fp32 pass: with more listed points than the
  candidate masks hold (n / 8, at least 4096) the list goes to the exact kernel, below that to the MFMA re-check
  (here with more than 64 mask words per point at k = 3000) -- the labels are the exact tier's either way."""
  from scipy.spatial.distance import cdist
  x = RNG.rand(n, d).astype(np.float32)
  half = RNG.rand(k // 2, d)
  c = np.concatenate([half, half], axis=0)
  fused = _nearest(x, c, getattr(_hip, 'NEAREST_' + tier))
  want = np.argmin(cdist(x, c), axis=1)
  np.testing.assert_array_equal(fused, want)
  assert fused.max() < k // 2
  unchecked = _nearest(x, c, getattr(_hip, 'NEAREST_%s_UNCHECKED' % tier))
  assert np.all(unchecked < 0)           # the first pass listed every point


@pytest.mark.parametrize('n,k,d', [(512 * 128 + 4000, 600, 32), (700, 2304, 40), (512 * 128, 513, 32)])
@pytest.mark.parametrize('tier', ['FUSED', 'SPLIT'])
def test_nearest_center_split_tail(n, k, d, tier):
  """The partly filled last round of first-pass workgroups runs split over ranges of centre blocks and is merged
  afterwards (3 ranges; 9 blocks in 5 ranges of 2,2,2,2,1; a tile of whole rounds only): same labels as the exact
  tier, with the duplicate centres in different ranges."""
  from scipy.spatial.distance import cdist
  x = RNG.rand(n, d).astype(np.float32)
  c = RNG.rand(k, d)
  c[k - 1] = c[3]                        # tie across the first and the last range
  x[:50] = c[3].astype(np.float32)
  fused = _nearest(x, c, getattr(_hip, 'NEAREST_' + tier))
  want = np.argmin(cdist(x[-5000:], c), axis=1)
  np.testing.assert_array_equal(fused[-5000:], want)
  np.testing.assert_array_equal(fused[:50], np.argmin(cdist(x[:50], c), axis=1))
  assert not np.any(fused == k - 1)
  np.testing.assert_array_equal(fused, _nearest(x, c, _hip.NEAREST_EXACT))


@pytest.mark.parametrize('case', ['wide_range', 'signed', 'tiny', 'denormal', 'huge', 'clustered'])
def test_nearest_center_bf16_split_on_hard_data(case):
  """The split tier's error window must hold whatever the data look like: values spread over 12 orders of
  magnitude inside a row (the bf16 cut is relative to each VALUE, the bound to the row norms), both signs, tiny
  and huge scales, and points packed around the centres (many near ties).  Labels = argmin(cdist) in fp64."""
  from scipy.spatial.distance import cdist
  rng = np.random.RandomState(99)
  n, k, d = 5000, 300, 96
  x = rng.rand(n, d)
  c = rng.rand(k, d)
  if case == 'wide_range':
    x *= 10.0 ** rng.randint(-6, 6, size=(n, d))
    c *= 10.0 ** rng.randint(-6, 6, size=(k, d))
  elif case == 'signed':
    x, c = x - 0.5, c - 0.5
  elif case == 'tiny':
    x, c = x * 1e-18, c * 1e-18
  elif case == 'denormal':               # (below fp32's / bf16's normal range: the filter must not decide anything)
    x, c = x * 1e-38, c * 1e-38
  elif case == 'huge':
    x, c = x * 1e15, c * 1e15
  else:
    x = c[rng.randint(0, k, size=n)] + rng.randn(n, d) * 1e-4
  x = x.astype(np.float32)
  got = _nearest(x, c, _hip.NEAREST_SPLIT)
  dd = cdist(x.astype(np.float64), c)
  want = np.argmin(dd, axis=1)
  differ = np.nonzero(got != want)[0]
  assert all(dd[i, got[i]] == dd[i, want[i]] for i in differ), differ[:5]     # (exactly equal distances: either index)
  listed = int((_nearest(x, c, _hip.NEAREST_SPLIT_UNCHECKED) < 0).sum())
  assert listed < n or case in ('clustered', 'tiny', 'denormal')   # (below 1e-28 in |x||c| the filter leaves every point to the exact stage)


def test_nearest_center_prepared_points_are_the_same_call():
  from scipy.spatial.distance import cdist
  x = RNG.rand(20000, 72).astype(np.float32)
  big = np.pad(x, ((0, 0), (4, 4)))
  xt = dev(big)[:, 4:76]                 # strided rows: the images are dense whatever the points' layout
  prepared = kernels.prepare_points(xt)
  for seed in (1, 2):
    c = np.random.RandomState(seed).rand(200, 72)
    labels = D.empty((20000,), np.int64)
    kernels.nearest_center(xt, dev(c), labels, prepared=prepared)
    D.synchronize()
    np.testing.assert_array_equal(host(labels), np.argmin(cdist(x, c), axis=1))


@pytest.mark.parametrize('layout', ['shuffled', 'every_third_row_far_away'])
def test_nearest_center_prepared_points_shift_from_a_sample_of_the_rows(layout):
  """A prepared buffer of more than 131 072 points takes its shift from every (n / 65 536)-th row (kmeans.hip:
  sp_kmeans_points_prepare).  The shift is free -- labels are argmin(cdist) whatever it is -- including when the sampled
  rows are NOT representative: here every third row (exactly the sampled ones, n / 65536 = 3) sits 50 units away from
  the rest, strided rows on top."""
  from scipy.spatial.distance import cdist
  rng = np.random.RandomState(12)
  n, d, k = 3 * 65536 + 11, 40, 300
  x = rng.rand(n, d).astype(np.float32)
  if layout == 'every_third_row_far_away':
    x[::3] += 50.0
  c = x[rng.choice(n, k, replace=False)].astype(np.float64) + rng.randn(k, d) * 1e-3
  xt = dev(np.pad(x, ((0, 0), (0, 8))))[:, :d]
  prepared = kernels.prepare_points(xt)
  labels = D.empty((n,), np.int64)
  kernels.nearest_center(xt, dev(c), labels, prepared=prepared)
  D.synchronize()
  got = host(labels)
  want = np.empty(n, np.int64)
  for a in range(0, n, 32768):
    want[a:a + 32768] = np.argmin(cdist(x[a:a + 32768].astype(np.float64), c), axis=1)
  np.testing.assert_array_equal(got, want)


def test_bf16_mfma_accumulation_model():
  """kmeans_split.hpp bounds the fp32 accumulation of the bf16 MFMA by one rounding to nearest per product (3 D of
  them) plus the terms it leaves out.  Checked here on the kernel's own scores: with ONE centre the unchecked tier
  leaves a point undecided never (the second best is +inf), so the test reads the scores back through the
  re-check's window instead -- |score_fp64 - score_kernel| <= E for every (point, centre), which is what makes the
  candidate list complete: the exact argmin is always among the centres marked for a listed point."""
  from scipy.spatial.distance import cdist
  rng = np.random.RandomState(5)
  for d in (32, 256, 1000):
    n, k = 4096, 512
    x = (rng.rand(n, d) * 10.0 ** rng.randint(-3, 3, size=(n, d))).astype(np.float32)
    base = rng.rand(k // 2, d)
    c = np.concatenate([base, base * (1 + 1e-7)], axis=0)       # every centre has a near twin: all points listed
    got = _nearest(x, c, _hip.NEAREST_SPLIT)
    dd = cdist(x.astype(np.float64), c)
    want = np.argmin(dd, axis=1)
    differ = np.nonzero(got != want)[0]
    assert all(dd[i, got[i]] == dd[i, want[i]] for i in differ), (d, differ[:5])


def test_nearest_center_strided_rows_and_auto_tier():
  from scipy.spatial.distance import cdist
  big = RNG.rand(6000, 96).astype(np.float32)
  c = RNG.rand(64, 80).astype(np.float32)
  xt = dev(big)[:, 8:88]                 # row stride 96, 80 features, 32-B offset
  labels = D.empty((6000,), np.int64)
  kernels.nearest_center(xt, dev(c), labels)
  D.synchronize()
  np.testing.assert_array_equal(host(labels), np.argmin(cdist(big[:, 8:88], c), axis=1))


@pytest.mark.parametrize('n,k', [(0, 4), (1, 1), (1000, 7), (100000, 1024), (5000, 16384)])
def test_bincount(n, k):
  lab = RNG.randint(0, k, size=n).astype(np.int64)
  counts = D.empty((k,), np.int64)
  kernels.bincount(dev(lab) if n else D.empty((0,), np.int64), k, counts)
  D.synchronize()
  np.testing.assert_array_equal(host(counts), np.bincount(lab, minlength=k))


@pytest.mark.parametrize('dt', [np.float32, np.float64])
@pytest.mark.parametrize('n,k,d', [(1, 1, 1), (200, 5, 6), (5000, 37, 70), (20000, 1024, 256), (3000, 3, 130)])
def test_segment_sum_is_numpy_masked_sum(n, k, d, dt):
  x = (RNG.rand(n, d) * 100).astype(dt)
  lab = RNG.randint(0, k, size=n).astype(np.int64)
  if k > 2:
    lab[lab == 1] = 0                    # an empty cluster
  out = D.empty((k, d), dt)
  kernels.segment_sum(dev(x), dev(lab), k, out)
  D.synchronize()
  want = np.zeros((k, d), dt)
  for i in range(k):
    want[i] = x[lab == i].sum(axis=0)    # k_means_.py:91-95
  got = host(out)
  small = np.bincount(lab, minlength=k) <= 512
  # labels with <= 512 rows are added in NumPy's own (sequential) order: bit-identical
  np.testing.assert_array_equal(got[small], want[small])
  # larger ones in 512-row chunks combined in order: rounding-level differences only
  np.testing.assert_allclose(got[~small], want[~small], rtol=1e-5 if dt == np.float32 else 1e-13)


def test_segment_sum_is_deterministic_and_balanced():
  """One huge label next to tiny ones (the shape k-means produces): same bits on every run."""
  n, k, d = 60000, 64, 128
  x = (RNG.rand(n, d) * 10).astype(np.float32)
  lab = np.zeros(n, np.int64)
  lab[::7] = RNG.randint(1, k, size=len(lab[::7]))
  outs = []
  for _ in range(3):
    out = D.empty((k, d), np.float32)
    kernels.segment_sum(dev(x), dev(lab), k, out)
    D.synchronize()
    outs.append(host(out))
  np.testing.assert_array_equal(outs[0], outs[1])
  np.testing.assert_array_equal(outs[0], outs[2])
  want = np.stack([x[lab == i].astype(np.float64).sum(axis=0) for i in range(k)])
  np.testing.assert_allclose(outs[0], want, rtol=2e-5)


@pytest.mark.parametrize('n,k,d', [(1, 1, 1), (5000, 7, 3), (70000, 1024, 16), (3000, 16384, 4)])
def test_segment_sum_counts_are_numpy_bincount(n, k, d):
  """sp_segment_sum_counts: the same sums as sp_segment_sum, and np.bincount(labels, minlength=k) (labels outside
  [0, k) ignored, as sp_bincount_i64) out of the counting sort -- k_means_.py:69-72 and :75-97 in one call."""
  x = (RNG.rand(n, d) * 10).astype(np.float32)
  lab = RNG.randint(0, k, size=n).astype(np.int64)
  if n > 10:
    lab[3], lab[7] = -1, k + 5             # ignored
  plain, both, counts = D.empty((k, d), np.float32), D.empty((k, d), np.float32), D.empty((k,), np.int64)
  kernels.segment_sum(dev(x), dev(lab), k, plain)
  kernels.segment_sum(dev(x), dev(lab), k, both, counts)
  D.synchronize()
  np.testing.assert_array_equal(host(both), host(plain))
  ok = lab[(lab >= 0) & (lab < k)]
  np.testing.assert_array_equal(host(counts), np.bincount(ok, minlength=k))
  alone = D.empty((k,), np.int64)
  kernels.bincount(dev(lab), k, alone)
  np.testing.assert_array_equal(host(counts), host(alone))


def test_a_fit_counts_with_its_segment_sums_and_converts_no_labels(monkeypatch):
  """Inside KMeans.fit on the HIP backend the count join takes the counts the segment sum of the same labels made
  (no bincount launch), and the joins read the int64 labels the assignment produced instead of converting the float32
  target back (no astype launch) -- with centres bit-identical to the same fit with both shortcuts switched off."""
  import spartan_amd as sp
  from spartan_amd.examples.sklearn.cluster import KMeans
  calls = {'bincount': 0}
  real = kernels.bincount
  monkeypatch.setattr(kernels, 'bincount', lambda *a, **kw: (calls.__setitem__('bincount', calls['bincount'] + 1), real(*a, **kw))[1])
  x = RNG.rand(21000, 32).astype(np.float32)       # (3 x 7000: the label target is cut like the points)
  init = RNG.rand(50, 32)
  out = {}
  for mode in ('shortcuts', 'plain'):
    ctx = sp.initialize('hip', num_workers=3)
    try:
      if mode == 'plain':
        monkeypatch.setattr(type(ctx.backend), 'fixed_points', None, raising=False)
      calls['bincount'] = 0
      c, lab = KMeans(50, 3).fit(sp.from_numpy(x), init.copy(), implementation='map2', reducer=np.add)
      out[mode] = (np.asarray(c), lab.glom(), calls['bincount'])
    finally:
      sp.shutdown()
  np.testing.assert_array_equal(out['shortcuts'][0], out['plain'][0])
  np.testing.assert_array_equal(out['shortcuts'][1], out['plain'][1])
  assert out['plain'][2] >= 3 * 3 and out['shortcuts'][2] == 0, (out['plain'][2], out['shortcuts'][2])


# ---- fp64 GEMM (gemm_f64.hip) ------------------------------------------------------
@pytest.mark.parametrize('mnk', [(1, 1, 1), (16, 16, 4), (128, 128, 8), (130, 70, 33), (257, 300, 129), (64, 2, 1024),
                                 (512, 384, 256)])
def test_gemm_f64_integer_valued_exact(mnk):
  M, N, K = mnk
  a = RNG.randint(-4, 5, size=(M, K)).astype(np.float64)
  b = RNG.randint(-4, 5, size=(K, N)).astype(np.float64)
  c = D.empty((M, N), np.float64)
  kernels.gemm_f32(dev(a), dev(b), c)
  D.synchronize()
  np.testing.assert_array_equal(host(c), a.dot(b))
  kernels.gemm_f32(dev(a), dev(b), c, accumulate=True)     # C += A.B (the np.add merge fused in)
  D.synchronize()
  np.testing.assert_array_equal(host(c), 2 * a.dot(b))


def test_gemm_f64_random_tolerance_and_strides():
  a = RNG.rand(300, 520)
  b = RNG.rand(520, 260)
  big_a = dev(np.pad(a, ((0, 0), (0, 8))))[:, :520]        # lda = 528
  big_c = D.zeros((300, 264), np.float64)
  kernels.gemm_f32(big_a, dev(b), big_c[:, :260])
  D.synchronize()
  np.testing.assert_allclose(host(big_c)[:, :260], a.dot(b), rtol=1e-13)
  assert np.all(host(big_c)[:, 260:] == 0)


def test_dot_float64_uses_the_mfma_gemm():
  import spartan_amd as sp
  ctx = sp.initialize('hip', num_workers=3)
  try:
    a = RNG.randint(-3, 4, size=(200, 96)).astype(np.float64)
    b = RNG.randint(-3, 4, size=(96, 150)).astype(np.float64)
    ctx.backend.gemm_events = []
    got = sp.dot(sp.from_numpy(a), sp.from_numpy(b)).glom()
    assert got.dtype == np.float64 and len(ctx.backend.gemm_events) > 0
    ctx.backend.gemm_events = None
    np.testing.assert_array_equal(got, a.dot(b))
  finally:
    sp.shutdown()


@pytest.mark.parametrize('dt', [np.float32, np.float64])
@pytest.mark.parametrize('mnk', [(64, 64, 100000), (10, 10, 65536), (512, 384, 20000), (130, 70, 4099), (1, 1, 5000)])
def test_gemm_split_k(mnk, dt):
  """Small output, long contraction (x^T x): the split-K path -- exact for integer-valued operands,
  same tolerance as the plain kernel otherwise, identical bits on every run, `accumulate` honoured."""
  M, N, K = mnk
  assert _hip.lib().sp_gemm_workspace_bytes(_hip.sp_dtype(dt), M, N, K) > 0
  a = RNG.randint(-3, 4, size=(M, K)).astype(dt)
  b = RNG.randint(-3, 4, size=(K, N)).astype(dt)
  c = D.full((M, N), 5, dt)
  kernels.gemm_f32(dev(a), dev(b), c)
  D.synchronize()
  want = a.astype(np.float64).dot(b.astype(np.float64))
  np.testing.assert_array_equal(host(c), want.astype(dt))
  kernels.gemm_f32(dev(a), dev(b), c, accumulate=True)
  D.synchronize()
  np.testing.assert_array_equal(host(c), (2 * want).astype(dt))
  x = (RNG.rand(M, K) - 0.5).astype(dt)
  y = (RNG.rand(K, N) - 0.5).astype(dt)
  outs = []
  for _ in range(2):
    kernels.gemm_f32(dev(x), dev(y), c)
    D.synchronize()
    outs.append(host(c).copy())
  np.testing.assert_array_equal(outs[0], outs[1])
  ref = x.astype(np.float64).dot(y.astype(np.float64))
  np.testing.assert_allclose(outs[0], ref, rtol=0, atol=(2e-4 if dt == np.float32 else 1e-10) * np.sqrt(K))


@pytest.mark.parametrize('mnk', [(2304, 2304, 2304),      # 162 tiles of 256 x 128: all remainder, every tile cut
                                 (3000, 3000, 3000),      # ragged edges and a K tail on the cut tiles
                                 (4608, 4608, 1024),      # one whole round of 512 tiles + 136 balanced
                                 (3584, 3584, 1000),      # short contraction with a tail
                                 (2300, 2304, 2056)])
def test_gemm_balanced_remainder(mnk):
  """Tile counts that do not fill the resident workgroups: the balanced kernel (whole data-parallel rounds, then the
  k-tiles of the remaining tiles cut into equal ranges; partial images added in k order by the fix-up pass).  Exact
  for integer-valued operands, `accumulate` honoured, identical bits on every run, within the plain kernel's
  tolerance otherwise, padded leading dimensions."""
  M, N, K = mnk
  assert _hip.lib().sp_gemm_workspace_bytes(_hip.SP_F32, M, N, K) > 0
  a = RNG.randint(-3, 4, size=(M, K)).astype(np.float32)
  b = RNG.randint(-3, 4, size=(K, N)).astype(np.float32)
  want = a.astype(np.float64).dot(b.astype(np.float64)).astype(np.float32)
  c = D.full((M, N), 5, np.float32)
  kernels.gemm_f32(dev(a), dev(b), c)
  np.testing.assert_array_equal(host(c), want)
  kernels.gemm_f32(dev(a), dev(b), c, accumulate=True)
  np.testing.assert_array_equal(host(c), 2 * want)
  big_a, big_b, big_c = D.zeros((M, K + 8), np.float32), D.zeros((K, N + 4), np.float32), D.zeros((M, N + 12), np.float32)
  big_a[:, :K] = dev(a)
  big_b[:, :N] = dev(b)
  kernels.gemm_f32(big_a[:, :K], big_b[:, :N], big_c[:, :N])
  np.testing.assert_array_equal(host(big_c)[:, :N], want)
  assert not host(big_c)[:, N:].any()
  x = (RNG.rand(M, K) - 0.5).astype(np.float32)
  y = (RNG.rand(K, N) - 0.5).astype(np.float32)
  outs = []
  for _ in range(2):
    kernels.gemm_f32(dev(x), dev(y), c)
    outs.append(host(c).copy())
  np.testing.assert_array_equal(outs[0], outs[1])
  ref = x.astype(np.float64).dot(y.astype(np.float64))
  assert np.abs(outs[0] - ref).max() <= 2 * K * np.finfo(np.float32).eps * 0.25


def test_gemm_balanced_forced_fuzz():
  """SP_GEMM_SK=1 sends every shape the balanced kernel applies to through it (the default only where its cost model
  predicts a win): tools/fuzz_gemm.py's random shapes / paddings / accumulate flags, all exact."""
  import os
  import subprocess
  import sys
  tools = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools')
  out = subprocess.run([sys.executable, os.path.join(tools, 'fuzz_gemm.py'), '11'], cwd=tools, capture_output=True, text=True,
                       env=dict(os.environ, SP_GEMM_SK='1'), timeout=600)
  assert out.returncode == 0, out.stderr[-2000:]
  assert ' 0 bad' in out.stdout, out.stdout[-2000:]


@pytest.mark.parametrize('mnk', [(2048, 2048, 256), (2048, 1920, 1000), (40, 4096, 4096), (2000, 2048, 72)])
def test_gemm_64x128_tiles(mnk):
  """Shapes whose 128 x 128 tiling leaves one workgroup per CU (or fewer): the 64 x 128 macro-tile."""
  M, N, K = mnk
  a = RNG.randint(-3, 4, size=(M, K)).astype(np.float32)
  b = RNG.randint(-3, 4, size=(K, N)).astype(np.float32)
  c = D.full((M, N), 5, np.float32)
  kernels.gemm_f32(dev(a), dev(b), c)
  want = a.astype(np.float64).dot(b.astype(np.float64)).astype(np.float32)
  np.testing.assert_array_equal(host(c), want)
  kernels.gemm_f32(dev(a), dev(b), c, accumulate=True)
  np.testing.assert_array_equal(host(c), 2 * want)


def test_jit_code_objects_persist_across_processes(tmp_path):
  """SPARTAN_JIT_CACHE=<dir>: the first process compiles a run-time specialised kernel and leaves its code object in
  the directory, the second loads it instead of compiling (same result; `loaded ... from` in the verbose log)."""
  import subprocess
  import sys
  import os
  prog = (
      "import numpy as np, spartan_amd as sp\n"
      "sp.initialize('hip')\n"
      "x = sp.from_numpy(np.arange(1 << 16, dtype=np.float32).reshape(256, 256) / 7)\n"
      "r = (sp.sqrt(sp.abs(x * 3 - 2)) * x + x / 5 - 1).optimized().glom()\n"
      "print('SUM %.6f' % float(r.astype(np.float64).sum()))\n")
  env = dict(os.environ, SPARTAN_JIT_CACHE=str(tmp_path), SP_JIT_SYNC='1', SP_JIT_MIN_ELEMS='0', SP_JIT_VERBOSE='1',
             SP_JIT_PRELOAD='0')       # (the background preload of the backend logs `preloaded N code objects`)
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  outs = []
  for _ in range(2):
    p = subprocess.run([sys.executable, '-c', prog], env=env, cwd=root, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    outs.append(p)
  files = [f for f in os.listdir(str(tmp_path)) if f.endswith('.spco')]
  assert files, 'no code object was written'
  assert 'compiled' in outs[0].stderr and 'loaded' not in outs[0].stderr
  assert 'loaded' in outs[1].stderr and ' compiled ' not in outs[1].stderr
  assert outs[0].stdout.strip().splitlines()[-1] == outs[1].stdout.strip().splitlines()[-1]
  # with the preload on (the default) a third process finds the code object already loaded: nothing compiled,
  # nothing read on the launch path, same result
  env3 = dict(env, SP_JIT_PRELOAD='1')
  del env3['SP_JIT_SYNC']        # (set at all = compile / load in the caller)
  wait = prog.replace("sp.initialize('hip')\n", "sp.initialize('hip')\nimport time; time.sleep(1.0)\n")
  p3 = subprocess.run([sys.executable, '-c', wait], env=env3, cwd=root, capture_output=True, text=True, timeout=300)
  assert p3.returncode == 0, p3.stderr[-2000:]
  assert 'preloaded' in p3.stderr and ' compiled ' not in p3.stderr and 'loaded sp_map_kernel' not in p3.stderr, p3.stderr[-2000:]
  assert p3.stdout.strip().splitlines()[-1] == outs[0].stdout.strip().splitlines()[-1]


# ---- one-pass least-squares gradient (rowdot.hip) ------------------------------------------------
@pytest.mark.parametrize('n,d,pad', [(1000, 4096, 0), (257, 64, 0), (5, 260, 4), (3000, 4092, 8), (1, 4, 0), (70000, 512, 0)])
@pytest.mark.parametrize('with_y', [True, False])
def test_rowdot_colsum_kernel(n, d, pad, with_y):
  """out[c] = sum_i x[i, c] * (x[i, :] . w - y[i]) against float64 NumPy: the error of an fp32 row dot of d terms
  carried through a column sum of n terms; rows in a padded buffer; `accumulate`; identical bits on every run."""
  x = (RNG.rand(n, d) - 0.5).astype(np.float32)
  w = (RNG.rand(d) - 0.5).astype(np.float32)
  y = (RNG.rand(n) - 0.5).astype(np.float32) if with_y else None
  big = D.zeros((n, d + pad), np.float32)
  big[:, :d] = dev(x)
  xd = big[:, :d]
  out = D.full((d,), 3.0, np.float32)
  assert kernels.rowdot_colsum(xd, dev(w), dev(y) if with_y else None, out)
  x64 = x.astype(np.float64)
  t = x64.dot(w.astype(np.float64))
  r = t - y if with_y else t
  want = (x64 * r[:, None]).sum(0)
  scale = (np.abs(x64) * (np.abs(x64).dot(np.abs(w.astype(np.float64))) + (np.abs(y) if with_y else 0))[:, None]).sum(0)
  eps = np.finfo(np.float32).eps
  got = host(out)
  assert np.all(np.abs(got - want) <= (d + n + 8) * eps * scale + 1e-30)
  first = got.copy()
  assert kernels.rowdot_colsum(xd, dev(w), dev(y) if with_y else None, out)
  np.testing.assert_array_equal(host(out), first)
  assert kernels.rowdot_colsum(xd, dev(w), dev(y) if with_y else None, out, accumulate=True)
  np.testing.assert_array_equal(host(out), first + first)
  # layouts the kernel does not take are reported, not guessed at
  if d > 4:
    assert kernels.rowdot_colsum(big[:, 1:d - 3], dev(w[:d - 4]), None, D.empty((d - 4,), np.float32)) is False


def test_lreg_gradient_runs_as_one_pass_per_tile():
  """examples.lreg through the expression API: the gradient DAG is rewritten (expr/rowdot.py) and each step is one
  pass over a row tile; the weights agree with the two-launch form (rewrite off) to rounding."""
  import importlib
  import spartan_amd as sp
  from spartan_amd.examples import lreg
  from spartan_amd.expr.rowdot import RowDotColSumExpr
  optimize = importlib.import_module('spartan_amd.expr.optimize')
  ctx = sp.initialize('hip', num_workers=3)
  try:
    rng = np.random.RandomState(4)
    xh, yh = rng.rand(3001, 256).astype(np.float32), rng.rand(3001, 1).astype(np.float32)
    w = rng.rand(256, 1).astype(np.float32)
    x, y = sp.Val(val=sp.from_numpy(xh).force()), sp.Val(val=sp.from_numpy(yh).force())
    g = lreg.gradient(x, y, w).optimized()
    assert isinstance(g, RowDotColSumExpr)
    calls = []
    inner = ctx.backend.rowdot_colsum
    ctx.backend.rowdot_colsum = lambda *a: (calls.append(1), inner(*a))[1]
    got = g.glom()
    del ctx.backend.rowdot_colsum
    assert len(calls) == len(x.val.tiles)                            # one pass per row tile
    want = (xh.astype(np.float64) * (xh.astype(np.float64).dot(w.astype(np.float64)) - yh)).sum(0)
    np.testing.assert_allclose(got, want, rtol=1e-4)
    w1 = lreg.fit(x, y, 10, alpha=1e-5, w=w)
    optimize.FLAGS['opt_rowdot_fusion'] = False
    try:
      w0 = lreg.fit(x, y, 10, alpha=1e-5, w=w)
    finally:
      optimize.FLAGS['opt_rowdot_fusion'] = True
    np.testing.assert_allclose(w1, w0, rtol=1e-5, atol=2e-6)      # fp32 sums in two different orders, ten steps
  finally:
    sp.shutdown()


@pytest.mark.parametrize('workers', [1, 8])
@pytest.mark.parametrize('n,d', [(17, 4), (1000, 4096), (5000, 132), (64, 260)])
def test_lreg_gradient_rewrite_shapes(workers, n, d):
  """The rewritten gradient through the expression API on ragged row tilings and the smallest / largest widths,
  with and without y, against float64 NumPy."""
  import spartan_amd as sp
  from spartan_amd.expr.rowdot import RowDotColSumExpr
  sp.initialize('hip', num_workers=workers)
  try:
    rng = np.random.RandomState(n + d)
    xh, yh = (rng.rand(n, d) - 0.5).astype(np.float32), (rng.rand(n, 1) - 0.5).astype(np.float32)
    w = (rng.rand(d, 1) - 0.5).astype(np.float32)
    x, y = sp.Val(val=sp.from_numpy(xh).force()), sp.Val(val=sp.from_numpy(yh).force())
    x64, w64 = xh.astype(np.float64), w.astype(np.float64)
    for build, want in ((lambda: sp.sum(x * (sp.dot(x, w) - y), axis=0), (x64 * (x64.dot(w64) - yh)).sum(0)),
                        (lambda: sp.sum((sp.dot(x, w) - y) * x, axis=0), (x64 * (x64.dot(w64) - yh)).sum(0)),
                        (lambda: sp.sum(x * sp.dot(x, w), axis=0), (x64 * x64.dot(w64)).sum(0))):
      e = build().optimized()
      assert isinstance(e, RowDotColSumExpr)
      got = e.glom()
      assert got.dtype == np.float32 and got.shape == (d,)
      scale = (np.abs(x64) * (np.abs(x64).dot(np.abs(w64)) + np.abs(yh))).sum(0)
      assert np.all(np.abs(got - want) <= (n + d + 8) * np.finfo(np.float32).eps * scale + 1e-30)
  finally:
    sp.shutdown()
