"""Parity tests proper: the same Spartan programs through the HIP backend
(fused-map / reduce / argreduce / merge / GEMM kernels behind the C-ABI) on
the MI355X, for 1, 3 and 8 logical workers hosted on the one GPU, against NumPy
on identical inputs (bit-exact for integer/index/comparison results and for
integer-valued fp32 data; stated tolerances otherwise -- tests/programs.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import spartan_amd as sp  # noqa: E402
from tests import programs  # noqa: E402

PROGS = programs.programs()


@pytest.fixture(params=[1, 3, 8], ids=lambda n: 'workers%d' % n)
def ctx(request):
  c = sp.initialize('hip', num_workers=request.param)
  yield c
  sp.shutdown()


@pytest.mark.parametrize('prog', PROGS, ids=[p[0] for p in PROGS])
def test_program(ctx, prog):
  name, build, expected, tol = prog
  got = programs.run(name, build, sp, ctx.num_workers)
  programs.check(name, got, expected(), tol)


def test_hip_backend_is_the_one_running(ctx):
  from spartan_amd import _hip
  from spartan_amd import devarray as D
  before = ctx.backend.launches
  r = (sp.ones((256, 256)) + 1).force()
  assert ctx.backend.launches > before
  t = ctx.tile(list(r.tiles.values())[0]).data
  # the tile is a view of a blob of the library's own store -- not a tensor of some other framework
  assert isinstance(t, D.DevArray) and t.storage.on_device and D.blob_stats()[0] >= 1
  assert _hip.lib().sp_abi_version() == 1


def test_user_callables_traced_then_device_tiles_then_host(ctx):
  """`map` takes any Python callable, as the reference does (FnCallExpr.evaluate, local.py:115-127): an element-wise
  function becomes part of the fused kernel (traced); one that is not runs on the device tiles, which answer ndarray
  calls with kernels; one that needs more than they offer runs on host copies -- announced, counted, never silent.
  A function NumPy itself refuses is refused."""
  import warnings
  be = ctx.backend
  ones = lambda: sp.ones((8, 8))    # noqa: E731
  np.testing.assert_array_equal(sp.map(ones(), fn=lambda x: x * 3 + 1).glom(), np.full((8, 8), 4, np.float32))
  assert be.host_round_trips == 0
  # not element-wise, but within what device tiles do: reductions, views, NumPy functions
  got = sp.map(sp.arange((8, 8), dtype=np.float32), fn=lambda x: x - x.sum(axis=1, keepdims=True) / 8 + np.maximum(x.T, 0).T * 0).glom()
  ref = np.arange(64, dtype=np.float32).reshape(8, 8)
  np.testing.assert_array_equal(got, ref - ref.sum(axis=1, keepdims=True) / 8)
  assert be.host_round_trips == 0
  # beyond them: host copies
  with warnings.catch_warnings(record=True) as seen:
    warnings.simplefilter('always')
    got = sp.map(sp.arange((8, 8), dtype=np.float32), fn=lambda x: np.cumsum(x, axis=1)).glom()
  np.testing.assert_array_equal(got, np.cumsum(ref, axis=1))
  assert be.host_round_trips >= 1 and any('host copies' in str(w.message) for w in seen)
  with pytest.raises(ValueError):
    sp.map(ones(), fn=lambda x: x + 1 if x > 0 else x).force()      # ambiguous truth value: NumPy's own error


def test_fused_map_is_one_launch_per_tile(ctx):
  a = sp.from_numpy(np.arange(4096, dtype=np.float32).reshape(64, 64)).force()
  av = sp.Val(val=a)
  e = (av * av + av - 1).optimized()
  before = ctx.backend.launches
  r = e.force()
  assert ctx.backend.launches - before == len(r.tiles)
  np.testing.assert_array_equal(r.glom(), a.glom() * a.glom() + a.glom() - 1)


@pytest.mark.parametrize('n', [1024, 2000])
def test_dot_random_tolerance(ctx, n):
  # |dC| <= 2 K eps max|a| max|b|  (uniform [-1,1) operands)
  rng = np.random.RandomState(20150708)
  a = (rng.rand(n, n) * 2 - 1).astype(np.float32)
  b = (rng.rand(n, n) * 2 - 1).astype(np.float32)
  got = sp.dot(sp.from_numpy(a), sp.from_numpy(b)).glom()
  ref = a.astype(np.float64).dot(b.astype(np.float64))
  assert np.abs(got - ref).max() <= 2 * n * np.finfo(np.float32).eps


def test_planted_argmax_across_tiles(ctx):
  # BASELINE config 3 style: known maxima with duplicates across tiles
  rng = np.random.RandomState(7)
  x = rng.rand(512, 384).astype(np.float32)
  x[100, 7] = 5.0
  x[400, 7] = 5.0          # duplicate in another tile: the first one wins
  x[33, 380] = 9.0
  x[300, 2] = 9.0
  dx = sp.from_numpy(x)
  assert int(sp.argmax(dx).glom()) == int(np.argmax(x))
  np.testing.assert_array_equal(sp.argmax(dx, 0).glom(), np.argmax(x, 0))
  np.testing.assert_array_equal(sp.argmax(dx, 1).glom(), np.argmax(x, 1))
  np.testing.assert_array_equal(sp.argmin(dx, 0).glom(), np.argmin(x, 0))


@pytest.mark.gpu
def test_trees_larger_than_one_kernel_are_carved_into_several_launches():
  """More operands than a kernel has input slots, and more operators than a program has instructions
  (include/spartan_hip.h SP_MAX_INPUTS = 8, SP_MAX_INSTR = 64): the lowered tree is cut into launches instead of
  being refused -- results are NumPy's, bit for bit on integer-valued data."""
  ctx = sp.initialize('hip', num_workers=2)
  try:
    rng = np.random.RandomState(3)
    arrays = [rng.randint(-4, 5, size=(64, 48)).astype(np.float32) for _ in range(12)]
    exprs = [sp.from_numpy(a) for a in arrays]

    def many(*t):                      # ONE local function over 12 tiles (no sub-maps to split at)
      return (t[0] + t[1]) * t[2] - t[3] * t[4] + t[5] - (t[6] + t[7] * t[8]) * t[9] + t[10] * t[11]
    got = sp.map(exprs, many).optimized().glom()
    np.testing.assert_array_equal(got, many(*arrays))
    before = ctx.backend.launches
    x = sp.from_numpy(arrays[0])
    e = x
    want = arrays[0].copy()
    for i in range(1, 50):             # 98 operators in one fused tree, values stay small integers
      e = (e + (i % 7)) - (i % 3)
      want = (want + np.float32(i % 7)) - np.float32(i % 3)
    got = e.optimized().glom()
    np.testing.assert_array_equal(got, want)
    assert ctx.backend.launches - before > 2 * 2        # more than one launch per tile
  finally:
    sp.shutdown()


@pytest.mark.parametrize('shape', [(3001, 3001), (2048, 4100), (2896, 2896)])
def test_streaming_policy_does_not_change_results(shape):
  """Above 8 M elements a launch reads operands it touches once, and writes map outputs, non-temporally
  (SP_PAD_STREAM): integer-valued data, so every result is exact -- a ragged inner dimension, an aligned one,
  row / column broadcasts (which keep the default policy), reductions along both axes and argmax."""
  sp.initialize('hip', num_workers=1)
  try:
    rows, cols = shape
    rng = np.random.RandomState(11)
    x = rng.randint(-8, 9, size=shape).astype(np.float32)
    row = rng.randint(-3, 4, size=(cols,)).astype(np.float32)
    col = rng.randint(-3, 4, size=(rows, 1)).astype(np.float32)
    X, R, C = sp.from_numpy(x), sp.from_numpy(row), sp.from_numpy(col)
    np.testing.assert_array_equal((X * R + C - X).optimized().glom(), x * row + col - x)
    np.testing.assert_array_equal((X * X + X).optimized().glom(), x * x + x)
    np.testing.assert_array_equal(sp.sum(X * R, axis=0).optimized().glom(), (x * row).sum(axis=0))
    np.testing.assert_array_equal(sp.sum(X + C, axis=1).optimized().glom(), (x + col).sum(axis=1))
    np.testing.assert_array_equal(sp.sum(X).glom(), x.sum(dtype=np.float64).astype(np.float32))
    np.testing.assert_array_equal(sp.argmax(X, axis=1).glom(), np.argmax(x, axis=1))
    np.testing.assert_array_equal(sp.argmin(X, axis=0).glom(), np.argmin(x, axis=0))
  finally:
    sp.shutdown()


def test_driver_arrays_are_uploaded_again_when_their_bytes_change():
  """`dot(x, w)` with a driver-side NumPy `w` (dot.py:172-187: the reference pickles it into every request): a new
  array object is uploaded without being hashed, an object that comes back is re-used only while its bytes are the
  same -- a driver that updates `w` in place between steps must see the new values."""
  ctx = sp.initialize('hip', num_workers=3)
  try:
    rng = np.random.RandomState(3)
    xh = rng.randint(-3, 4, size=(300, 40)).astype(np.float32)
    x = sp.from_numpy(xh)
    w = rng.randint(-3, 4, size=(40, 1)).astype(np.float32)
    np.testing.assert_array_equal(sp.dot(x, w).glom(), xh.dot(w))       # first sight: uploaded, not hashed
    np.testing.assert_array_equal(sp.dot(x, w).glom(), xh.dot(w))       # same object, same bytes
    w[5, 0] += 2.0                                                       # the driver steps its weights in place
    np.testing.assert_array_equal(sp.dot(x, w).glom(), xh.dot(w))
    w[:] = 0
    np.testing.assert_array_equal(sp.dot(x, w).glom(), np.zeros((300, 1), np.float32))
    w2 = w + 1.0                                                         # a new object every step
    np.testing.assert_array_equal(sp.dot(x, w2).glom(), xh.dot(w2))
  finally:
    sp.shutdown()
