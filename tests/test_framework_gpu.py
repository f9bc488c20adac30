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
  got = build(sp).glom()
  programs.check(name, got, expected(), tol)


def test_hip_backend_is_the_one_running(ctx):
  import torch
  from spartan_amd import _hip
  assert ctx.backend.name == 'hip'
  before = ctx.backend.launches
  r = (sp.ones((256, 256)) + 1).force()
  assert ctx.backend.launches > before
  t = ctx.tile(list(r.tiles.values())[0]).data
  assert isinstance(t, torch.Tensor) and t.is_cuda
  assert _hip.lib().sp_abi_version() == 1


def test_user_callables_are_traced_or_refused_loudly(ctx):
  """An element-wise Python function becomes part of the kernel (traced); anything else is refused --
  there is no CPU fallback to run it on."""
  from spartan_amd.lower import NotLowerable
  np.testing.assert_array_equal(sp.map(sp.ones((8, 8)), fn=lambda x: x * 3 + 1).glom(), np.full((8, 8), 4, np.float32))
  for bad in (lambda x: np.cumsum(x), lambda x: x[::2], lambda x: x.sum(), lambda x: x + 1 if x > 0 else x):
    with pytest.raises(NotLowerable):
      sp.map(sp.ones((8, 8)), fn=bad).force()


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
