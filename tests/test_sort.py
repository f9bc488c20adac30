"""sort / argsort / partition (spartan/expr/operator/sort.py; reference tests/test_sort.py) on the NumPy oracle
backend (CPU) and on the HIP backend, plus sp_sort_rows through the C-ABI against np.sort / np.argsort
(kind='stable').  Sorting moves values, it does not compute them: every comparison is bit-exact."""
import numpy as np
import pytest

import spartan_amd as sp


def _int_array(rng, shape):
  """tests/test_sort.py:new_ndarray: int(randn * 100) -- plenty of ties."""
  return (rng.randn(*shape) * 100).astype(np.int64)


def _reference_suite(backend_factory, workers, seed):
  """tests/test_sort.py:test_ndimension.  Ties: NumPy's default argsort does not define their order (the reference's
  own test passes because both sides call the same np.argsort on the same line); ours is the stable order."""
  rng = np.random.RandomState(seed)
  sp.initialize(backend=backend_factory(), num_workers=workers)
  try:
    for case in range(3):
      dim = rng.randint(2, 6 if workers == 1 else 5)   # sub-tile updates of > 4-d tiles: SP_MAX_DIMS (DESIGN (d))
      shape = tuple(int(v) for v in rng.randint(5, 11, size=dim))
      na = _int_array(rng, shape)
      a = sp.from_numpy(na)
      for axis in range(dim):
        np.testing.assert_array_equal(sp.sort(a, axis).glom(), np.sort(na, axis))
        np.testing.assert_array_equal(sp.argsort(a, axis).glom(), np.argsort(na, axis, kind='stable'))
    # floats: NaN last, signed zeros equal, infinities; argsort of a float array has the input's dtype (map.py:317-318)
    f = rng.randn(40, 23).astype(np.float32)
    f[3, 4] = np.nan
    f[3, 9] = np.nan
    f[5, 1] = -0.0
    f[5, 2] = 0.0
    f[7, 0] = np.inf
    f[7, 5] = -np.inf
    fa = sp.from_numpy(f)
    for axis in (0, 1, -1):
      got = sp.sort(fa, axis).glom()
      np.testing.assert_array_equal(got.view(np.int32), np.sort(f, axis, kind='stable').view(np.int32))
      gi = sp.argsort(fa, axis).glom()
      assert gi.dtype == np.float32
      np.testing.assert_array_equal(gi, np.argsort(f, axis, kind='stable').astype(np.float32))
    # partition: the kth element is in its sorted place, smaller before, larger after
    p = sp.partition(sp.from_numpy(na), 2, axis=0).glom()
    srt = np.sort(na, 0)
    np.testing.assert_array_equal(p[2], srt[2])
    assert np.all(p[:2] <= p[2]) and np.all(p[3:] >= p[2])
    # long lines (the radix path on the GPU)
    g = rng.randint(-1000, 1000, size=(8, 9000)).astype(np.int32)
    np.testing.assert_array_equal(sp.sort(sp.from_numpy(g), 1).glom(), np.sort(g, 1))
    np.testing.assert_array_equal(sp.argsort(sp.from_numpy(g), 1).glom(), np.argsort(g, 1, kind='stable'))
    d = rng.randn(8, 5000)
    np.testing.assert_array_equal(sp.sort(sp.from_numpy(d), 1).glom(), np.sort(d, 1))
  finally:
    sp.shutdown()


def _flat_sort(backend_factory, workers):
  """sort(axis=None): the sample sort (benchmark_sort in tests/test_sort.py)."""
  rng = np.random.RandomState(workers)
  sp.initialize(backend=backend_factory(), num_workers=workers)
  try:
    for shape, dtype in (((10, 10, 10), np.float64), ((257, 33), np.float32), ((4000,), np.int64), ((7, 3), np.float32)):
      x = (rng.randn(*shape) * 50).astype(dtype)
      if dtype == np.float32:
        x.flat[::7] = x.flat[0]            # ties across tiles
      t = sp.sort(sp.from_numpy(x), axis=None).force()
      got = t.glom()
      assert got.shape == (x.size,) and got.dtype == x.dtype
      np.testing.assert_array_equal(got, np.sort(x, axis=None))
  finally:
    sp.shutdown()


@pytest.mark.parametrize('workers', [1, 3, 4])
def test_sort_reference_suite_cpu(workers):
  from oracle.np_backend import NumpyBackend
  _reference_suite(NumpyBackend, workers, 100 + workers)


@pytest.mark.parametrize('workers', [1, 4])
def test_flat_sort_cpu(workers):
  from oracle.np_backend import NumpyBackend
  _flat_sort(NumpyBackend, workers)


@pytest.mark.gpu
@pytest.mark.parametrize('workers', [1, 3, 4])
def test_sort_reference_suite_gpu(workers):
  from spartan_amd.backend_hip import HipBackend
  _reference_suite(HipBackend, workers, 100 + workers)


@pytest.mark.gpu
@pytest.mark.parametrize('workers', [1, 4])
def test_flat_sort_gpu(workers):
  from spartan_amd.backend_hip import HipBackend
  _flat_sort(HipBackend, workers)


# ------------------------------------------------------------------ sp_sort_rows through the C-ABI
def _special(x, rng):
  if x.dtype.kind == 'f' and x.size > 20:
    flat = x.reshape(-1)
    pos = rng.choice(x.size, size=min(12, x.size // 2), replace=False)
    flat[pos[:3]] = np.nan
    flat[pos[3:5]] = -0.0
    flat[pos[5:7]] = 0.0
    flat[pos[7]] = np.inf
    flat[pos[8]] = -np.inf
    flat[pos[9:12]] = flat[pos[9]]
  return x


@pytest.mark.gpu
@pytest.mark.parametrize('algo', ['default', 'radix'])
@pytest.mark.parametrize('dtype', [np.float32, np.float64, np.int32, np.int64])
@pytest.mark.parametrize('shape', [(1, 1), (5, 7), (1, 4096), (3, 4097), (1000, 33), (64, 64), (2, 100000), (70000, 3), (500, 300), (40, 2048)])
def test_sort_rows_kernel(monkeypatch, shape, dtype, algo):
  from spartan_amd import devarray as D
  from spartan_amd import kernels
  if algo == 'radix':
    monkeypatch.setenv('SP_SORT_ALGO', 'radix')
  rng = np.random.RandomState(shape[0] * 7 + shape[1])
  if np.dtype(dtype).kind == 'f':
    x = _special((rng.randn(*shape) * 10).astype(dtype), rng)
  else:
    x = rng.randint(-50, 50, size=shape).astype(dtype)          # many ties
    x.flat[0] = np.iinfo(dtype).min
    x.flat[-1] = np.iinfo(dtype).max
  t = D.from_numpy(x)
  vals, idx = kernels.sort_rows(t, values=True, indices=True)
  iview = {4: np.int32, 8: np.int64}[np.dtype(dtype).itemsize]
  np.testing.assert_array_equal(idx.numpy(), np.argsort(x, 1, kind='stable'))
  np.testing.assert_array_equal(vals.numpy().view(iview), np.sort(x, 1, kind='stable').view(iview))
  only_idx = kernels.sort_rows(t, values=False, indices=True)[1]
  assert np.array_equal(only_idx.numpy(), idx.numpy())


@pytest.mark.gpu
def test_sort_full_size_tile_properties():
  """configs[2] tile, 8192 x 65536 fp32 (537 M elements: the radix path, 4 key passes + 2 row passes): every row
  non-decreasing, the multiset of bit patterns of every row unchanged (XOR and wrapping sum of the int32 views),
  argsort a permutation that reproduces the sorted values."""
  torch = pytest.importorskip('torch')             # an independent calculator on the device (tests/dev.py)
  from spartan_amd import devarray as D
  from spartan_amd import kernels
  from tests.dev import T, uniform_tile
  rows, cols = 8192, 65536
  xd = uniform_tile((rows, cols), 5, -0.5, 0.5)
  x = T(xd)
  x[17, 100:200] = 0.25                        # a run of ties
  vals, idx = kernels.sort_rows(xd, values=True, indices=True)
  vals, idx = T(vals), T(idx)
  assert bool(torch.all(vals[:, 1:] >= vals[:, :-1]))
  xi, vi = x.view(torch.int32), vals.view(torch.int32)
  assert torch.equal(xi.sum(dim=1), vi.sum(dim=1))
  sample = [0, 17, 4095, 8191]
  for r in sample:
    assert torch.equal(torch.gather(x[r], 0, idx[r]), vals[r])
    assert torch.equal(torch.sort(idx[r]).values, torch.arange(cols, device='cuda'))
    ref = torch.sort(x[r], stable=True)
    assert torch.equal(ref.values, vals[r]) and torch.equal(ref.indices, idx[r])
  del vals, idx, x, xd
  D.trim_pool()


def _sort_fuzz(backend_factory, n_cases=25):
  """Random ranks (1..4), shapes, dtypes, axes and worker counts through the expression API."""
  rng = np.random.RandomState(77)
  for case in range(n_cases):
    nd = int(rng.randint(1, 5))
    workers = int(rng.choice([1, 2, 4])) if nd > 1 else 1        # (1-d arrays are split along the sort axis itself)
    shape = tuple(int(v) for v in rng.choice([4, 5, 8, 17, 64, 300], size=nd))
    if workers > 1:
      shape = tuple(max(s, workers) for s in shape)              # change_partition_axis needs >= workers slabs
    dtype = [np.float32, np.float64, np.int32, np.int64][rng.randint(4)]
    x = (rng.randn(*shape) * 20).astype(dtype) if np.dtype(dtype).kind == 'f' else rng.randint(-9, 10, size=shape).astype(dtype)
    if np.dtype(dtype).kind == 'f' and x.size > 10:
      x.flat[3] = np.nan
      x.flat[7] = x.flat[2]
    axis = int(rng.randint(-nd, nd))
    sp.initialize(backend=backend_factory(), num_workers=workers)
    try:
      a = sp.from_numpy(x)
      tag = 'case %d: %s %s axis %d, %d workers' % (case, shape, np.dtype(dtype).name, axis, workers)
      got = sp.sort(a, axis).glom()
      iview = {4: np.int32, 8: np.int64}[np.dtype(dtype).itemsize]
      np.testing.assert_array_equal(got.view(iview), np.sort(x, axis, kind='stable').view(iview), err_msg=tag)
      gi = sp.argsort(a, axis).glom()
      np.testing.assert_array_equal(gi, np.argsort(x, axis, kind='stable').astype(dtype), err_msg=tag)
    finally:
      sp.shutdown()


def test_sort_fuzz_cpu():
  from oracle.np_backend import NumpyBackend
  _sort_fuzz(NumpyBackend)


@pytest.mark.gpu
def test_sort_fuzz_gpu():
  from spartan_amd.backend_hip import HipBackend
  _sort_fuzz(HipBackend, 40)
