"""write / assign / region_map / retile / norm_cdf (the loaders next to the map-reduce path, SURVEY 8f.1: reference
write_array.py, assign.py, region_map.py, retile.py, statistics.py:224-225) against NumPy on the same inputs; integer-valued data, so results are bit-exact.  CPU leg on the oracle backend, GPU leg on the HIP kernels."""
import numpy as np
import pytest

import spartan_amd as sp


def _check_all():
  a = (np.arange(40 * 6, dtype=np.float32).reshape(40, 6) % 17)
  A = sp.from_numpy(a).evaluate()
  v = -np.arange(20 * 3, dtype=np.float32).reshape(20, 3)
  V = sp.from_numpy(v).evaluate()
  # assign: scalar / host array / distributed value; the source array is NOT modified
  want = a.copy()
  want[5:25, 1:4] = 7.5
  np.testing.assert_array_equal(sp.assign(A, np.index_exp[5:25, 1:4], 7.5).glom(), want)
  want[5:25, 1:4] = v
  np.testing.assert_array_equal(sp.assign(A, np.index_exp[5:25, 1:4], v).glom(), want)
  np.testing.assert_array_equal(sp.assign(A, np.index_exp[5:25, 1:4], V).glom(), want)
  np.testing.assert_array_equal(sp.assign(A, 3, 1.0).glom()[3], np.ones(6, np.float32))
  row = np.arange(6, dtype=np.float32) - 2                     # a value with fewer axes than the box
  np.testing.assert_array_equal(sp.assign(A, np.index_exp[10, ], row).glom()[10], row)
  np.testing.assert_array_equal(A.glom(), a)
  # region_map: the function sees a backend view of the cells a box shares with the tile
  seen = []

  def minus_one(view, ex):
    seen.append(tuple(view.shape))
    return -1.0
  got = sp.region_map(A, sp.extent.from_slice(np.index_exp[0:10, 0:6], a.shape), minus_one).glom()
  want2 = a.copy()
  want2[0:10] = -1
  np.testing.assert_array_equal(got, want2)
  assert seen and all(s[1] == 6 for s in seen)
  got = sp.region_map(A, [sp.extent.from_slice(np.index_exp[38:40, 2:4], a.shape)], lambda view, ex, k: view * k, fn_kw={'k': 2}).glom()
  want2 = a.copy()
  want2[38:40, 2:4] *= 2
  np.testing.assert_array_equal(got, want2)
  # retile keeps the values, changes the tiles
  R = sp.retile(A, (40, 2)).evaluate()
  np.testing.assert_array_equal(R.glom(), a)
  assert all(ex.shape[1] <= 2 for ex in R.tiles)
  # write mutates in place (write_array.py:1-9), from host data and from another distributed array
  Bm = sp.from_numpy(a.copy()).evaluate()
  sp.write(Bm, np.index_exp[0:10, 0:6], np.full((10, 6), 3, np.float32), np.index_exp[0:10, 0:6]).evaluate()
  w = a.copy()
  w[0:10] = 3
  np.testing.assert_array_equal(Bm.glom(), w)
  sp.write(Bm, np.index_exp[30:40, 0:3], V, np.index_exp[5:15, 0:3]).evaluate()
  w[30:40, 0:3] = v[5:15]
  np.testing.assert_array_equal(Bm.glom(), w)
  # norm_cdf (statistics.py:224-225): scipy.stats.norm.cdf per tile; float64 in, float64 out
  import scipy.stats
  z = (np.arange(41 * 7, dtype=np.float64).reshape(41, 7) - 140) / 23.0
  got = sp.norm_cdf(sp.from_numpy(z)).glom()
  assert got.dtype == np.float64
  np.testing.assert_allclose(got, scipy.stats.norm.cdf(z), rtol=1e-13, atol=1e-300)
  got32 = sp.norm_cdf(sp.from_numpy(z.astype(np.float32)) * 2).optimized().glom()
  np.testing.assert_allclose(got32, scipy.stats.norm.cdf(z.astype(np.float32) * 2), rtol=1e-6)


@pytest.mark.parametrize('workers', [1, 4])
def test_region_ops_host_framework(workers):
  from oracle.np_backend import NumpyBackend
  sp.initialize(backend=NumpyBackend(), num_workers=workers)
  try:
    _check_all()
  finally:
    sp.shutdown()


@pytest.mark.gpu
@pytest.mark.parametrize('workers', [1, 3])
def test_region_ops_hip(workers):
  ctx = sp.initialize('hip', num_workers=workers)
  try:
    before = ctx.backend.launches
    _check_all()
    assert ctx.backend.launches > before
  finally:
    sp.shutdown()
