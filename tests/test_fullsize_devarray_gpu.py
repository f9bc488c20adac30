"""Parity at BASELINE.json's FULL sizes WITHOUT torch: the same size-independent properties as
tests/test_fullsize_gpu.py (planted extremes with duplicates, sums of sums, integer-exact closed forms, spot rows and
columns against float64), with the product's own device arrays for planting and reading and NumPy on the host as the
independent calculator -- so these tests run (and cannot silently skip) on a box that has the library and no torch.

  configs[1]  dot 8192^3 fp32: integer-valued closed form bit-exact; uniform operands, spot rows vs float64
  configs[2]  the 8192 x 65536 tile and the whole 65536^2 array as 8 tiles: reductions and arg-reductions
  configs[3]  one Lloyd iteration on the 1 250 000 x 256 tile, k = 1024
  configs[4]  one gradient step on the 125 000 x 4096 tile
"""
import numpy as np
import pytest

import spartan_amd as sp
from spartan_amd import devarray as D
from tests.dev import uniform

pytestmark = pytest.mark.gpu


@pytest.fixture
def ctx():
  c = sp.initialize('hip')
  yield c
  sp.shutdown()
  D.trim_pool()


@pytest.fixture
def ctx8():
  D.trim_pool()
  c = sp.initialize('hip', num_workers=8)
  yield c
  sp.shutdown()
  D.trim_pool()


def _tile(ctx, array, row=0):
  """(device array of the tile holding `row`, its extent)."""
  for ex, tid in array.tiles.items():
    if ex.ul[0] <= row < ex.lr[0]:
      return ctx.tile(tid).data, ex
  raise KeyError(row)


def _plant(ctx, array, r, c, v):
  t, ex = _tile(ctx, array, r)
  t[r - ex.ul[0], c] = v                     # DevArray.__setitem__: a one-element box fill


def test_config3_tile_and_full_array_reductions(ctx8):
  R = C = 65536
  X = uniform(sp, (R, C), 141).force()
  assert sorted(ex.shape for ex in X.tiles) == [(8192, C)] * 8
  for r, c, v in [(60000, 11, 9.0), (100, 11, 9.0), (20000, 40000, 9.0),       # max 9.0: first at (100, 11)
                  (50000, 7, -3.0), (9000, 7, -3.0), (9000, 65535, -3.0)]:     # min -3.0: first at (9000, 7)
    _plant(ctx8, X, r, c, v)
  Xv = sp.Val(val=X)
  n = float(R) * C
  total = float(sp.sum(Xv).glom())
  by_col = sp.sum(Xv, 0).glom().astype(np.float64)
  by_row = sp.sum(Xv, 1).glom().astype(np.float64)
  assert abs(by_col.sum() - total) <= 1e-6 * n and abs(by_row.sum() - total) <= 1e-6 * n
  assert abs(total - 0.5 * n) < 1e-3 * n
  assert int(sp.argmax(Xv).glom()) == 100 * C + 11              # duplicates across tiles: first in row-major order
  assert int(sp.argmin(Xv).glom()) == 9000 * C + 7
  am0 = sp.argmax(Xv, 0).glom()
  assert am0[11] == 100 and am0[40000] == 20000
  an0 = sp.argmin(Xv, 0).glom()
  assert an0[7] == 9000 and an0[65535] == 9000
  am1 = sp.argmax(Xv, 1).glom()
  assert am1[60000] == 11 and am1[100] == 11 and am1[20000] == 40000
  assert float(sp.max(Xv).glom()) == 9.0 and float(sp.min(Xv).glom()) == -3.0
  # spot rows and columns against float64 NumPy (rows / strided columns read back through the product's box copy)
  rows = [0, 100, 9000, 65535]
  for r in rows:
    t, ex = _tile(ctx8, X, r)
    np.testing.assert_allclose(by_row[r], t[r - ex.ul[0]].numpy().astype(np.float64).sum(), rtol=1e-6)
  cols = [0, 11, 40000, 65535]
  ref = np.zeros(len(cols))
  for ex, tid in X.tiles.items():
    t = ctx8.tile(tid).data
    ref += np.stack([t[:, c].numpy().astype(np.float64).sum() for c in cols])
  np.testing.assert_allclose(by_col[cols], ref, rtol=2e-6)
  # the fused map on one whole tile, bit-exact against the same fp32 operations in NumPy on a sampled band
  y = (Xv * Xv + Xv).optimized().force()
  t, ex = _tile(ctx8, X, 30000)
  ty, _ = _tile(ctx8, y, 30000)
  band = slice(30000 - ex.ul[0], 30000 - ex.ul[0] + 64)
  xs = t[band].numpy()
  np.testing.assert_array_equal(ty[band].numpy(), xs * xs + xs)


def test_config2_dot_8192(ctx):
  n = 8192
  ii = np.arange(n, dtype=np.float32)
  a = ((ii[:, None] + ii[None, :]) % 3) - 1                      # integer-valued: the product is exact in fp32
  A = sp.from_numpy(a)
  Cd = sp.dot(A, sp.ones((n, n))).force()
  c, _ = _tile(ctx, Cd)
  np.testing.assert_array_equal(c.numpy(), np.broadcast_to(a.sum(1, keepdims=True), (n, n)))
  U = uniform(sp, (n, n), 111, -1.0, 1.0).force()
  V = uniform(sp, (n, n), 112, -1.0, 1.0).force()
  W = sp.dot(sp.Val(val=U), sp.Val(val=V)).force()
  u, _ = _tile(ctx, U)
  v, _ = _tile(ctx, V)
  w, _ = _tile(ctx, W)
  rows = [0, 1, 4095, 8191]
  v64 = v.numpy().astype(np.float64)
  for r in rows:
    ref = u[r].numpy().astype(np.float64) @ v64
    assert np.abs(w[r].numpy().astype(np.float64) - ref).max() <= 2 * n * np.finfo(np.float32).eps


def test_config4_kmeans_iteration(ctx):
  from scipy.spatial.distance import cdist
  from spartan_amd.examples.sklearn.cluster import KMeans
  n, k, d = 1250000, 1024, 256
  X = uniform(sp, (n, d), 131).force()
  x, _ = _tile(ctx, X)
  rng = np.random.RandomState(7)
  centers = rng.rand(k, d)
  centers[700] = centers[3]                                      # exact duplicate: index 3 must win ties
  plant = rng.randint(0, n, size=64)
  for j, r in enumerate(plant):
    x[int(r)] = centers[j].astype(np.float32)                    # a point ON centre j
  new_centers, labels = KMeans(k, 1).fit(sp.Val(val=X), centers.copy(), implementation='map2', reducer=np.add)
  lab = labels.glom()
  assert lab.shape == (n,) and lab.dtype == np.float32           # map2 targets take the points' dtype (map.py:317-318)
  lab = lab.astype(np.int64)
  host = x.numpy()                                               # 1.28 GB: the independent calculator is NumPy
  sample = np.concatenate([plant, rng.randint(0, n, size=4000)])
  np.testing.assert_array_equal(lab[sample], np.argmin(cdist(host[sample], centers), axis=1))
  assert not np.any(lab == 700)
  counts = np.bincount(lab, minlength=k)
  assert counts.sum() == n and counts[700] == 0
  col = host.sum(axis=0, dtype=np.float64)
  np.testing.assert_allclose((new_centers * counts[:, None]).sum(axis=0), col, rtol=2e-6)
  # the builder the reference's users call, on the same labels (statistics.py:115-137 asserts min > 0: shifted)
  shifted = sp.astype(labels, np.int64) + 1
  np.testing.assert_array_equal(sp.bincount(shifted).glom()[1:], counts)


def test_config5_lreg_step(ctx):
  N, Dm = 125000, 4096
  X = uniform(sp, (N, Dm), 121).force()
  y = uniform(sp, (N, 1), 122).force()
  w = np.random.RandomState(1).rand(Dm, 1).astype(np.float32)
  Xv, yv = sp.Val(val=X), sp.Val(val=y)
  yp = sp.dot(Xv, w)
  grad = sp.sum(Xv * (yp - yv), axis=0).optimized().glom()
  x, _ = _tile(ctx, X)
  yt, _ = _tile(ctx, y)
  host = x.numpy()                                               # 2 GB
  r = np.zeros((N, 1))
  for lo in range(0, N, 25000):                                  # float64 in blocks: X.w - y
    r[lo:lo + 25000] = host[lo:lo + 25000].astype(np.float64) @ w.astype(np.float64)
  r -= yt.numpy().astype(np.float64)
  cols = [0, 1, 2047, 4095]
  ref = (host[:, cols].astype(np.float64) * r).sum(0)
  np.testing.assert_allclose(grad[cols], ref, rtol=2e-5)
  np.testing.assert_allclose(yp.glom()[:5, 0], (host[:5].astype(np.float64) @ w.astype(np.float64))[:, 0], rtol=2e-6)
