"""Sparse-tile programs (tests/sparse_programs.py) against golden outputs recorded by RUNNING
THE REFERENCE (tests/golden/make_golden.py --sparse -> sparse_w{1,3,4,8}.npz):
  * host logic + the NumPy/scipy tile backend on CPU (every pytest run) -- this pins the oracle;
  * the HIP backend (device CSR tiles, spartan_amd/csrc/sparse.hip) with -m gpu.
All programs hold small-integer values, so results are bit-identical unless the program states a tolerance.
Programs the reference itself cannot run for a worker count (it fails slicing a COO tile,
tile.pyx:91-98) are still run here and compared with the dense NumPy evaluation of the same program.
"""
import json
import os

import numpy as np
import pytest
import scipy.sparse as sps

import spartan_amd as sp
from tests import sparse_programs

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
META = json.load(open(os.path.join(HERE, 'sparse_meta.json')))
PROGS = sparse_programs.programs()


def _run(backend_factory, workers, name, build):
  sp.initialize(backend=backend_factory(), num_workers=workers)
  try:
    res = build(sp)
    res = res.force() if hasattr(res, 'force') else res
    return sparse_programs.to_dense(res.glom())
  finally:
    sp.shutdown()


def _check(backend_factory, workers):
  gold = np.load(os.path.join(HERE, 'sparse_w%d.npz' % workers))
  gold1 = np.load(os.path.join(HERE, 'sparse_w1.npz'))
  meta = META[str(workers)]
  pinned = 0
  for name, build, tol in PROGS:
    got, was_sparse = _run(backend_factory, workers, name, build)
    m = meta[name]
    if 'skipped' in m:
      if name not in gold1.files:
        continue
      want, want_meta = gold1[name], META['1'][name]      # the value does not depend on the tiling
    else:
      want, want_meta = gold[name], m
      pinned += 1
    assert list(got.shape) == want_meta['shape'], name
    assert was_sparse == want_meta['sparse'], '%s: sparse result %s, reference %s' % (name, was_sparse, want_meta['sparse'])
    assert got.dtype.str == want_meta['dtype'], '%s: dtype %s, reference %s' % (name, got.dtype.str, want_meta['dtype'])
    if tol is None:
      np.testing.assert_array_equal(got, want, err_msg=name)
    else:
      np.testing.assert_allclose(got, want, rtol=tol[0], atol=tol[1], err_msg=name)
  assert pinned >= 18


@pytest.mark.parametrize('workers', [1, 3, 4, 8])
def test_sparse_programs_match_reference_cpu(workers):
  from oracle.np_backend import NumpyBackend
  _check(NumpyBackend, workers)


@pytest.mark.gpu
@pytest.mark.parametrize('workers', [1, 3, 4, 8])
def test_sparse_programs_match_reference_gpu(workers):
  from spartan_amd.backend_hip import HipBackend
  _check(HipBackend, workers)


def _numpy_rhs(backend_factory):
  """dot(sparse, NumPy array): the reference's map2 numpy mapper fails on it (IndexError in both worker
  counts), so the expected value is the dense product."""
  got, was_sparse = _run(backend_factory, 1, 'links_dot_numpy',
                         lambda s: s.dot(sparse_programs.links(s, (96, 64), 9), np.arange(64 * 3, dtype=np.float32).reshape(64, 3) % 5))
  w, _ = _run(backend_factory, 1, 'links', lambda s: sparse_programs.links(s, (96, 64), 9))
  np.testing.assert_array_equal(got, w @ (np.arange(64 * 3, dtype=np.float32).reshape(64, 3) % 5))
  assert not was_sparse


def test_sparse_dot_numpy_rhs_cpu():
  from oracle.np_backend import NumpyBackend
  _numpy_rhs(NumpyBackend)


@pytest.mark.gpu
def test_sparse_dot_numpy_rhs_gpu():
  from spartan_amd.backend_hip import HipBackend
  _numpy_rhs(HipBackend)


def _dense_times_sparse(backend_factory):
  """dense x sparse (rows <= cols: the map2 join; rows > cols: the outer path), against the dense product."""
  for shape_a, shape_w in (((24, 64), (64, 48)), ((96, 32), (32, 20))):
    a = (np.arange(shape_a[0] * shape_a[1], dtype=np.float32).reshape(shape_a) % 5) - 2
    got, was_sparse = _run(backend_factory, 1, 'dense_dot_links',
                           lambda s: s.dot(s.from_numpy(a), sparse_programs.links(s, shape_w, 6)))
    w, _ = _run(backend_factory, 1, 'links', lambda s: sparse_programs.links(s, shape_w, 6))
    np.testing.assert_array_equal(got, a @ w)
    assert not was_sparse


def test_dense_times_sparse_cpu():
  from oracle.np_backend import NumpyBackend
  _dense_times_sparse(NumpyBackend)


@pytest.mark.gpu
def test_dense_times_sparse_gpu():
  from spartan_amd.backend_hip import HipBackend
  _dense_times_sparse(HipBackend)


def _sparse_rand(backend_factory):
  sp.initialize(backend=backend_factory(), num_workers=4)
  try:
    x = sp.sparse_rand((400, 300), density=0.05).force()
    g = x.glom()
    assert sps.issparse(g) and g.shape == (400, 300) and g.dtype == np.float32
    assert 0.03 * 120000 < g.nnz <= 0.05 * 120000 + 4          # colliding positions are merged
    assert g.data.min() >= 0 and g.data.max() < 1.0
    e = sp.sparse_empty((30, 20)).force().glom()
    assert sps.issparse(e) and e.nnz == 0 and e.shape == (30, 20)   # (float64: tile.pyx:77, see Tile.get)
    total = float(sp.sum(sp.Val(val=x)).glom())
    np.testing.assert_allclose(total, g.data.astype(np.float64).sum(), rtol=1e-5)
  finally:
    sp.shutdown()


def test_sparse_rand_cpu():
  from oracle.np_backend import NumpyBackend
  _sparse_rand(NumpyBackend)


@pytest.mark.gpu
def test_sparse_rand_gpu():
  from spartan_amd.backend_hip import HipBackend
  _sparse_rand(HipBackend)


def _sparse_scan(backend_factory):
  """tests/test_scan.py:test_sparse_scan, with its own lambdas as reduce / scan functions."""
  sp.initialize(backend=backend_factory(), num_workers=4)
  try:
    eye = np.eye(10)
    s = sp.sparse_diagonal((10, 10), np.float32, (5, 5))
    red = lambda x, **kw: x.sum(axis=kw['axis'])          # noqa: E731
    scn = lambda x, **kw: x.cumsum(axis=kw['axis'])       # noqa: E731
    np.testing.assert_array_equal(sp.scan(s, reduce_fn=red, scan_fn=scn, axis=None).glom(), np.cumsum(eye).reshape(10, 10))
    for axis in (0, 1):
      np.testing.assert_array_equal(sp.scan(s, reduce_fn=red, scan_fn=scn, axis=axis).glom(), np.cumsum(eye, axis))
  finally:
    sp.shutdown()


def test_sparse_scan_cpu():
  from oracle.np_backend import NumpyBackend
  _sparse_scan(NumpyBackend)


@pytest.mark.gpu
def test_sparse_scan_gpu():
  from spartan_amd.backend_hip import HipBackend
  _sparse_scan(HipBackend)


def _framework_fuzz(backend_factory, n_cases):
  """Random shapes, tilings and worker counts through the expression API; integer values, expectations from the
  dense NumPy evaluation of the same program."""
  rng = np.random.RandomState(99)
  for case in range(n_cases):
    workers = int(rng.choice([1, 2, 4]))
    m, k, n = [int(v) for v in rng.choice([8, 12, 20, 36, 64], size=3)]

    def hint(shape):
      kind = rng.randint(3)
      if kind == 0:
        return None
      if kind == 1:
        return (max(1, shape[0] // workers), shape[1])
      return (shape[0], max(1, shape[1] // workers))
    ha, hb = hint((m, k)), hint((k, n))
    da = np.asarray(sparse_programs.link_block((0, 0), (m, k), 100 + case).todense())
    db = np.asarray(sparse_programs.link_block((0, 0), (k, n), 200 + case).todense())
    x = (np.arange(k * 3, dtype=np.float32).reshape(k, 3) % 5) - 2
    sp.initialize(backend=backend_factory(), num_workers=workers)
    try:
      A = sparse_programs.links(sp, (m, k), 100 + case, ha)
      B = sparse_programs.links(sp, (k, n), 200 + case, hb)
      tag = 'case %d: %s x %s, %d workers, hints %s %s' % (case, (m, k), (k, n), workers, ha, hb)
      checks = [
          (lambda: sp.dot(A, sp.from_numpy(x)), da @ x),
          (lambda: sp.sum(A, axis=0), da.sum(0)),
          (lambda: sp.sum(A, axis=1), da.sum(1)),
          (lambda: sp.add(A, A), da + da),
          (lambda: sp.transpose(A), da.T),
          (lambda: sp.add(A, sp.ones((m, k))), da + 1),
      ]
      checks.append((lambda: sp.dot(A, B), da @ db))       # (any tiling: sparse tiles are sliceable here)
      for build, want in checks:
        got, _ = sparse_programs.to_dense(build().force().glom() if hasattr(build(), 'force') else build().glom())
        np.testing.assert_array_equal(got, want, err_msg=tag)
    finally:
      sp.shutdown()


def test_sparse_framework_fuzz_cpu():
  from oracle.np_backend import NumpyBackend
  _framework_fuzz(NumpyBackend, 30)


@pytest.mark.gpu
def test_sparse_framework_fuzz_gpu():
  from spartan_amd.backend_hip import HipBackend
  _framework_fuzz(HipBackend, 30)
