"""Sparse-tile test programs, written against the builder API the reference and spartan_amd share.
They follow the reference's own sparse tests (tests/test_sparse.py, test_transpose.py:19-25, test_pagerank.py /
benchmark_pagerank.py) and are run (a) by tests/golden/make_golden.py --sparse through the REFERENCE to record
golden outputs, (b) through the NumPy/scipy tile backend on CPU and (c) through the HIP backend.

Entry: (name, build(sp) -> Expr or value, tol); the result is compared as a dense array (`.todense()` when the
program yields a sparse array) and `sparse` (whether glom() returned a scipy matrix) must match as well.
"""
import numpy as np
import scipy.sparse as sps

F32 = np.float32
SUM_TOL = (1e-6, 1e-5)


def link_block(ul, lr, seed):
  """Entries of a fixed pseudo-random integer matrix inside the box [ul, lr): cell (r, c) is stored when
  h(r, c, seed) % 7 == 0, with value 1 + (h // 7) % 4 -- a function of the GLOBAL coordinates, so the
  matrix does not depend on how the array is tiled."""
  r = np.arange(ul[0], lr[0], dtype=np.int64)[:, None]
  c = np.arange(ul[1], lr[1], dtype=np.int64)[None, :]
  h = (r * 2654435761 + c * 40503 + (seed + 1) * 977) % 1000003
  rows, cols = np.nonzero(h % 7 == 0)
  data = (1 + (h[rows, cols] // 7) % 4).astype(F32)
  return sps.coo_matrix((data, (rows.astype(np.int32), cols.astype(np.int32))),
                        shape=(lr[0] - ul[0], lr[1] - ul[1]), dtype=F32)


def _make_links(tile, ex, seed=None):
  """Shuffle mapper in the style of benchmark_pagerank.py:_make_site_sparse: one sparse block per tile."""
  yield ex, link_block(ex.ul, ex.lr, seed)


def links(sp, shape, seed, tile_hint=None):
  target = sp.ndarray(shape, dtype=F32, sparse=True, tile_hint=tile_hint)
  return sp.shuffle(target, _make_links, target=target, kw={'seed': seed})


def _pagerank(sp, n, iters, tile_hint=None):
  w = links(sp, (n, n), 11, tile_hint)
  p = sp.from_numpy(np.linspace(0.5, 1.5, n, dtype=F32).reshape(n, 1))
  for _ in range(iters):
    p = sp.dot(w, p)
  return p


def programs():
  P = []
  add = P.append
  # ---- tests/test_sparse.py
  add(('diag_glom', lambda sp: sp.sparse_diagonal((10, 10)), None))
  add(('diag_wide', lambda sp: sp.sparse_diagonal((107, 401)), None))
  add(('diag_tall_tiled', lambda sp: sp.sparse_diagonal((401, 107), tile_hint=(100, 107)), None))
  add(('diag_sum_all', lambda sp: sp.sum(sp.sparse_diagonal((10, 10))), None))
  add(('diag_sum_axis0', lambda sp: sp.sum(sp.sparse_diagonal((40, 30)), axis=0), None))
  add(('diag_sum_axis1', lambda sp: sp.sum(sp.sparse_diagonal((40, 30)), axis=1), None))
  add(('diag_add', lambda sp: sp.add(sp.sparse_diagonal((10, 10)), sp.sparse_diagonal((10, 10))), None))
  add(('diag_sub', lambda sp: sp.sub(sp.sparse_diagonal((10, 10)), sp.sparse_diagonal((10, 10))), None))
  add(('diag_dot_diag', lambda sp: sp.dot(sp.sparse_diagonal((10, 10)), sp.sparse_diagonal((10, 10))), None))
  # ---- sparse arrays built by a shuffle mapper (benchmark_pagerank.py:pagerank_sparse)
  add(('links_glom', lambda sp: links(sp, (60, 50), 3), None))
  add(('links_sum_axis0', lambda sp: sp.sum(links(sp, (60, 50), 3), axis=0), None))
  add(('links_sum_axis1', lambda sp: sp.sum(links(sp, (60, 50), 3), axis=1), None))
  add(('links_add', lambda sp: sp.add(links(sp, (60, 50), 3), links(sp, (60, 50), 4)), None))
  add(('links_sub', lambda sp: sp.sub(links(sp, (60, 50), 3), links(sp, (60, 50), 4)), None))
  add(('links_dot_links', lambda sp: sp.dot(links(sp, (40, 64), 5), links(sp, (64, 48), 6)), None))
  add(('links_plus_dense', lambda sp: sp.add(links(sp, (24, 16), 7), sp.ones((24, 16))), None))
  # ---- sparse x dense (tests/test_pagerank.py: p = dot(wts, p), iterated)
  add(('links_dot_vec', lambda sp: sp.dot(links(sp, (96, 96), 8), sp.from_numpy(np.arange(96, dtype=F32).reshape(96, 1))), None))
  add(('links_dot_mat', lambda sp: sp.dot(links(sp, (96, 64), 9), sp.from_numpy(np.arange(64 * 5, dtype=F32).reshape(64, 5) % 7)), None))

  add(('pagerank_3_iters', lambda sp: _pagerank(sp, 120, 3), (1e-5, 0)))
  add(('pagerank_col_tiles', lambda sp: _pagerank(sp, 120, 2, tile_hint=(120, 30)), (1e-5, 0)))
  # ---- tests/test_transpose.py:test_transpose3
  add(('diag_transpose', lambda sp: sp.transpose(sp.sparse_diagonal((107, 401))), None))
  add(('links_transpose', lambda sp: sp.transpose(links(sp, (60, 50), 3)), None))
  # ---- tests/test_reshape.py:test_reshape8
  add(('diag_reshape', lambda sp: sp.reshape(sp.sparse_diagonal((137, 113)), (113, 137)), None))
  add(('links_reshape', lambda sp: sp.reshape(links(sp, (60, 50), 3), (100, 30)), None))
  return P


def to_dense(v):
  """glom() value -> (dense ndarray, was_sparse)."""
  if sps.issparse(v):
    return np.asarray(v.todense()), True
  return np.asarray(v), False
