"""PageRank-style sparse multiply: the program of the reference's tests/benchmark_pagerank.py and
tests/test_pagerank.py (the reference ships it with its benchmarks, not under spartan/examples).

  wts = pagerank_sparse(num_pages, num_outlinks, same_site_prob)   # sparse [pages x pages], built tile by tile
  p   = dot(wts, p)  (num_iter times)                              # CSR x dense vector on the GPU: sp_csr_spmm

The link structure is drawn on the host with NumPy, exactly as the reference's mapper does
(benchmark_pagerank.py:57-76; its `_build_site_coo` loop is a plain repeat of the page index), and handed to the
framework as a scipy COO block; the framework uploads it once and keeps it as a device CSR tile.
"""
import numpy as np
import scipy.sparse

from .. import expr
from ..array import extent


def _make_site_sparse(tile, ex, num_outlinks=None, same_site_prob=None):
  """benchmark_pagerank.py:57-76: `num_outlinks` links per page of this tile's page range; a link stays inside
  the tile's own range ("site") with probability same_site_prob."""
  if ex.shape[0] == tile.shape[0]:
    tile_pages = ex.shape[1]
    ul, lr = ex.ul[1], ex.lr[1]
  else:
    tile_pages = ex.shape[0]
    ul, lr = ex.ul[0], ex.lr[0]
  n = num_outlinks * tile_pages
  same_site = np.random.rand(n) <= same_site_prob
  outlink = np.zeros(n, dtype=np.int32)
  outlink[same_site] = np.random.randint(ul, lr, np.count_nonzero(same_site))
  outlink[~same_site] = np.random.randint(0, tile.shape[0], np.count_nonzero(~same_site))
  cols = np.repeat(np.arange(tile_pages, dtype=np.int32), num_outlinks)      # _build_site_coo (:37-54)
  data = np.ones(n, dtype=np.float32)
  result = scipy.sparse.coo_matrix((data, (outlink, cols)), shape=(tile.shape[0], tile_pages), dtype=np.float32)
  result_ex = extent.create((0, ul), (tile.shape[0], lr), tile.shape)
  yield result_ex, result


def pagerank_sparse(num_pages, num_outlinks, same_site_prob, tile_hint=None):
  """benchmark_pagerank.py:104-116.  `tile_hint` (not in the reference's signature) lets the caller ask for the
  column tiling `[num_pages, pages_per_worker]` the benchmark's comments describe; with the default (row) tiling
  the mapper's extents are relative to the TILE shape, as in the reference."""
  result = expr.ndarray((num_pages, num_pages), dtype=np.float32, sparse=True, tile_hint=tile_hint)
  cost = num_pages * num_pages
  return expr.shuffle(result, target=result, fn=_make_site_sparse,
                      kw={'num_outlinks': num_outlinks, 'same_site_prob': same_site_prob},
                      cost_hint={hash(result): {'11': 0, '01': cost, '10': cost, '00': cost}})


def sparse_multiply(wts, p, num_iter=5):
  """benchmark_pagerank.py:22-27: p <- wts . p, num_iter times."""
  for _ in range(num_iter):
    p = expr.dot(wts, p).optimized()
  return p.evaluate()
