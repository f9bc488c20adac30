"""Sparse tile kernels: CSR x dense vector on pagerank-shaped tiles (tests/benchmark_pagerank.py:124-127:
900 000 pages per worker, 10 out-links per page, 90 % of them inside the page's site) and the COO -> CSR build."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spartan_amd import sparse as S  # noqa: E402
from tools.kbench import prewarm, timeit  # noqa: E402

DEV = 'cuda'


def site_graph(n, deg, same_site, sites, seed=0):
  """rows = link targets, cols = source pages (the layout _make_site_sparse builds)."""
  g = torch.Generator(device=DEV)
  g.manual_seed(seed)
  cols = torch.arange(n, device=DEV, dtype=torch.int64).repeat_interleave(deg)
  site = cols // (n // sites)
  local = site * (n // sites) + torch.randint(0, n // sites, (n * deg,), device=DEV, generator=g)
  far = torch.randint(0, n, (n * deg,), device=DEV, generator=g)
  pick = torch.rand(n * deg, device=DEV, generator=g) <= same_site
  rows = torch.where(pick, local, far)
  return rows.int(), cols.int()


prewarm()
for n, deg, sites in ((900000, 10, 1), (900000, 10, 64), (7200000, 10, 8), (4000000, 40, 1), (200000, 400, 1)):
  rows, cols = site_graph(n, deg, 0.9, sites)
  vals = torch.ones(n * deg, device=DEV, dtype=torch.float32)
  ms_build = timeit(lambda: S.from_coo((n, n), np.float32, rows, cols, vals), iters=3, warmup=1)
  W = S.from_coo((n, n), np.float32, rows, cols, vals)
  x = torch.rand((n, 1), device=DEV, dtype=torch.float32)
  y = torch.empty((n, 1), device=DEV, dtype=torch.float32)
  alg = W.nnz * 8 + n * (8 + 4 + 4)      # values + column indices, indptr, y, x once
  line = '%8d x %3d sites %3d nnz %9d  build %7.2f ms ' % (n, deg, sites, W.nnz, ms_build)
  for g in (0, 2, 4, 8, 16, 32, 64):
    if g:
      os.environ['SP_SPMV_G'] = str(g)
    else:
      os.environ.pop('SP_SPMV_G', None)
    ms = timeit(lambda: S.spmm(W, x, out=y), iters=10, warmup=2)
    line += ' G%-2d %6.3f ms %6.1f GB/s |' % (g, ms, alg / ms / 1e6)
  os.environ.pop('SP_SPMV_G', None)
  print(line, flush=True)
  for ncol in (8, 64):
    B = torch.rand((n, ncol), device=DEV, dtype=torch.float32)
    C = torch.empty((n, ncol), device=DEV, dtype=torch.float32)
    ms = timeit(lambda: S.spmm(W, B, out=C), iters=5, warmup=1)
    print('    x dense [n, %d]: %7.3f ms  %6.1f GB/s (algorithmic: entries + B + C once)' %
          (ncol, ms, (W.nnz * 8 + n * 8 + 2 * n * ncol * 4) / ms / 1e6), flush=True)
  del W, rows, cols, vals, x, y, B, C
  torch.cuda.empty_cache()
