"""Times the sparse multiply of bench.py's sparse section (900 000 pages x 10 links, 90 % site-local, 8 sites) with the
column-blocked plan and with the stream kernel (SP_SPMV_BLOCKED=0), and the one-off plan build."""
import os
import time

import numpy as np

from _dev import D, timeit
from spartan_amd import sparse as S

n, deg, sites = 900000, 10, 8
rng = np.random.RandomState(31)
cols = np.repeat(np.arange(n, dtype=np.int64), deg)
local = (cols // (n // sites)) * (n // sites) + rng.randint(0, n // sites, size=n * deg)
far = rng.randint(0, n, size=n * deg)
rows = D.from_numpy(np.where(rng.rand(n * deg) <= 0.9, local, far).astype(np.int32))
W = S.from_coo((n, n), np.float32, rows, D.from_numpy(cols.astype(np.int32)), D.full((n * deg,), 1, np.float32))
p = D.full((n, 1), 1.0 / n, np.float32)
out = D.empty((n, 1), np.float32)
D.synchronize()
t0 = time.perf_counter()
bp = S.spmv_block_plan(W)
D.synchronize()
print('block plan: %s, built in %.2f ms' % ('yes' if bp is not False else 'no', (time.perf_counter() - t0) * 1e3))
ms = timeit(lambda: S.spmm(W, p, out=out), 50)
nbytes = 8.0 * W.nnz + 16.0 * n
print('spmv %.1f us = %.0f GB/s of the algorithmic bytes (%s)' % (ms * 1e3, nbytes / ms / 1e6, 'blocked' if bp is not False else 'stream'))
ref = out.numpy().copy()
os.environ['SP_SPMV_BLOCKED'] = '0'
W._block_plan = None
ms = timeit(lambda: S.spmm(W, p, out=out), 50)
print('spmv %.1f us = %.0f GB/s (stream kernel)' % (ms * 1e3, nbytes / ms / 1e6))
print('bit-identical:', bool(np.array_equal(ref, out.numpy())))
