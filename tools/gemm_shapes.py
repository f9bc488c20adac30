"""Times sp_gemm_f32 on a list of M,N,K shapes (the per-chunk GEMMs of the K-split dot pipeline among them).
python tools/gemm_shapes.py [M,N,K ...]"""
import sys

import numpy as np

from _dev import D, kernels, rand, timeit

shapes = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]] or \
    [(2048, 2048, 2048), (3072, 3072, 3072), (4096, 4096, 4096), (5000, 5000, 5000), (32768, 4096, 4096), (8192, 8192, 8192)]
for m, n, k in shapes:
  a, b = rand((m, k), -1, 1, seed=1), rand((k, n), -1, 1, seed=2)
  c = D.empty((m, n), np.float32)
  reps = max(3, int(2e13 / (2.0 * m * n * k)))
  ms = timeit(lambda: kernels.gemm_f32(a, b, c), reps, warmup=5)
  print('%6d x %6d x %6d: %8.3f ms = %6.1f TFLOP/s = %.3f of peak (%d launches)' %
        (m, n, k, ms, 2.0 * m * n * k / ms / 1e9, 2.0 * m * n * k / ms / 1e9 / 157.3, reps), flush=True)
