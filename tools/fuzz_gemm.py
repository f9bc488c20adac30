"""Random shapes / paddings / accumulate flags of sp_gemm_f32 with small-integer operands: every product must be exact
(SP_GEMM_VARIANT pins a kernel).  Usage: python tools/fuzz_gemm.py [seed]"""
import sys
import time

import numpy as np

from _dev import D, kernels

rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
t0 = time.time()
for it in range(300):
  M = int(rng.choice([1, 7, 128, 129, 255, 256, 257, 384, 1000, 1024, 2048, 3072, 4100]))
  N = int(rng.choice([1, 4, 12, 128, 132, 256, 260, 1000, 1024, 2048, 3072]))
  K = int(rng.choice([1, 8, 16, 17, 20, 36, 48, 64, 100, 250, 256, 1000, 1024, 4096, 5000]))
  padA, padB, padC = (int(rng.choice([0, 0, 4, 3])) for _ in range(3))
  a = rng.randint(-4, 5, size=(M, K)).astype(np.float32)
  b = rng.randint(-4, 5, size=(K, N)).astype(np.float32)
  A = D.from_numpy(np.pad(a, ((0, 0), (0, padA))))[:, :K]
  B = D.from_numpy(np.pad(b, ((0, 0), (0, padB))))[:, :N]
  c0 = rng.randint(-3, 4, size=(M, N)).astype(np.float32)
  C = D.from_numpy(np.pad(c0, ((0, 0), (0, padC))))[:, :N]
  acc = bool(rng.rand() < 0.4)
  kernels.gemm_f32(A, B, C, accumulate=acc)
  want = a.astype(np.float64) @ b.astype(np.float64) + (c0.astype(np.float64) if acc else 0)   # exact: small integers
  got = C.numpy()
  if not np.array_equal(got, want.astype(np.float32)):
    print('MISMATCH', M, N, K, padA, padB, padC, acc, np.abs(got - want).max())
    bad += 1
print('done', it + 1, 'cases', bad, 'bad', round(time.time() - t0, 1), 's')
