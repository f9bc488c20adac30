import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from _dev import D, kernels
rng = np.random.RandomState(5)
bad = 0
for (M, N, K) in [(2304, 2304, 2304), (3072, 3072, 3072), (5000, 5000, 5000), (2300, 2304, 2056), (4608, 4608, 1024)]:
  a = rng.randint(-3, 4, size=(M, K)).astype(np.float32)
  b = rng.randint(-3, 4, size=(K, N)).astype(np.float32)
  want = D.from_numpy(a.astype(np.float64).dot(b.astype(np.float64)).astype(np.float32))
  A, B = D.from_numpy(a), D.from_numpy(b)
  c = D.empty((M, N), np.float32)
  for rep in range(150):
    c.fill(-7.0)
    kernels.gemm_f32(A, B, c)
    diff = int((c != want).sum().item())
    if diff:
      bad += 1
      print('MISMATCH', (M, N, K), 'rep', rep, diff, 'elements')
      break
print('stress done,', bad, 'bad')
