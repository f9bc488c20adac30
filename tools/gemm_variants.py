"""Times sp_gemm_f32 at 8192^3 under each SP_GEMM_VARIANT (one process per variant: the knob is read once)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == 'one':
  import numpy as np
  from _dev import D, kernels, rand, timeit
  n = 8192
  a, b = rand((n, n), -1, 1, seed=1), rand((n, n), -1, 1, seed=2)
  c = D.empty((n, n), np.float32)
  ms = timeit(lambda: kernels.gemm_f32(a, b, c), 10, warmup=8)
  ref = a[:64].numpy().astype(np.float64) @ b.numpy().astype(np.float64)
  err = float(np.abs(c[:64].numpy() - ref).max())
  print('variant %s: %.3f ms = %.1f TFLOP/s, max err vs fp64 on 64 rows %.2e' % (os.environ.get('SP_GEMM_VARIANT'), ms, 2.0 * n ** 3 / ms / 1e9, err))
else:
  for v in sys.argv[1:] or ['0', '1', '4', '5']:
    env = dict(os.environ, SP_GEMM_VARIANT=v)
    subprocess.run([sys.executable, os.path.abspath(__file__), 'one'], env=env, timeout=120)
