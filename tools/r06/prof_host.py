import cProfile, pstats, sys, os, time, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
import spartan_amd as sp
from spartan_amd import devarray as D
ctx = sp.initialize('hip')
X = sp.Val(val=sp.from_tile_fn((1024, 4096), np.float32, lambda ex: bench.device_uniform(ex, 0.0, 1.0, 7)).force())
progs = {'x_plus_1': lambda: (X + 1).force(),
         'chain5': lambda: (((X * X + X) * 0.5 - X) / (X + 2.0)).optimized().force(),
         'sum0': lambda: sp.sum(X, 0).force()}
for name, fn in progs.items():
  for _ in range(50): fn()
  D.synchronize()
  t0 = time.perf_counter()
  for _ in range(500): fn()
  dt = time.perf_counter() - t0
  D.synchronize()
  print('%s: %.1f us per force' % (name, dt / 500 * 1e6))
  pr = cProfile.Profile()
  pr.enable()
  for _ in range(500): fn()
  pr.disable()
  D.synchronize()
  s = io.StringIO()
  pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28)
  print('\n'.join(l[:150] for l in s.getvalue().splitlines()[:48]))
