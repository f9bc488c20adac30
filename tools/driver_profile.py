"""Where the host time of the two workload drivers goes (cProfile of the timed loops of bench.py's lreg / kmeans
sections on the per-GPU tiles).  python tools/driver_profile.py [lreg|kmeans]"""
import cProfile
import io
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import spartan_amd as sp  # noqa: E402
from spartan_amd import devarray as D  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else 'lreg'
ctx = sp.initialize('hip')
if which == 'lreg':
  from spartan_amd.examples import lreg
  N, Dm = 125000, 4096
  X = sp.Val(val=sp.from_tile_fn((N, Dm), np.float32, lambda ex: bench.device_uniform(ex, 0.0, 1.0, 11)).force())
  y = sp.Val(val=sp.from_tile_fn((N, 1), np.float32, lambda ex: bench.device_uniform(ex, 0.0, 1.0, 12)).force())
  w = np.random.RandomState(0).rand(Dm, 1).astype(np.float32)
  w = lreg.fit(X, y, 5, alpha=1e-10, w=w)
  D.synchronize()
  fn = lambda: lreg.fit(X, y, 100, alpha=1e-10, w=w)   # noqa: E731
else:
  from spartan_amd.examples.sklearn.cluster import KMeans
  n, k, d = 1250000, 1024, 256
  X = sp.Val(val=sp.from_tile_fn((n, d), np.float32, lambda ex: bench.device_uniform(ex, 0.0, 1.0, 21)).force())
  c0 = np.random.RandomState(0).rand(k, d)
  c0, _ = KMeans(k, 2).fit(X, c0, implementation='map2', reducer=np.add)
  D.synchronize()

  def fn():
    c = c0
    for _ in range(10):
      c, _ = KMeans(k, 1).fit(X, c, implementation='map2', reducer=np.add)
t0 = time.perf_counter()
fn()
D.synchronize()
print('%s: %.3f ms per step (unprofiled)' % (which, (time.perf_counter() - t0) * 1e3 / (100 if which == 'lreg' else 10)))
pr = cProfile.Profile()
pr.enable()
fn()
D.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats(os.environ.get('SORT', 'cumulative')).print_stats(int(os.environ.get('ROWS', '45')))
print(s.getvalue()[:20000])
