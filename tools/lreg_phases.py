"""Host time of one lreg step by phase (no profiler): build + optimise, evaluate (two launches), glom."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import spartan_amd as sp
from spartan_amd import devarray as D
from spartan_amd.examples import lreg
ctx = sp.initialize('hip')
N, Dm = 125000, 4096
X = sp.Val(val=sp.from_tile_fn((N, Dm), np.float32, lambda ex: bench.device_uniform(ex, 0.0, 1.0, 11)).force())
y = sp.Val(val=sp.from_tile_fn((N, 1), np.float32, lambda ex: bench.device_uniform(ex, 0.0, 1.0, 12)).force())
w = np.random.RandomState(0).rand(Dm, 1).astype(np.float32)
acc = {'build': 0.0, 'optimize': 0.0, 'evaluate_issue': 0.0, 'wait+d2h': 0.0, 'host_update': 0.0}
for it in range(105):
  t0 = time.perf_counter()
  g = lreg.gradient(X, y, w)
  t1 = time.perf_counter()
  o = g.optimized()
  t2 = time.perf_counter()
  r = o.evaluate()
  t3 = time.perf_counter()
  gv = r.glom()
  t4 = time.perf_counter()
  w = w - gv.reshape((Dm, 1)) * 1e-10
  t5 = time.perf_counter()
  if it >= 5:
    for k, v in zip(acc, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
      acc[k] += v
print({k: round(v * 10, 4) for k, v in acc.items()}, 'ms per step; total', round(sum(acc.values()) * 10, 4))
