import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import spartan_amd as sp
from spartan_amd import devarray as D, kernels
from spartan_amd import expr
from spartan_amd.examples.sklearn.cluster import k_means_ as KM
ctx = sp.initialize('hip')
n, k, d = 1250000, 1024, 256
X = sp.Val(val=sp.from_tile_fn((n, d), np.float32, lambda ex: bench.device_uniform(ex, 0.0, 1.0, 21)).force())
centers = np.random.RandomState(0).rand(k, d)
km = KM.KMeans(k, 1)
acc = {}
def lap(name, t0):
  t1 = time.perf_counter(); acc[name] = acc.get(name, 0.0) + (t1 - t0); return t1
for it in range(12):
  if it == 2: acc.clear(); D.synchronize(); T0 = time.perf_counter()
  t = time.perf_counter()
  labels = expr.map2(X, 0, fn=KM.kmeans_map2_dist_mapper, fn_kw={'centers': centers}, shape=(n,))
  counts = expr.map2(labels, 0, fn=KM.kmeans_count_mapper, fn_kw={'centers_count': k}, shape=(k,), reducer=np.add)
  sums = expr.map2((X, labels), (0, 0), fn=KM.kmeans_center_mapper, fn_kw={'centers_count': k}, shape=(k, d), reducer=np.add)
  t = lap('build', t)
  counts, sums = counts.optimized(), sums.optimized()
  t = lap('optimize', t)
  counts.evaluate()
  t = lap('counts.evaluate (assign+bincount issue)', t)
  sums.evaluate()
  t = lap('sums.evaluate (segment_sum issue)', t)
  c = counts.glom()
  t = lap('counts.glom (wait)', t)
  s = sums.glom()
  t = lap('sums.glom (d2h 1 MB)', t)
  s, c = km._finish(s, c)
  centers = s / c.reshape(k, 1)
  t = lap('host finish', t)
D.synchronize()
tot = time.perf_counter() - T0
for k_, v in acc.items(): print('%-45s %8.1f us' % (k_, v * 1e5))
print('iteration %.3f ms' % (tot * 1e2))
cd = D.from_numpy(centers)
def T(name, fn, reps=50):
  for _ in range(3): fn()
  D.synchronize(); t0 = time.perf_counter()
  for _ in range(reps): fn()
  D.synchronize(); print('%-45s %8.1f us' % (name, (time.perf_counter() - t0) / reps * 1e6))
T('from_numpy(centers fp64 2MB) + sync', lambda: D.from_numpy(centers))
big = D.empty((k, d), np.float32)
T('numpy() of 1 MB', lambda: big.numpy())
