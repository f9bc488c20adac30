import os, sys, time, functools
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import spartan_amd as sp
from spartan_amd import devarray as D, kernels, backend_hip
from spartan_amd.array import distarray, tile
import importlib
map_mod = importlib.import_module('spartan_amd.expr.map')
from spartan_amd.examples.sklearn.cluster import k_means_ as KM
ctx = sp.initialize('hip')
be = ctx.backend
n, k, d = 1250000, 1024, 256
X = sp.Val(val=sp.from_tile_fn((n, d), np.float32, lambda ex: bench.device_uniform(ex, 0.0, 1.0, 21)).force())
centers = np.random.RandomState(0).rand(k, d)
log = []
depth = [0]
def wrap(obj, name, label=None):
  fn = getattr(obj, name)
  @functools.wraps(fn)
  def inner(*a, **kw):
    t0 = time.perf_counter(); depth[0] += 1
    try: return fn(*a, **kw)
    finally:
      depth[0] -= 1; log.append((depth[0], label or name, (time.perf_counter() - t0) * 1e6))
  setattr(obj, name, inner)
for name in ('nearest_center', 'bincount', 'segment_sum', 'cached_numpy', 'astype', '_run_map', 'from_numpy', 'empty', 'copy', 'contiguous'):
  wrap(be, name)
wrap(distarray, 'create', 'distarray.create')
wrap(distarray.DistArrayImpl, 'fetch'); wrap(distarray.DistArrayImpl, 'update'); wrap(distarray.UpdateBatch, 'flush')
wrap(distarray, 'run_kernel'); wrap(map_mod, 'join_mapper')
wrap(kernels, 'nearest_center', 'kernels.nearest_center'); wrap(kernels, 'update', 'kernels.update')
wrap(tile.Tile, 'update', 'Tile.update')
km = KM.KMeans(k, 1)
for it in range(4):
  del log[:]
  D.synchronize()
  t0 = time.perf_counter()
  centers, labels = km.fit(X, centers, implementation='map2', reducer=np.add)
  tot = (time.perf_counter() - t0) * 1e6
print('iteration %.0f us' % tot)
for dep, name, us in log:
  if us > 8: print('%s%-30s %8.1f' % ('  ' * dep, name, us))
