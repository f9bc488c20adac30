import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import spartan_amd as sp
from spartan_amd import devarray as D, kernels, _hip
from spartan_amd.examples.sklearn.cluster import KMeans
import bench
ctx = sp.initialize('hip')
n, k, d = 1250000, 1024, 256
X = sp.from_tile_fn((n, d), np.float32, lambda ex: bench.device_uniform(ex, 0.0, 1.0, bench.SEED + 21)).force()
Xv = sp.Val(val=X)
x = ctx.tile(list(X.tiles.values())[0]).data
c = np.random.RandomState(bench.SEED).rand(k, d)
labels = D.empty((n,), np.int64)
os.environ['SP_KM_SPLIT'] = '0'
for it in range(6):
  cd = ctx.backend.from_numpy(c)
  out = []
  for tier in (_hip.NEAREST_FUSED_UNCHECKED, _hip.NEAREST_SPLIT_UNCHECKED):
    kernels.nearest_center(x, cd, labels, tier)
    out.append(int((labels < 0).sum().item()))
  print('iteration', it, 'listed: fp32 window', out[0], ' split window', out[1], ' of', n, ' |c| spread', float(np.std(c)))
  c, _ = KMeans(k, 1).fit(Xv, c, implementation='map2', reducer=np.add)
