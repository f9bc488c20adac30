import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import spartan_amd as sp
from spartan_amd import _hip, kernels
import bench
ctx = sp.initialize('hip')
rows, cols = 8192, 65536
X = sp.from_tile_fn((rows, cols), np.float32, lambda ex: bench.device_uniform(ex, 0.0, 1.0, 7)).force()
Xv = sp.Val(val=X)
n = rows*cols
for rep in range(3):
  ms = bench.event_time(lambda: sp.sum((Xv - 0.5) * (Xv - 0.5), axis=0).optimized().force(), 10)
  print('sum_sq_dev axis0 GB/s', round(4.0*n/ms/1e6,1), 'compiled', _hip.lib().sp_jit_compiled_count())
ms = bench.event_time(lambda: sp.sum(Xv, 0).force(), 10)
print('sum axis0', round(4.0*n/ms/1e6,1))
