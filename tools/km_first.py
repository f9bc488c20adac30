import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import spartan_amd as sp
from spartan_amd import devarray as D, kernels, _hip
import bench
ctx = sp.initialize('hip')
n, k, d = 1250000, 1024, 256
x = bench.device_uniform(type('E', (), {'shape': (n, d), 'ul': (0, 0)})(), 0.0, 1.0, bench.SEED + 21)
c = ctx.backend.from_numpy(np.random.RandomState(bench.SEED).rand(k, d))
labels = D.empty((n,), np.int64)
prep = kernels.prepare_points(x)
ms = bench.event_time(lambda: kernels.nearest_center(x, c, labels, _hip.NEAREST_SPLIT_UNCHECKED, prepared=prep), 10, warmup=3)
print(os.environ.get('SPARTAN_HIP_LIB', 'default').split('/')[-1], 'first pass only: %.3f ms' % ms)
