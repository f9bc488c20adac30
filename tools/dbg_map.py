import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import spartan_amd as sp
from spartan_amd import kernels
import bench
ctx = sp.initialize('hip')
rows, cols = 8192, 65536
X = sp.from_tile_fn((rows, cols), np.float32, lambda ex: bench.device_uniform(ex, 0.0, 1.0, 7)).force()
Xv = sp.Val(val=X)
for name, fn in (('x*x+x', lambda: (Xv * Xv + Xv).optimized().force()), ('x+1', lambda: (Xv + 1).force()), ('x*x+x unopt', lambda: (Xv * Xv + Xv).force())):
  for it in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    e0, e1 = kernels.Event(), kernels.Event()
    e0.record(); r = fn(); e1.record(); e1.synchronize()
    t1 = time.perf_counter()
    print(name, it, 'wall %.2f ms' % ((t1 - t0) * 1e3), 'gpu %.3f ms' % e0.elapsed_ms(e1), 'launches', ctx.backend.launches, 'mem GiB %.1f' % (torch.cuda.memory_allocated() / 2**30), 'reserved %.1f' % (torch.cuda.memory_reserved() / 2**30))
