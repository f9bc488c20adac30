import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import spartan_amd as sp
from spartan_amd import devarray as D, kernels
from spartan_amd.array import distarray, extent
ctx = sp.initialize('hip')
be = ctx.backend
N, Dm = 125000, 4096
Xl = sp.from_tile_fn((N, Dm), np.float32, lambda ex: bench.device_uniform(ex, 0.0, 1.0, 11)).force()
yl = sp.from_tile_fn((N, 1), np.float32, lambda ex: bench.device_uniform(ex, 0.0, 1.0, 12)).force()
x = ctx.tile(list(Xl.tiles.values())[0]).data
y = ctx.tile(list(yl.tiles.values())[0]).data
w = np.random.rand(Dm, 1).astype(np.float32)
def T(name, fn, reps=200, sync_each=True):
  for _ in range(5): fn()
  D.synchronize()
  tot = 0.0
  for _ in range(reps):
    t0 = time.perf_counter(); r = fn(); tot += time.perf_counter() - t0
    if sync_each: D.synchronize()
  print('%-40s %7.1f us' % (name, tot / reps * 1e6))
T('D.from_numpy(w) idle stream', lambda: D.from_numpy(w))
T('D.empty((4096,))', lambda: D.empty((Dm,), np.float32))
out = D.empty((Dm,), np.float32)
wd = D.from_numpy(w).reshape(Dm)
yd = y.reshape(N)
T('kernels.rowdot_colsum launch', lambda: kernels.rowdot_colsum(x, wd, yd, out))
T('be.rowdot_colsum', lambda: be.rowdot_colsum(x, w * 1.0, y))
T('distarray.create((4096,))', lambda: distarray.create((Dm,), np.float32, reducer=np.add))
T('Xl.fetch(tile)', lambda: Xl.fetch(list(Xl.tiles)[0]), sync_each=False)
g = D.empty((Dm,), np.float32)
T('g.numpy() (d2h 16 KB, idle)', lambda: g.numpy())
def step():
  kernels.rowdot_colsum(x, wd, yd, out)
  return out.numpy()
T('launch + d2h (kernel 0.34 ms)', step)
T('D.synchronize() idle', lambda: D.synchronize())
ev = D.Event()
T('event record+sync idle', lambda: (ev.record(), ev.synchronize()))
