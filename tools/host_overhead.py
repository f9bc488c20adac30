"""Host-side overhead of the tile path: wall time per step vs GPU time, and a
cProfile of the driver for the lreg step (many small launches per step)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import spartan_amd as sp
import bench
ctx = sp.initialize('hip')
N, D = 125000, 4096
Xl = sp.from_tile_fn((N, D), np.float32, lambda ex: bench.device_uniform(ex, 0.0, 1.0, 11))
yl = sp.from_tile_fn((N, 1), np.float32, lambda ex: bench.device_uniform(ex, 0.0, 1.0, 12))
w = np.random.RandomState(1).rand(D, 1).astype(np.float32)
def step():
  yp = sp.dot(Xl, w)
  return sp.sum(Xl * (yp - yl), axis=0).optimized().force()
for _ in range(5): step()
torch.cuda.synchronize()
l0 = ctx.backend.launches
t0 = time.perf_counter()
for _ in range(50): step()
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print('lreg step: issue %.3f ms/step, complete %.3f ms/step, %d launches/step' % (t_issue / 50 * 1e3, t_all / 50 * 1e3, (ctx.backend.launches - l0) // 50))
pr = cProfile.Profile(); pr.enable()
for _ in range(50): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
