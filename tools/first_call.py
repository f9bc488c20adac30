import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t_imp = time.perf_counter()
import bench
import spartan_amd as sp
from spartan_amd import devarray as D, kernels, lower, _hip
ctx = sp.initialize('hip')
be = ctx.backend
rows, cols = 8192, 65536
n = rows * cols
X = sp.from_tile_fn((rows, cols), np.float32, lambda ex: bench.device_uniform(ex, 0.0, 1.0, 7)).force()
Xv = sp.Val(val=X)
D.synchronize()
print('init + fill: %.1f ms since import' % ((time.perf_counter() - t_imp) * 1e3))
marks = []
orig_infer, orig_mf = lower.infer, kernels.map_fused
def infer(*a, **k):
  t0 = time.perf_counter(); r = orig_infer(*a, **k); marks.append(('infer', (time.perf_counter() - t0) * 1e6)); return r
def mf(*a, **k):
  t0 = time.perf_counter(); r = orig_mf(*a, **k); marks.append(('map_fused call', (time.perf_counter() - t0) * 1e6)); return r
lower.infer = infer; kernels.map_fused = mf
for _ in range(3):
  (Xv * Xv + Xv).optimized().force(); (Xv + 1).force()
D.synchronize()
orig_run = be._run_map
def run_map(*a, **k):
  t0 = time.perf_counter(); r = orig_run(*a, **k); marks.append(('_run_map', (time.perf_counter() - t0) * 1e6)); return r
be._run_map = run_map
orig_key = be._lowering_key
def lkey(*a, **k):
  t0 = time.perf_counter(); r = orig_key(*a, **k); marks.append(('key', (time.perf_counter() - t0) * 1e6)); return r
be._lowering_key = lkey
pending = [(((Xv * Xv + Xv) * 0.5 - Xv) / (Xv + 2.0)).optimized() for _ in range(3)]
for i, e in enumerate(pending):
  D.synchronize()
  del marks[:]
  e0, e1 = D.Event(), D.Event()
  t0 = time.perf_counter()
  e0.record()
  e.force()
  t1 = time.perf_counter()
  e1.record(); e1.synchronize()
  t2 = time.perf_counter()
  print('call %d: events %.3f ms  host-until-return %.1f us  total wall %.1f us   %s' % (i, e0.elapsed_ms(e1), (t1 - t0) * 1e6, (t2 - t0) * 1e6, [(k, round(v, 1)) for k, v in marks]))

import cProfile, pstats, gc
print('gc counts', gc.get_count(), gc.get_threshold())
pending = [(((Xv * Xv + Xv) * 0.25 - Xv) / (Xv + 3.0)).optimized() for _ in range(3)]
for i, e in enumerate(pending):
  D.synchronize()
  pr = cProfile.Profile()
  t0 = time.perf_counter()
  pr.enable(); e.force(); pr.disable()
  print('profiled call %d: %.1f us' % (i, (time.perf_counter() - t0) * 1e6))
  if i in (0, 2):
    pstats.Stats(pr).sort_stats('tottime').print_stats(8)
