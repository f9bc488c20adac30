"""Host time of the driver-side paths on the HIP backend (what the device waits for between launches):
time until force() RETURNS (launches are asynchronous), per phase for one lreg step, and the plan / lowering
table statistics.  python tools/host_times.py"""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import spartan_amd as sp  # noqa: E402
from spartan_amd import devarray as D  # noqa: E402
from spartan_amd.examples import lreg  # noqa: E402

plan = importlib.import_module('spartan_amd.expr.plan')
ctx = sp.initialize('hip')
rows, cols = 1024, 4096
X = sp.Val(val=sp.from_tile_fn((rows, cols), np.float32, lambda ex: bench.device_uniform(ex, 0.0, 1.0, 7)).force())


def host_us(fn, reps=300):
  for _ in range(20):
    fn()
  D.synchronize()
  t0 = time.perf_counter()
  for _ in range(reps):
    fn()
  dt = time.perf_counter() - t0
  D.synchronize()
  return dt / reps * 1e6


out = {}
out['x_plus_1_force_us'] = host_us(lambda: (X + 1).force())
out['xx_plus_x_optimized_force_us'] = host_us(lambda: (X * X + X).optimized().force())
out['chain5_optimized_force_us'] = host_us(lambda: (((X * X + X) * 0.5 - X) / (X + 2.0)).optimized().force())
out['sum_axis0_force_us'] = host_us(lambda: sp.sum(X, 0).force())
out['sum_none_force_us'] = host_us(lambda: sp.sum(X).force())
out['argmax1_force_us'] = host_us(lambda: sp.argmax(X, 1).force())
N, Dm = 125000, 4096
Xl = sp.Val(val=sp.from_tile_fn((N, Dm), np.float32, lambda ex: bench.device_uniform(ex, 0.0, 1.0, 11)).force())
yl = sp.Val(val=sp.from_tile_fn((N, 1), np.float32, lambda ex: bench.device_uniform(ex, 0.0, 1.0, 12)).force())
w = np.random.RandomState(0).rand(Dm, 1).astype(np.float32)
acc = {'build': 0.0, 'optimize': 0.0, 'evaluate_issue': 0.0, 'wait+d2h': 0.0, 'host_update': 0.0}
for it in range(105):
  t0 = time.perf_counter()
  g = lreg.gradient(Xl, yl, w)
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
out['lreg_step_us'] = {k: round(v * 1e4, 1) for k, v in acc.items()}
out['lreg_step_total_us'] = round(sum(acc.values()) * 1e4, 1)
out['plan_table'] = dict(plan.stats)
out['lowering_hits'] = ctx.backend.lowering_hits
print({k: (round(v, 1) if isinstance(v, float) else v) for k, v in out.items()})
