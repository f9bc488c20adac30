"""First launch of a seeded fused map -> reduce program on an idle device: where do the milliseconds go?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import spartan_amd as sp
from spartan_amd import devarray as D, _hip
from bench import device_uniform, SEED
ctx = sp.initialize('hip')
rows, cols = 8192, 65536
n = rows * cols
X = sp.from_tile_fn((rows, cols), np.float32, lambda ex: device_uniform(ex, 0.0, 1.0, SEED + 7)).force()
Xv = sp.Val(val=X)
time.sleep(float(os.environ.get('SLEEP', '1.0')))
if os.environ.get('WARM'):
  t0 = time.perf_counter(); sp.sum(sp.ones((64, 64)) * 2.0, axis=0).optimized().glom(); print('tiny reduction first: %.3f ms' % ((time.perf_counter() - t0) * 1e3))
lib = _hip.lib()
for name, build in (('sumsq ax0', lambda: sp.sum((Xv - 0.5) * (Xv - 0.5), axis=0).optimized()),
                    ('sumsq ax1', lambda: sp.sum((Xv - 0.5) * (Xv - 0.5), axis=1).optimized()),
                    ('sum(x*y) ax0', lambda: sp.sum(Xv * Xv, axis=0).optimized()),
                    ('max(2x+1) ax0', lambda: sp.max(Xv * 2.0 + 1.0, axis=0).optimized())):
  for rep in range(3):
    e = build()
    D.synchronize()
    c0 = lib.sp_jit_compiled_count()
    t0 = time.perf_counter()
    e.force()
    t1 = time.perf_counter()
    D.synchronize()
    t2 = time.perf_counter()
    print('%-14s call %d: host %.3f ms, total %.3f ms, jit compiled %d -> %d' % (name, rep, (t1 - t0) * 1e3, (t2 - t0) * 1e3, c0, lib.sp_jit_compiled_count()), flush=True)
