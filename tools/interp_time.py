"""Interpreter tier timing: SP_NO_JIT=1 SP_NO_STATIC=1 python tools/_exp/interp_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('SP_NO_JIT', '1'); os.environ.setdefault('SP_NO_STATIC', '1')
import numpy as np
import spartan_amd as sp
from spartan_amd import devarray as D
from bench import device_uniform, event_time, SEED
sp.initialize('hip')
rows, cols = 8192, 65536
n = rows * cols
X = sp.from_tile_fn((rows, cols), np.float32, lambda ex: device_uniform(ex, 0.0, 1.0, SEED + 7)).force()
Xv = sp.Val(val=X)
cases = [('x+1', lambda: (Xv + 1).force()),
         ('x*x+x', lambda: (Xv * Xv + Xv).optimized().force()),
         ('5op', lambda: (((Xv * Xv + Xv) * 0.5 - Xv) / (Xv + 2.0)).optimized().force()),
         ('12op', lambda: ((((((Xv * Xv + Xv) * 0.5 - Xv) / (Xv + 2.0)) * Xv + 1.0) * Xv - 3.0) * (Xv + 0.25) + Xv * 0.125).optimized().force()),
         ('sqrt_exp', lambda: sp.sqrt(sp.exp(Xv) + 1.0).optimized().force())]
only = [c for c in os.environ.get('INTERP_CASES', '').split(',') if c]
out = []
for name, fn in cases:
  if only and name not in only:
    continue
  ms = event_time(fn, 5)
  out.append('%s %.3f ms %.0f GB/s' % (name, ms, 8.0 * n / ms / 1e6))
print(os.environ.get('SP_MAP_UNROLL', '1'), ' | '.join(out))
