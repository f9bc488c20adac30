"""Interpreter tier of the fused map -> reduce kernels (what an unseeded program runs on until hipRTC is done):
SP_NO_JIT=1 SP_NO_STATIC=1 [SPARTAN_HIP_LIB=another build] python tools/interp_reduce_time.py"""
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
cases = [('sumsq ax0', lambda: sp.sum((Xv - 0.5) * (Xv - 0.5), axis=0).optimized().force(), 4.0),
         ('sumsq ax1', lambda: sp.sum((Xv - 0.5) * (Xv - 0.5), axis=1).optimized().force(), 4.0),
         ('sumsq all', lambda: sp.sum((Xv - 0.5) * (Xv - 0.5)).optimized().force(), 4.0),
         ('sum x ax0', lambda: sp.sum(Xv, axis=0).force(), 4.0),
         ('max(x*2+1) ax0', lambda: sp.max(Xv * 2.0 + 1.0, axis=0).optimized().force(), 4.0)]
out = []
for name, fn, bpe in cases:
  ms = event_time(fn, 5)
  out.append('%s %.3f ms %.0f GB/s' % (name, ms, bpe * n / ms / 1e6))
print(os.environ.get('SPARTAN_HIP_LIB', 'default'), ' | '.join(out))
