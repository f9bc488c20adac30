"""sp_nearest_center: the fp32-MFMA filter against the bf16-split one (stand-alone call, and with prepared points as
inside a fit) over shapes -- what SP_NEAREST_AUTO should pick.  python tools/km_tier_sweep.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import spartan_amd as sp
from spartan_amd import devarray as D, kernels, _hip
import bench
ctx = sp.initialize('hip')
rng = np.random.RandomState(3)
print('%9s %6s %5s | %9s %9s %9s | listed fp32 / split' % ('n', 'k', 'd', 'fp32 ms', 'split ms', 'prepared'))
for n, k, d in [(1250000, 1024, 256), (1250000, 1024, 64), (1250000, 1024, 32), (1250000, 256, 256), (1250000, 64, 64),
                (200000, 1024, 256), (200000, 128, 32), (50000, 512, 128), (1250000, 4096, 256), (500000, 1024, 1024)]:
  x = bench.device_uniform(type('E', (), {'shape': (n, d), 'ul': (0, 0)})(), 0.0, 1.0, 5)
  c = ctx.backend.from_numpy(rng.rand(k, d))
  labels = D.empty((n,), np.int64)
  t32 = bench.event_time(lambda: kernels.nearest_center(x, c, labels, _hip.NEAREST_FUSED), 5, warmup=2)
  ts = bench.event_time(lambda: kernels.nearest_center(x, c, labels, _hip.NEAREST_SPLIT), 5, warmup=2)
  prep = kernels.prepare_points(x)
  tp = bench.event_time(lambda: kernels.nearest_center(x, c, labels, _hip.NEAREST_SPLIT, prepared=prep), 5, warmup=2)
  kernels.nearest_center(x, c, labels, _hip.NEAREST_FUSED_UNCHECKED); l32 = int((labels < 0).sum().item())
  kernels.nearest_center(x, c, labels, _hip.NEAREST_SPLIT_UNCHECKED, prepared=prep); ls = int((labels < 0).sum().item())
  print('%9d %6d %5d | %9.3f %9.3f %9.3f | %d / %d' % (n, k, d, t32, ts, tp, l32, ls))
  del x, c, labels, prep
  D.trim_pool()
