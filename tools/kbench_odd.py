"""Shapes that defeat the 16-B vector paths: odd inner dimensions, offsets, int64 / bool data."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spartan_amd as sp  # noqa: E402
from spartan_amd import _hip, kernels  # noqa: E402
from tools.kbench import prewarm, timeit  # noqa: E402

ctx = sp.initialize('hip')
prewarm()
for rows, cols in ((8192, 65536), (8192, 65535), (8191, 65537)):
  n = rows * cols
  X = sp.from_tile_fn((rows, cols), np.float32, lambda ex: torch.rand(ex.shape, device='cuda')).force()
  Xv = sp.Val(val=X)
  for name, fn, bpe in (('x+1', lambda: (Xv + 1).force(), 8), ('x*x+x', lambda: (Xv * Xv + Xv).optimized().force(), 8),
                        ('sum0', lambda: sp.sum(Xv, 0).force(), 4), ('sum1', lambda: sp.sum(Xv, 1).force(), 4),
                        ('sumN', lambda: sp.sum(Xv).force(), 4), ('argmax1', lambda: sp.argmax(Xv, 1).force(), 4),
                        ('x>0.5', lambda: (Xv > 0.5).force(), 5), ('slice+1', lambda: (Xv[1:, 1:] + 1).force(), 8)):
    for _ in range(3):
      fn()
    _hip.lib().sp_jit_wait()
    ms = timeit(fn, iters=5, warmup=1)
    print('%5dx%5d %-8s %7.3f ms %7.1f GB/s' % (rows, cols, name, ms, bpe * n / ms / 1e6))
  del X, Xv
  torch.cuda.empty_cache()
