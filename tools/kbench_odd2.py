"""More cliff hunting: broadcasts on odd widths, other dtypes, 3-D middle-axis reductions, transposes."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spartan_amd as sp  # noqa: E402
from spartan_amd import _hip  # noqa: E402
from tools.kbench import prewarm, timeit  # noqa: E402

ctx = sp.initialize('hip')
prewarm()


def run(name, fn, nbytes):
  for _ in range(3):
    fn()
  _hip.lib().sp_jit_wait()
  ms = timeit(fn, iters=5, warmup=1)
  print('%-34s %7.3f ms %7.1f GB/s' % (name, ms, nbytes / ms / 1e6))


for rows, cols in ((8192, 65536), (8192, 65535)):
  n = rows * cols
  X = sp.Val(val=sp.from_tile_fn((rows, cols), np.float32, lambda ex: torch.rand(ex.shape, device='cuda')).force())
  r = sp.Val(val=sp.from_numpy(np.random.rand(1, cols).astype(np.float32)).force())
  c = sp.Val(val=sp.from_numpy(np.random.rand(rows, 1).astype(np.float32)).force())
  run('%dx%d x - row' % (rows, cols), lambda: (X - r).force(), 8 * n)
  run('%dx%d x * col' % (rows, cols), lambda: (X * c).force(), 8 * n)
  run('%dx%d (x-row)*(x-row) sum0' % (rows, cols), lambda: sp.sum((X - r) * (X - r), 0).optimized().force(), 4 * n)
  run('%dx%d x.T + 1' % (rows, cols), lambda: (X.T + 1).force(), 8 * n)
  run('%dx%d astype f64' % (rows, cols), lambda: X.astype(np.float64).force(), 12 * n)
  del X, r, c
  torch.cuda.empty_cache()
for shape in ((256, 512, 1024), (255, 511, 1023)):
  n = int(np.prod(shape))
  X = sp.Val(val=sp.from_tile_fn(shape, np.float32, lambda ex: torch.rand(ex.shape, device='cuda')).force())
  for ax in (0, 1, 2):
    run('%s sum axis %d' % (shape, ax), lambda: sp.sum(X, ax).force(), 4 * n)
  del X
  torch.cuda.empty_cache()
for dt, bpe in ((np.int64, 16), (np.float64, 16), (np.int32, 8)):
  rows, cols = 4096, 65535
  n = rows * cols
  X = sp.Val(val=sp.from_tile_fn((rows, cols), dt, lambda ex: (torch.rand(ex.shape, device='cuda') * 100).to(
      {np.int64: torch.int64, np.float64: torch.float64, np.int32: torch.int32}[dt])).force())
  run('%s x*3+1' % np.dtype(dt).name, lambda: (X * 3 + 1).optimized().force(), bpe * n)
  run('%s sum1' % np.dtype(dt).name, lambda: sp.sum(X, 1).force(), bpe // 2 * n)
  del X
  torch.cuda.empty_cache()
