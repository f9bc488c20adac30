"""sp_sort_rows on configs[2]-sized tiles: the LDS bitonic path (lines <= 4096) and the radix path."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spartan_amd import kernels  # noqa: E402
from tools.kbench import prewarm, timeit  # noqa: E402

prewarm()
for rows, cols, dt in ((131072, 4096, torch.float32), (2097152, 256, torch.float32), (33554432, 16, torch.float32),
                       (1048576, 256, torch.float64), (16777216, 16, torch.int64), (131072, 2048, torch.float64), (65536, 4096, torch.float64),
                       (8192, 65536, torch.float32), (1, 268435456, torch.float32), (8192, 16384, torch.float64),
                       (8192, 32768, torch.int32)):
  n = rows * cols
  if dt in (torch.float32, torch.float64):
    x = torch.rand((rows, cols), device='cuda', dtype=dt)
  else:
    x = torch.randint(-2**31, 2**31 - 1, (rows, cols), device='cuda', dtype=dt)
  if dt == torch.int64:
    x = torch.randint(-2**62, 2**62, (rows, cols), device='cuda', dtype=dt)
  for what, kw in (('sort', dict(values=True, indices=False)), ('argsort', dict(values=False, indices=True))):
    ms = timeit(lambda: kernels.sort_rows(x, **kw), iters=3, warmup=1)
    es = x.element_size()
    alg = n * (es + (es if what == 'sort' else 8))
    print('%9d x %9d %-8s %-8s %9.3f ms  %7.2f Gkeys/s  %7.1f GB/s (read + write once)' %
          (rows, cols, str(dt).split('.')[-1], what, ms, n / ms / 1e6, alg / ms / 1e6), flush=True)
  del x
  torch.cuda.empty_cache()
