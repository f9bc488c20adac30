"""One sort and one argsort of an 8192 x 65536 fp32 tile and of 131072 x 4096 (target of rocprofv3 --kernel-trace --stats)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spartan_amd import kernels  # noqa: E402

for rows, cols in ((8192, 65536), (131072, 4096)):
  x = torch.rand((rows, cols), device='cuda', dtype=torch.float32)
  for _ in range(3):
    kernels.sort_rows(x, values=True, indices=True)
  torch.cuda.synchronize()
  del x
  torch.cuda.empty_cache()
