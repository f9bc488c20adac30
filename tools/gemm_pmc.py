"""Runs the 8192^3 fp32 GEMM a few times (target of `rocprofv3 --pmc` passes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spartan_amd import kernels  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
a = torch.rand(n, n, device='cuda:0') * 2 - 1
b = torch.rand(n, n, device='cuda:0') * 2 - 1
c = torch.empty(n, n, device='cuda:0')
for _ in range(4):
  kernels.gemm_f32(a, b, c)
torch.cuda.synchronize()
