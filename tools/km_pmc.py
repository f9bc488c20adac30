"""Runs the fused k-means assign kernel a few times (target of `rocprofv3 --pmc` passes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spartan_amd import _hip, kernels  # noqa: E402

n, k, d = 1250000, 1024, 256
g = torch.Generator(device='cuda:0')
g.manual_seed(20150708)
x = torch.rand(n, d, dtype=torch.float32, device='cuda:0', generator=g)
c = torch.rand(k, d, dtype=torch.float64, device='cuda:0', generator=g)
labels = torch.empty(n, dtype=torch.int64, device='cuda:0')
for _ in range(3):
  kernels.nearest_center(x, c, labels, _hip.NEAREST_FUSED_UNCHECKED)
torch.cuda.synchronize()
