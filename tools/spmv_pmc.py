"""Runs the CSR x vector kernel on the bench.py `sparse` tile a few times (target of `rocprofv3 --kernel-trace
--stats` and of the separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spartan_amd import sparse as S  # noqa: E402

n, deg, sites = 900000, 10, 8
g = torch.Generator(device='cuda')
g.manual_seed(20150708 + 31)
cols = torch.arange(n, device='cuda', dtype=torch.int64).repeat_interleave(deg)
local = (cols // (n // sites)) * (n // sites) + torch.randint(0, n // sites, (n * deg,), device='cuda', generator=g)
far = torch.randint(0, n, (n * deg,), device='cuda', generator=g)
rows = torch.where(torch.rand(n * deg, device='cuda', generator=g) <= 0.9, local, far).int()
W = S.from_coo((n, n), np.float32, rows, cols.int(), torch.ones(n * deg, device='cuda', dtype=torch.float32))
x = torch.rand((n, 1), device='cuda', dtype=torch.float32, generator=g)
y = torch.empty((n, 1), device='cuda', dtype=torch.float32)
for _ in range(5):
  S.spmm(W, x, out=y)
torch.cuda.synchronize()
print('nnz', W.nnz, 'algorithmic bytes per launch', W.nnz * 8 + n * 16)
