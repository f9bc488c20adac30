"""Times sp_gemm_f32 on a list of M,N,K shapes (the per-chunk GEMMs of the K-split dot pipeline among them)."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from spartan_amd import kernels

shapes = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]] or [(4096, 4096, 4096), (32768, 4096, 4096), (8192, 8192, 8192)]
for m, n, k in shapes:
  a = torch.rand(m, k, device='cuda:0') * 2 - 1
  b = torch.rand(k, n, device='cuda:0') * 2 - 1
  c = torch.empty(m, n, device='cuda:0')
  for _ in range(5):
    kernels.gemm_f32(a, b, c)
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  reps = max(3, int(2e13 / (2.0 * m * n * k)))
  e0.record()
  for _ in range(reps):
    kernels.gemm_f32(a, b, c)
  e1.record()
  torch.cuda.synchronize()
  ms = e0.elapsed_time(e1) / reps
  print('%6d x %6d x %6d: %8.3f ms = %6.1f TFLOP/s (%d launches)' % (m, n, k, ms, 2.0 * m * n * k / ms / 1e9, reps))
