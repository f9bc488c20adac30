"""Times sp_gemm_f32 at 8192^3 under each SP_GEMM_VARIANT (one process per variant: the knob is read once)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == 'one':
  sys.path.insert(0, ROOT)
  import torch
  from spartan_amd import kernels
  n = 8192
  a = torch.rand(n, n, device='cuda:0') * 2 - 1
  b = torch.rand(n, n, device='cuda:0') * 2 - 1
  c = torch.empty(n, n, device='cuda:0')
  for _ in range(8):
    kernels.gemm_f32(a, b, c)
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(10):
    kernels.gemm_f32(a, b, c)
  e1.record()
  torch.cuda.synchronize()
  ms = e0.elapsed_time(e1) / 10
  ref = (a[:64].double() @ b.double()).float()
  err = (c[:64] - ref).abs().max().item()
  print('variant %s: %.3f ms = %.1f TFLOP/s, max err vs fp64 on 64 rows %.2e' % (os.environ.get('SP_GEMM_VARIANT'), ms, 2.0 * n ** 3 / ms / 1e9, err))
else:
  for v in sys.argv[1:] or ['0', '1', '4', '5']:
    env = dict(os.environ, SP_GEMM_VARIANT=v)
    subprocess.run([sys.executable, os.path.abspath(__file__), 'one'], env=env, timeout=120)
