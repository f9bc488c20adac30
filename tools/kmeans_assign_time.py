"""Times sp_nearest_center on BASELINE configs[3]'s tile (1.25 M x 256 points, 1024 centres) -- the number DESIGN.md
quotes for the assign step.  SP_KM_TAIL_SPLIT=0 turns the split of the last workgroup round off."""
import torch, numpy as np, time, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from spartan_amd import kernels
n,k,d=1250000,1024,256
x=torch.rand(n,d,device='cuda'); c=torch.rand(k,d,device='cuda',dtype=torch.float64)
lab=torch.empty(n,dtype=torch.int64,device='cuda')
for _ in range(3): kernels.nearest_center(x,c,lab)
torch.cuda.synchronize()
t=time.time()
for _ in range(20): kernels.nearest_center(x,c,lab)
torch.cuda.synchronize()
ms=(time.time()-t)/20*1e3
print('assign ms',ms,'TF',2*n*k*d/ms/1e9, 'frac',2*n*k*d/ms/1e9/157.3)
