"""Runs one large fused map outside the prebuilt library; prints which tier served it."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spartan_amd as sp  # noqa: E402
from spartan_amd import _hip  # noqa: E402

ctx = sp.initialize('hip')
x = sp.from_numpy(np.random.rand(4096, 4096).astype(np.float32))
e = lambda: (((x * x + x) * 0.5 - x) / (x + 2.0)).optimized().force()
e()
_hip.lib().sp_jit_wait()
e()
torch.cuda.synchronize()
print('specialised kernels compiled:', _hip.lib().sp_jit_compiled_count())
