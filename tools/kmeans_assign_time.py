"""Times sp_nearest_center on BASELINE configs[3]'s tile (1.25 M x 256 points, 1024 centres) -- the number DESIGN.md
quotes for the assign step.  SP_KM_TAIL_SPLIT=0 turns the split of the last workgroup round off."""
import numpy as np

from _dev import D, kernels, rand, timeit

n, k, d = 1250000, 1024, 256
x = rand((n, d), seed=21)
c = D.from_numpy(np.random.RandomState(0).rand(k, d))
lab = D.empty((n,), np.int64)
ms = timeit(lambda: kernels.nearest_center(x, c, lab), 20)
print('assign ms', ms, 'TF', 2 * n * k * d / ms / 1e9, 'frac', 2 * n * k * d / ms / 1e9 / 157.3)
