"""Times the accumulate step of BASELINE configs[3]'s tile (1.25 M x 256 points, 1024 clusters): sp_bincount_i64 and
sp_segment_sum on the labels of one assignment, separately and together, and checks both against NumPy."""
import numpy as np

from _dev import D, kernels, rand, timeit

n, k, d = 1250000, 1024, 256
x = rand((n, d), seed=21)
c = D.from_numpy(np.random.RandomState(0).rand(k, d))
lab = D.empty((n,), np.int64)
kernels.nearest_center(x, c, lab)
sums = D.empty((k, d), np.float32)
counts = D.empty((k,), np.int64)
kernels.bincount(lab, k, counts)
kernels.segment_sum(x, lab, k, sums)
lh = lab.numpy()
want = np.bincount(lh, minlength=k)
assert np.array_equal(counts.numpy(), want), 'bincount'
xs = x.numpy()
ref = np.zeros((k, d), np.float64)
np.add.at(ref, lh, xs.astype(np.float64))
err = np.abs(sums.numpy() - ref).max()
print('segment_sum max |err| vs fp64 %.3e (rows per cluster ~%d)' % (err, n // k))
assert err < 1e-2
tb = timeit(lambda: kernels.bincount(lab, k, counts), 50)
ts = timeit(lambda: kernels.segment_sum(x, lab, k, sums), 50)
tt = timeit(lambda: (kernels.bincount(lab, k, counts), kernels.segment_sum(x, lab, k, sums)), 50)
print('bincount %.1f us, segment_sum %.1f us, both %.1f us = %.0f GB/s (4 n d bytes)' % (tb * 1e3, ts * 1e3, tt * 1e3, 4.0 * n * d / tt / 1e6))
