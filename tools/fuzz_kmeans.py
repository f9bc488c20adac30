"""Random (n, k, d, dtypes, scales, row strides, duplicate centres, points on centres) for sp_nearest_center against
argmin(cdist) in fp64.  Usage: python tools/fuzz_kmeans.py [seed]"""
import sys, time
import numpy as np
from scipy.spatial.distance import cdist
from _dev import D, kernels
from spartan_amd import _hip
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
t0 = time.time()
for it in range(400):
  n = int(rng.choice([1, 5, 127, 128, 129, 1000, 4096, 20000, 66000, 70000]))
  k = int(rng.choice([1, 2, 17, 255, 256, 257, 512, 600, 1025, 2304]))
  d = int(rng.choice([1, 3, 8, 16, 31, 32, 33, 64, 100, 256]))
  if n * k * d > 3e9: continue
  xdt = rng.choice([np.float32, np.float64]); cdt = rng.choice([np.float32, np.float64])
  scale = float(rng.choice([1.0, 1e-3, 1e3]))
  x = (rng.rand(n, d) * scale).astype(xdt); c = (rng.rand(k, d) * scale).astype(cdt)
  if k > 3 and rng.rand() < 0.5: c[k - 1] = c[0]
  m = min(n, k)
  if rng.rand() < 0.5: x[:m] = c[:m].astype(xdt)
  pad = int(rng.choice([0, 0, 4, 7]))
  xt = D.from_numpy(np.pad(x, ((0, 0), (0, pad))))[:, :d]
  ct = D.from_numpy(c)
  lab = D.empty((n,), np.int64)
  tier = int(rng.choice([_hip.NEAREST_AUTO, _hip.NEAREST_FUSED, _hip.NEAREST_SPLIT, _hip.NEAREST_SPLIT]))
  try:
    kernels.nearest_center(xt, ct, lab, tier)
  except Exception as e:
    if xdt == np.float64 and tier in (_hip.NEAREST_FUSED, _hip.NEAREST_SPLIT): continue   # the MFMA tiers are fp32 points only
    print('EXC', n, k, d, xdt, cdt, tier, e); bad += 1; continue
  want = np.argmin(cdist(x.astype(np.float64), c.astype(np.float64)), axis=1)
  got = lab.numpy()
  if not np.array_equal(got, want):
    # ties in exact fp64 arithmetic may differ from cdist's rounding: accept only equal distances
    dd = cdist(x.astype(np.float64), c.astype(np.float64))
    diff = np.nonzero(got != want)[0]
    real = [i for i in diff if dd[i, got[i]] != dd[i, want[i]]]
    if real:
      print('MISMATCH', n, k, d, xdt, cdt, tier, pad, scale, len(real), real[:3]); bad += 1
print('done', it + 1, 'cases', bad, 'bad', round(time.time() - t0, 1), 's')
