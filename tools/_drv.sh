cd $GRAFT_REPO_ROOT
python - <<'PY'
import cProfile, pstats, io, sys, os
import numpy as np
sys.path.insert(0, '.')
import bench
import spartan_amd as sp
from spartan_amd import devarray as D
from spartan_amd.examples import lreg
ctx = sp.initialize('hip')
N, Dm = 125000, 4096
X = sp.Val(val=sp.from_tile_fn((N, Dm), np.float32, lambda ex: bench.device_uniform(ex, 0.0, 1.0, 11)).force())
y = sp.Val(val=sp.from_tile_fn((N, 1), np.float32, lambda ex: bench.device_uniform(ex, 0.0, 1.0, 12)).force())
w = np.random.RandomState(0).rand(Dm, 1).astype(np.float32)
w = lreg.fit(X, y, 5, alpha=1e-10, w=w)
pr = cProfile.Profile(); pr.enable()
w = lreg.fit(X, y, 200, alpha=1e-10, w=w)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(40); print(s.getvalue()[:7000])
PY
