"""Host-path timing WITHOUT a GPU: HostStorage tiles, launch errors ignored (the Python side is what is measured)."""
import cProfile, pstats, sys, os, time, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import spartan_amd as sp
from spartan_amd import _hip, devarray, kernels, sparse, context as _context, jit_seed
from spartan_amd.expr import base as _base
for m in (_hip, kernels, devarray, sparse):
  m.check = lambda rc: None
class FastHost(devarray.HostStorage):
  __slots__ = ()
  _pool = {}
  def __init__(self, nbytes):
    nbytes = max(int(nbytes), 1)
    buf = FastHost._pool.get(nbytes)
    if buf is None:
      buf = FastHost._pool[nbytes] = np.empty(nbytes, np.uint8)
    self.buf = buf
    self.ptr = buf.ctypes.data
    self.nbytes = nbytes
devarray._storage_cls[0] = FastHost
class B(jit_seed._SeedBackend):
  def _lowering_key(self, op, inputs, ex, extra):
    return jit_seed.HipBackend._lowering_key(self, op, inputs, ex, extra)
_context.set(_context.Context(B(), None, 1))
shape = (1024, 4096)
X = sp.Val(val=sp.from_tile_fn(shape, np.float32, lambda ex: devarray.empty(ex.shape, np.float32)).force())
progs = {'x_plus_1': lambda: (X + 1).force(),
         'xx_plus_x': lambda: (X * X + X).optimized().force(),
         'chain5': lambda: (((X * X + X) * 0.5 - X) / (X + 2.0)).optimized().force(),
         'chain5_build': lambda: (((X * X + X) * 0.5 - X) / (X + 2.0)),
         'sum0': lambda: sp.sum(X, 0).force()}
which = [a for a in sys.argv[1:] if a in progs] if sys.argv[1:] else list(progs)
for name in which:
  fn = progs[name]
  for _ in range(50): fn()
  best = 1e9
  for _ in range(5):
    t0 = time.perf_counter()
    for _ in range(300): fn()
    best = min(best, (time.perf_counter() - t0) / 300 * 1e6)
  print('%s: %.1f us' % (name, best))
  if os.environ.get('PROF'):
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(500): fn()
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(os.environ.get('SORT', 'tottime')).print_stats(40)
    print('\n'.join(l[:160] for l in s.getvalue().splitlines()[:60]))
if os.environ.get('PR'):
  from spartan_amd import sparse as S
  n = 900000
  rows = devarray.empty((n * 10,), np.int32); cols = devarray.empty((n * 10,), np.int32); vals = devarray.empty((n * 10,), np.float32)
  try:
    W = S.from_coo((n, n), np.float32, rows, cols, vals)
  except Exception as e:
    print('from_coo failed', type(e), e); raise
  x = devarray.empty((n, 1), np.float32)
  wts = sp.from_tile_fn((n, n), np.float32, lambda ex: W, sparse=True).force()
  p = sp.from_tile_fn((n, 1), np.float32, lambda ex: x).force()
  def five():
    q = sp.Val(val=p)
    for _ in range(5):
      q = sp.dot(sp.Val(val=wts), q).optimized()
    q.force()
  for _ in range(20): five()
  best = 1e9
  for _ in range(5):
    t0 = time.perf_counter()
    for _ in range(100): five()
    best = min(best, (time.perf_counter() - t0) / 100 * 1e6)
  print('five iterations: %.1f us' % best)
  if os.environ.get('PROF'):
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(200): five()
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(os.environ.get('SORT', 'tottime')).print_stats(45)
    print('\n'.join(l[:160] for l in s.getvalue().splitlines()[:60]))
if os.environ.get('PR'):
  y = devarray.empty((n, 1), np.float32)
  for _ in range(50): S.spmm(W, x, out=y)
  t0 = time.perf_counter()
  for _ in range(1000): S.spmm(W, x, out=y)
  print('spmm direct: %.1f us' % ((time.perf_counter() - t0) / 1000 * 1e6))
  def one_eval():
    sp.dot(sp.Val(val=wts), sp.Val(val=p)).force()
  def one_opt():
    sp.dot(sp.Val(val=wts), sp.Val(val=p)).optimized().force()
  for f in (one_eval, one_opt):
    for _ in range(50): f()
    t0 = time.perf_counter()
    for _ in range(500): f()
    print('%s: %.1f us' % (f.__name__, (time.perf_counter() - t0) / 500 * 1e6))
if os.environ.get('PR') and os.environ.get('PROF2'):
  f = one_eval if os.environ['PROF2'] == 'eval' else one_opt
  pr = cProfile.Profile()
  pr.enable()
  for _ in range(500): f()
  pr.disable()
  s = io.StringIO()
  pstats.Stats(pr, stream=s).sort_stats(os.environ.get('SORT', 'tottime')).print_stats(70)
  print('\n'.join(l[:150] for l in s.getvalue().splitlines()[:90]))
