#!/usr/bin/env python
"""Regenerates the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

This script only works in the build container (it reads /root/reference, which
does not exist on the GPU box); its OUTPUT -- small .npz / .json fixtures of
inputs and expected outputs -- is what is committed and what the tests read.
No reference source is copied into the repository: the reference tree is
transliterated into a scratch directory under /tmp, imported from there, and
thrown away.

The reference is Python 2.7 + Cython; to import its tile path under
Python 3.10 the script
  1. copies /root/reference/spartan to a scratch dir and runs lib2to3 on it
     (without the map/filter/reduce/zip/import fixers: Spartan defines its own
     map/reduce);
  2. fixes the handful of Python-2 semantics lib2to3 cannot see (integer `/`,
     `None > int` comparisons, implicit relative imports, removed numpy aliases);
  3. cythonizes the reference's own extent.pyx / tile.pyx / core.pyx /
     sparse.pyx / rpc/rlock.pyx (language_level=2);
  4. provides import-time stand-ins for packages that are not installable here
     and are OFF the tile path (traits, appdirs, zmq, parakeet, the CPython-2
     tiling extension), and an in-process replacement of spartan.rpc so that N
     Worker objects run in this process (the reference's own test fixture also
     runs its workers in-process, tests/test_common.py:128-136);
  5. runs the tile-path programs with 1, 3, 4 and 8 workers and records inputs
     and outputs.

Modes (each rewrites its own files; all of them are deterministic -- a second run leaves `git status` clean):
  (none)      extent / merge / fusion known answers, the programs of tests/programs.py at 1, 3, 4, 8 workers
  --examples  the k-means / regression drivers (examples_w{1,3,4,8}.npz, examples_inputs.npz)
  --sparse    the sparse-tile programs (sparse_w{1,3,4,8}.npz, sparse_meta.json)
  --joins     map2 / outer / shuffle with user tile functions, tests/join_programs.py (joins_w{1,3,4,8}.npz)
  --fuzz      300 random expression DAGs and 100 random dots of tests/test_fuzz_gpu.py's generators, as the reference
              computes them (fuzz_w{1,3,4,8}.npz, fuzz_meta.json)
  --region    map2(update_region=...), 4 workers (region_w4.npz)
  --dotgrid   spartan.dot on grid-tiled operands (dot_grid.npz)
  --tiling    the reference's tiling.cc on the cost graphs of every program (tiling_golden.json)

Everything the reference computes here is computed by the reference's own code:
extent.pyx, tile.pyx (merge), distarray.py (tiling, fetch, update), map/reduce/
dot/outer mappers, optimize.py (fusion), sorting.py (argmax/argmin).
"""
import importlib
import json
import os
import shutil
import subprocess
import sys
import types

import numpy as np

REF = '/root/reference'
SCRATCH = '/tmp/spartan_ref_build'
OUT = os.path.dirname(os.path.abspath(__file__))


def sub(path, pairs):
  s = open(path).read()
  for a, b in pairs:
    if a not in s:
      raise SystemExit('patch target not found in %s: %r' % (path, a))
    s = s.replace(a, b)
  open(path, 'w').write(s)


def prepare_tree():
  if os.path.exists(SCRATCH):
    shutil.rmtree(SCRATCH)
  os.makedirs(SCRATCH)
  shutil.copytree(os.path.join(REF, 'spartan'), os.path.join(SCRATCH, 'spartan'))
  os.chdir(SCRATCH)
  files = []
  for d in ('spartan', 'spartan/expr', 'spartan/expr/operator', 'spartan/array'):
    files += [os.path.join(d, f) for f in os.listdir(d) if f.endswith('.py')]
  subprocess.check_call([sys.executable, '-m', 'lib2to3', '-w', '-n', '-x', 'map', '-x', 'filter',
                         '-x', 'reduce', '-x', 'zip', '-x', 'import'] + files,
                        stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
  S = 'spartan/'
  # -- Python-2 integer division / None comparisons
  sub(S + 'array/extent.pyx', [
      ("    idx /= dim", "    idx //= dim"),
      ("ravelled_ul / shape[-1] == ravelled_lr / shape[-1]", "ravelled_ul // shape[-1] == ravelled_lr // shape[-1]"),
      ("original_index = int(ex.ul[old_axis] / step)", "original_index = int(ex.ul[old_axis] // step)"),
      ("        original_index /= n", "        original_index //= n"),
      ("blk_idx = (ex.ul[0]/ex.shape[0]) * util.divup(ex.array_shape[1], ex.shape[1]) + ex.ul[1]/ex.shape[1]",
       "blk_idx = (ex.ul[0]//ex.shape[0]) * util.divup(ex.array_shape[1], ex.shape[1]) + ex.ul[1]//ex.shape[1]"),
      ("    if slc.start > 0: assert slc.start <= dim", "    if slc.start is not None and slc.start > 0: assert slc.start <= dim"),
      ("    if slc.stop > 0: assert slc.stop <= dim", "    if slc.stop is not None and slc.stop > 0: assert slc.stop <= dim"),
      ("    if slice.start > 0: return False", "    if slice.start is not None and slice.start > 0: return False"),
      ("    if slice.stop < dim: return False", "    if slice.stop is not None and slice.stop < dim: return False"),
      ("xrange", "range"),
  ])
  sub(S + 'array/tile.pyx', [("ID = iter(xrange(100000000))", "ID = iter(range(100000000))"),
                             ("self.id = ID.next()", "self.id = next(ID)"), ("np.bool)", "np.bool_)")])
  sub(S + 'core.pyx', [("from node import Node", "from spartan.node import Node"),
                       ("raise Exception, 'WTF'", "raise Exception('WTF')")])
  sub(S + 'array/distarray.py', [
      ("    tile_size = np.prod(shape) / num_shards", "    tile_size = np.prod(shape) // num_shards"),
      ("    tile_size /= shape[idx]", "    tile_size //= shape[idx]"),
      ("      if (i / ctx.num_workers) % 2 == 1:", "      if (i // ctx.num_workers) % 2 == 1:"),
      ("dtype=np.float,", "dtype=float,"), ("    dtype = np.float\n", "    dtype = float\n"),
      ("Assert.isinstance(data, (np.ndarray, int, int, float))", "Assert.isinstance(data, (np.ndarray, int, float, np.generic))"),
  ])
  # Python scalars must stay "weak" operands, as under the NumPy 1.x value-based
  # casting the reference was written for (SURVEY 8c deviations (i),(iii)):
  # LocalWrapper hands the Python scalar itself to the ufunc instead of a 0-d array.
  sub(S + 'array/distarray.py', [
      ("    self._data = np.asarray(data)\n    self.sparse = False",
       "    self._data = np.asarray(data)\n    self._py = data if type(data) in (int, float, bool) else None\n    self.sparse = False"),
      ("  def fetch(self, ex):\n    return self._data[ex.to_slice()]",
       "  def fetch(self, ex):\n    if self._py is not None: return self._py\n    return self._data[ex.to_slice()]"),
  ])
  sub(S + '__init__.py', [("import core", "from . import core")])
  sub(S + 'expr/__init__.py', [("import mathematics", "from . import mathematics")])
  sub(S + 'expr/operator/ndarray.py', [("dtype=np.float,", "dtype=float,")])
  sub(S + 'expr/creation.py', [("dtype=np.float,", "dtype=float,"),
                               # a LIST of slices as an index meant the tuple of them until NumPy 1.23
                               ("result = tile[slices].diagonal()", "result = tile[tuple(slices)].diagonal()")])
  # `a / b` on expressions: only __div__/__rdiv__ exist (base.py:348-349,387-388)
  sub(S + 'expr/operator/base.py', [
      ("  def __eq__(self, other):\n    return _map(self, other, fn=np.equal)",
       "  __truediv__ = __div__\n\n  def __eq__(self, other):\n    return _map(self, other, fn=np.equal)"),
      ("  def reshape(self, new_shape):", "  __rtruediv__ = __rdiv__\n\n  def reshape(self, new_shape):"),
  ])
  # np.bool / np.int / np.float aliases used at module level elsewhere
  for root, _, fs in os.walk(S):
    for f in fs:
      if f.endswith(('.py', '.pyx')):
        p = os.path.join(root, f)
        s = open(p).read()
        s2 = s.replace('__builtin__', 'builtins')
        if s2 != s:
          open(p, 'w').write(s2)


def build_cython():
  from Cython.Build import cythonize
  from setuptools import Extension, setup
  os.chdir(SCRATCH)
  exts = [
      Extension('spartan.array.extent', ['spartan/array/extent.pyx'], include_dirs=[np.get_include()]),
      Extension('spartan.array.tile', ['spartan/array/tile.pyx'], include_dirs=[np.get_include()]),
      Extension('spartan.core', ['spartan/core.pyx'], include_dirs=[np.get_include()]),
      Extension('spartan.array.sparse', ['spartan/array/sparse.pyx'], include_dirs=[np.get_include()],
                language='c++', extra_compile_args=['-std=c++11']),
      Extension('spartan.rpc.rlock', ['spartan/rpc/rlock.pyx']),
  ]
  setup(script_args=['build_ext', '--inplace', '-q'],
        ext_modules=cythonize(exts, language_level=2, quiet=True,
                              compiler_directives={'always_allow_keywords': True}))


def install_stubs():
  """Stand-ins for uninstallable packages that are OFF the tile path."""
  # traits.api: attribute declarations with defaults; Node relies on __base_traits__
  traits = types.ModuleType('traits')
  api = types.ModuleType('traits.api')
  tr = types.ModuleType('traits.traits')

  class CTrait(object):
    _factory = staticmethod(lambda: None)

    def __init__(self, *args, **kw):
      if args and not isinstance(args[0], type):
        d = args[0]
        self.default = lambda d=d: d
      else:
        self.default = self._factory
      self.kw = kw

  def _mk(name, factory):
    return type(name, (CTrait,), {'_factory': staticmethod(factory)})

  class MetaHasTraits(type):
    def __new__(mcs, name, bases, ns):
      base_traits = {}
      for b in bases:
        base_traits.update(getattr(b, '__base_traits__', {}))
      for k, v in list(ns.items()):
        if isinstance(v, type) and issubclass(v, CTrait):
          v = v()               # `val = PythonValue` (uncalled) is a valid trait declaration
        if isinstance(v, CTrait):
          base_traits[k] = v
          ns[k] = None          # shadows inherited properties (Map2Expr.shape vs Expr.shape), like a real trait
      ns['__base_traits__'] = base_traits
      return type.__new__(mcs, name, bases, ns)

  class HasTraits(object, metaclass=MetaHasTraits):
    def __init__(self, *args, **kw):
      for k, v in self.__base_traits__.items():
        if k not in kw:
          object.__setattr__(self, k, v.default())
      for k, v in kw.items():
        object.__setattr__(self, k, v)

  api.HasTraits = HasTraits
  api.HasStrictTraits = HasTraits
  for nm, fac in [('Any', lambda: None), ('PythonValue', lambda: None), ('Instance', lambda: None),
                  ('Function', lambda: None), ('Int', lambda: 0), ('Float', lambda: 0.0), ('Str', lambda: ''),
                  ('Bool', lambda: False), ('List', list), ('Dict', dict), ('Tuple', tuple),
                  ('Trait', lambda: None)]:
    setattr(api, nm, _mk(nm, fac))
  tr.CTrait = CTrait
  traits.api = api
  traits.traits = tr
  sys.modules.update({'traits': traits, 'traits.api': api, 'traits.traits': tr})

  appdirs = types.ModuleType('appdirs')
  appdirs.user_config_dir = lambda *a, **k: '/tmp/spartan_ref_build/cfg'
  appdirs.user_data_dir = lambda *a, **k: '/tmp/spartan_ref_build/cfg'
  sys.modules['appdirs'] = appdirs
  zmq = types.ModuleType('zmq')
  sys.modules['zmq'] = zmq
  parakeet = types.ModuleType('parakeet')
  parakeet.jit = lambda f: f
  sys.modules['parakeet'] = parakeet
  np.float = float
  np.int = int


FAKE_RPC = '''
"""In-process replacement of spartan.rpc for golden generation: every RPC is a
direct call on the target Worker object (the reference's data path -- worker
handlers, Tile.merge, kernels -- runs unmodified)."""
class TimeoutException(Exception): pass
class RemoteException(Exception): pass
class RPCException(object):
  def __init__(self, py_exc=None): self.py_exc = py_exc

class Future(object):
  def __init__(self, addr=None, rpc_id=-1, result=None): self.result = result
  def done(self, result=None): self.result = result
  def exception(self):
    import traceback
    traceback.print_exc()
    raise
  def wait(self): return self.result

class FutureGroup(list):
  def wait(self): return [f.wait() for f in self]

def wait_for_all(futures): return [f.wait() for f in futures]

class PendingRequest(object):
  def __init__(self): self.result = None; self.exc = None
  def done(self, result=None): self.result = result
  def exception(self):
    import sys, traceback
    self.exc = sys.exc_info()
    traceback.print_exc()
    raise

def forall(clients, method, request, timeout=None):
  return FutureGroup([c.call(method, request) for c in clients])

class DirectClient(object):
  """Calls the worker's RPC handler synchronously."""
  def __init__(self, worker): self.worker = worker
  def call(self, method, request):
    h = PendingRequest()
    getattr(self.worker, method)(request, h)
    return Future(result=h.result)
  def __getattr__(self, method):
    if method == 'shutdown':
      return lambda *a, **k: Future()
    return lambda request=None, timeout=None: self.call(method, request)
  def addr(self): return ('local', 0)

def listen(*a, **k): raise NotImplementedError
def connect(*a, **k): raise NotImplementedError
def set_default_timeout(*a): pass
from . import rlock
'''


def import_reference():
  sys.path.insert(0, SCRATCH)
  # replace spartan/rpc with the in-process fake
  rpc_dir = os.path.join(SCRATCH, 'spartan', 'rpc')
  for f in os.listdir(rpc_dir):
    if f.endswith('.py'):
      os.remove(os.path.join(rpc_dir, f))
  open(os.path.join(rpc_dir, '__init__.py'), 'w').write(FAKE_RPC)
  open(os.path.join(rpc_dir, 'zeromq.py'), 'w').write('')
  # CPython-2 tiling extension: off the path, unless --tiling built it for this interpreter (build_tiling_ext)
  opdir = os.path.join(SCRATCH, 'spartan', 'expr', 'operator')
  if not [f for f in os.listdir(opdir) if f.startswith('tiling.') and f.endswith('.so')]:
    open(os.path.join(opdir, 'tiling.py'), 'w').write(
        'def mincost_tiling(*a, **k): raise NotImplementedError\n'
        'def maxedge_tiling(*a, **k): raise NotImplementedError\n'
        'def best_tiling(*a, **k): raise NotImplementedError\n'
        'def worse_tiling(*a, **k): raise NotImplementedError\n')
  os.makedirs(os.path.join(SCRATCH, 'cfg'), exist_ok=True)
  sys.argv = [sys.argv[0]]
  import spartan  # noqa
  return spartan



def start_cluster(sp, n):
  """N Worker objects + a Master in this process, wired with direct calls."""
  import weakref
  from spartan import blob_ctx, config, core, master, rpc, worker
  from spartan.config import FLAGS
  from spartan.rpc import rlock
  from spartan import util
  for name in ('log_debug', 'log_info', 'log_warn', 'log_error'):
    # util._setup_logger overrides findCaller with a Python-2 signature (util.py:29-35)
    setattr(util, name, lambda *a, **k: None)
  if not FLAGS._parsed:
    config.parse([])
  FLAGS.opt_auto_tiling = False
  FLAGS.opt_parakeet_gen = False
  FLAGS.log_level = 40
  master.Master.__del__ = lambda self: None   # (its shutdown path needs the sockets we never opened)
  m = master.Master.__new__(master.Master)
  m._workers = {}
  m.num_workers = n
  m._initialized = True
  m._worker_statuses = {}
  m._worker_scores = {}
  m._available_workers = list(range(n))
  m._arrays = weakref.WeakSet()
  master.MASTER = m
  workers = []
  for i in range(n):
    w = worker.Worker.__new__(worker.Worker)
    w.id = i
    w._initialized = True
    w._blobs = {}
    w._master = m
    w._running = True
    w._lock = rlock.FastRLock()
    w._kernel_remain_tiles = []
    w.worker_status = core.WorkerStatus(1 << 30, 1, 0.0, 0.0, 0.0, [], [])

    def run_kernel(req, handle, w=w):
      prev = blob_ctx.get()
      try:
        w._run_kernel(req, handle)   # the reference runs this on the worker's kernel thread
      finally:
        blob_ctx.set(prev)
    w.run_kernel = run_kernel
    workers.append(w)
  clients = dict((i, rpc.DirectClient(workers[i])) for i in range(n))
  for w in workers:
    w._peers = dict(clients)
    w._peers[blob_ctx.MASTER_ID] = rpc.DirectClient(m)
    w._ctx = blob_ctx.BlobCtx(w.id, w._peers, w)
  m._workers = clients
  m._ctx = blob_ctx.BlobCtx(blob_ctx.MASTER_ID, clients, m)
  blob_ctx.set(m._ctx)
  # fresh evaluation cache per cluster
  from spartan.expr.operator import base
  base.eval_cache.clear()
  return m, workers


# ----------------------------------------------------------------------------
def _ex_tuple(ex):
  return None if ex is None else [list(ex.ul), list(ex.lr), None if ex.array_shape is None else list(ex.array_shape)]


def _slices(t):
  return [[s.start, s.stop] for s in t]


def extent_goldens():
  """Known answers of the reference's extent.pyx (compiled from its own source)."""
  from spartan.array import distarray, extent
  import random
  rnd = random.Random(20150708)
  G = {}

  def boxes(shape, n):
    out = []
    for _ in range(n):
      ul = [rnd.randint(0, d - 1) for d in shape]
      lr = [rnd.randint(u + 1, d) for u, d in zip(ul, shape)]
      out.append((tuple(ul), tuple(lr)))
    return out

  cases = []
  for shape in [(20, 77), (16, 16), (5, 6, 7), (100,)]:
    bs = boxes(shape, 24)
    for (a, b) in zip(bs[::2], bs[1::2]):
      ea = extent.create(a[0], a[1], shape)
      eb = extent.create(b[0], b[1], shape)
      inter = extent.intersection(ea, eb)
      rec = {'shape': list(shape), 'a': _ex_tuple(ea), 'b': _ex_tuple(eb), 'intersection': _ex_tuple(inter)}
      if inter is not None:
        rec['offset_slice_a'] = _slices(extent.offset_slice(ea, inter))
        rec['offset_from_a'] = _ex_tuple(extent.offset_from(ea, inter))
      rec['ravelled_pos_a'] = int(ea.ravelled_pos())
      rec['unravelled'] = list(extent.unravelled_pos(ea.ravelled_pos(), shape))
      rec['a_shape'] = list(ea.shape)
      rec['a_size'] = int(ea.size)
      rec['to_global_5_none'] = int(ea.to_global(min(5, ea.size - 1), None))
      rec['to_global_3_axis0'] = int(ea.to_global(3, 0))
      for axis in [None] + list(range(len(shape))):
        d = extent.drop_axis(ea, axis)
        rec['drop_axis_%s' % axis] = _ex_tuple(d)
        rec['shape_for_reduction_%s' % axis] = list(extent.shape_for_reduction(shape, axis))
      cases.append(rec)
  G['pairs'] = cases
  # touching / degenerate boxes
  a = extent.create((0, 0), (5, 5), (10, 10))
  b = extent.create((5, 0), (10, 5), (10, 10))
  G['touching_intersection'] = _ex_tuple(extent.intersection(a, b))
  G['degenerate_create'] = _ex_tuple(extent.create((5, 5), (5, 5), (10, 10)))
  # from_slice / compute_slice
  fs = []
  for shape, idx in [((10, 12), np.index_exp[2:5]), ((10, 12), np.index_exp[:, 3:9]), ((10, 12), np.index_exp[4]),
                     ((10, 12), np.index_exp[-3:, :-2]), ((7,), np.index_exp[:]), ((4, 5, 6), np.index_exp[1:3, :, 2:4])]:
    ex = extent.from_slice(idx, shape)
    rec = {'shape': list(shape), 'idx': [[i.start, i.stop] if isinstance(i, slice) else int(i) for i in idx],
           'from_slice': _ex_tuple(ex)}
    fs.append(rec)
  G['from_slice'] = fs
  base = extent.create((2, 3), (8, 11), (10, 12))
  cs = []
  for idx in [np.index_exp[1:3], np.index_exp[:, 2:5], np.index_exp[0], np.index_exp[-2:, -3:]]:
    cs.append({'idx': [[i.start, i.stop] if isinstance(i, slice) else int(i) for i in idx],
               'result': _ex_tuple(extent.compute_slice(base, idx))})
  G['compute_slice'] = {'base': _ex_tuple(base), 'cases': cs}
  # change_partition_axis: 1-D re-partition, vector, grid<->1-D
  cpa = []
  for shape, nshards in [((32768, 32768), 8), ((8192, 8192), 4), ((100, 37), 3), ((64, 128), 8), ((1000, 10), 4)]:
    for ex in distarray.compute_extents(shape, None, nshards):
      for axis in (0, 1, -1):
        cpa.append({'ex': _ex_tuple(ex), 'axis': axis, 'result': _ex_tuple(extent.change_partition_axis(ex, axis))})
  for ex in distarray.compute_extents((64, 64), (16, 64), 4):
    cpa.append({'ex': _ex_tuple(ex), 'axis': [0, 1], 'result': _ex_tuple(extent.change_partition_axis(ex, (0, 1)))})
  for ex in distarray.compute_extents((64, 64), (32, 32), 4):
    for axis in (0, 1):
      cpa.append({'ex': _ex_tuple(ex), 'axis': axis, 'result': _ex_tuple(extent.change_partition_axis(ex, axis))})
  for ex in distarray.compute_extents((100,), None, 4):
    for axis in (0, 1):
      cpa.append({'ex': _ex_tuple(ex), 'axis': axis, 'result': _ex_tuple(extent.change_partition_axis(ex, axis))})
  G['change_partition_axis'] = cpa
  # tiling of every BASELINE shape (and the reference tests' shapes) at 1..8 shards
  til = []
  shapes = [(1000, 1000), (8192, 8192), (32768, 32768), (65536, 65536), (10000000, 256), (1000000, 4096),
            (1000000, 1), (4096, 1), (65536,), (4096,), (1024, 256), (100, 37), (137, 33), (11, 12, 13), (40, 30),
            (7,), (1, 1), ()]
  for shape in shapes:
    for n in (1, 2, 3, 4, 8):
      exts = distarray.compute_extents(shape, None, n)
      til.append({'shape': list(shape), 'num_shards': n,
                  'good_tile_shape': [int(v) for v in distarray.good_tile_shape(shape, n)] if len(shape) else [],
                  'extents': [[_ex_tuple(ex), int(i)] for ex, i in exts.items()]})
  for shape, hint, n in [((32768, 32768), (4096, 32768), 8), ((64, 64), (16, 16), 4), ((100, 10), (30, 10), 3)]:
    exts = distarray.compute_extents(shape, hint, n)
    til.append({'shape': list(shape), 'num_shards': n, 'tile_hint': list(hint),
                'extents': [[_ex_tuple(ex), int(i)] for ex, i in exts.items()]})
  G['tiling'] = til
  G['find_rect'] = [{'args': [ul, lr, list(shape)], 'result': [int(v) for v in extent.find_rect(ul, lr, shape)]}
                    for ul, lr, shape in [(3, 9, (4, 5)), (5, 9, (4, 5)), (0, 19, (4, 5)), (7, 8, (10, 1)), (13, 47, (3, 4, 5))]]
  G['find_shape'] = [int(v) for v in extent.find_shape(list(distarray.compute_extents((100, 37), None, 3).keys()))]
  G['is_complete'] = [bool(extent.is_complete((4, 5), (slice(0, 4), slice(0, 5)))),
                      bool(extent.is_complete((4, 5), (slice(0, 4), slice(1, 5)))),
                      bool(extent.is_complete((4, 5), (slice(None, None), slice(None, None))))]
  return G


def merge_goldens():
  """Truth table of the reference's Tile.merge (tile.pyx:200-297), dense branch."""
  from spartan.array import tile
  rng = np.random.RandomState(42)
  out = {}
  seqs = {
      'full_first_add': [('full', 'add')],
      'full_twice_add': [('full', 'add'), ('full', 'add')],
      'full_twice_none': [('full', None), ('full', None)],
      'full_max_min': [('full', 'maximum'), ('full', 'maximum'), ('full', 'minimum')],
      'sub_first_add': [((1, 4, 2, 6), 'add')],
      'sub_overlap_add': [((1, 4, 2, 6), 'add'), ((2, 6, 4, 8), 'add')],
      'sub_overlap_none': [((1, 4, 2, 6), None), ((2, 6, 4, 8), None)],
      'sub_then_full': [((0, 3, 0, 8), 'add'), ('full', 'add')],
      'full_then_sub': [('full', 'add'), ((2, 5, 1, 7), 'add'), ((2, 5, 1, 7), 'multiply')],
      'rows_disjoint': [((0, 2, 0, 8), 'add'), ((2, 4, 0, 8), 'add'), ((4, 6, 0, 8), 'add')],
  }
  red = {'add': np.add, 'maximum': np.maximum, 'minimum': np.minimum, 'multiply': np.multiply, None: None}
  for name, seq in seqs.items():
    t = tile.from_shape((6, 8), np.float32, tile.TYPE_DENSE)
    steps = []
    for box, r in seq:
      if box == 'full':
        upd = rng.randint(-4, 5, size=(6, 8)).astype(np.float32)
        sl = tuple(slice(0, n) for n in (6, 8))
        b = [0, 6, 0, 8]
      else:
        r0, r1, c0, c1 = box
        upd = rng.randint(-4, 5, size=(r1 - r0, c1 - c0)).astype(np.float32)
        sl = (slice(r0, r1), slice(c0, c1))
        b = list(box)
      t = t.update(sl, upd, red[r])
      steps.append({'box': b, 'reducer': r, 'update': upd.tolist()})
    mask = t.mask if isinstance(t.mask, np.ndarray) else np.full((6, 8), bool(t.mask))
    # Tile.get of the whole tile and of a box: a MaskedArray where some cell was never written (tile.pyx:100-113)
    reads = []
    for box in ((0, 6, 0, 8), (1, 5, 3, 8)):
      got = t.get((slice(box[0], box[1]), slice(box[2], box[3])))
      masked = isinstance(got, np.ma.MaskedArray)
      reads.append({'box': list(box), 'masked': bool(masked),
                    'values': np.asarray(got.filled(0) if masked else got).tolist(),
                    'valid': (~np.ma.getmaskarray(got)).astype(int).tolist()})
    out[name] = {'steps': steps, 'data': np.asarray(t.data).tolist(), 'mask': np.asarray(mask).astype(int).tolist(),
                 'reads': reads}
  # zero-dimensional tile (tile.pyx:212-217)
  t = tile.from_shape((), np.float32, tile.TYPE_DENSE)
  t = t.update(None, np.float32(3.0), np.add)
  v1 = float(t.data)
  t = t.update(None, np.float32(4.5), np.add)
  out['zero_dim'] = {'first': v1, 'second': float(t.data)}
  return out


def program_goldens(sp, workers):
  sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
  from tests import programs
  arrays = {}
  meta = {}
  for name, build, expected, tol in programs.programs():
    start_cluster(sp, workers)
    try:
      expr = build(sp)
      res = expr.evaluate() if hasattr(expr, 'evaluate') else expr
      val = np.asarray(res.glom())
    except Exception as e:  # programs the reference itself cannot run
      meta[name] = {'skipped': '%s: %s' % (type(e).__name__, str(e)[:200])}
      continue
    tiles = None
    if hasattr(res, 'tiles'):
      tiles = sorted([[_ex_tuple(ex), int(tid.worker)] for ex, tid in res.tiles.items()])
    m = {'dtype': val.dtype.str, 'shape': list(val.shape), 'tiles': tiles}
    if val.size <= 20000:
      arrays[name] = val
    else:
      import hashlib
      m['sha1'] = hashlib.sha1(np.ascontiguousarray(val).tobytes()).hexdigest()
      m['sum'] = float(val.astype(np.float64).sum())
    meta[name] = m
  return arrays, meta


def fusion_goldens(sp):
  """Fused LocalExpr trees (pretty strings, local.py:94-100) the optimiser builds."""
  start_cluster(sp, 1)
  out = {}
  e = (sp.ones((4, 4)) + sp.ones((4, 4)) + sp.ones((4, 4)) + sp.ones((4, 4))).optimized()
  out['add_many'] = e.op.pretty_str()
  a = sp.ones((4, 4))
  r = sp.sum(a * a + a, axis=0).optimized()
  out['sum_mul_add'] = r.op.pretty_str()
  out['sum_mul_add_nchildren'] = len(r.children)
  return out

def build_tiling_ext():
  """The reference's solver (spartan/expr/operator/tiling.cc: mincost / maxedge / best / worse tiling) as an
  extension of THIS interpreter.  The source is compiled from where it lies, included by a three-line translation
  unit written to the scratch directory that renames the two Python-2 C-API calls it makes (PyInt_AsLong,
  Py_InitModule) and supplies a Python-3 module definition for its own method table -- the same kind of
  transliteration lib2to3 does for the .py files.  Nothing of it is stored in the repository."""
  import sysconfig
  bdir = os.path.join(SCRATCH, 'tiling_build')
  os.makedirs(bdir, exist_ok=True)
  tu = os.path.join(bdir, 'tiling_py3.cc')
  open(tu, 'w').write(
      '#include <Python.h>\n'
      '#define PyInt_AsLong PyLong_AsLong\n'
      '#undef PyMODINIT_FUNC\n'
      '#define PyMODINIT_FUNC static void\n'
      '#define Py_InitModule(name, methods) ((PyObject*)(methods))\n'
      '#include "%s"\n'
      'static struct PyModuleDef sp_tiling_def = {PyModuleDef_HEAD_INIT, "tiling", NULL, -1, TilingMethods};\n'
      'extern "C" PyObject* PyInit_tiling(void) { (void)inittiling; return PyModule_Create(&sp_tiling_def); }\n'
      % os.path.join(REF, 'spartan', 'expr', 'operator', 'tiling.cc'))
  opdir = os.path.join(SCRATCH, 'spartan', 'expr', 'operator')
  out = os.path.join(opdir, 'tiling' + sysconfig.get_config_var('EXT_SUFFIX'))
  subprocess.check_call(['g++', '-O2', '-shared', '-fPIC', '-std=c++11', '-w', '-I' + sysconfig.get_paths()['include'],
                         tu, '-o', out])
  stub = os.path.join(opdir, 'tiling.py')
  if os.path.exists(stub):
    os.remove(stub)
  return out


def tiling_goldens(sp):
  """Cost graphs the reference's AutomaticTiling pass builds for the shared test programs (tests/programs.py), as
  it hands them to its solver -- (number of nodes, edges (u, v, cost), groups of four alternatives) -- with what
  its `mincost` (the default, FLAGS.tiling_alg) and exhaustive `best` solvers choose, the cost of both choices
  under the solver's own objective, and whether the program's value survived the pass.  (Run under Python 3 the
  reference's pass changes the VALUE of most reductions -- sum(arange((11, 12, 13)), 0) comes back divided by the
  number of row tiles it chose -- and cannot build the graph of some programs (None < int comparisons); such
  records carry value_unchanged = false / skipped.  What is pinned from here is the SOLVER: graph in, choice out.)"""
  sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
  from tests import programs
  from spartan.config import FLAGS
  from spartan.expr.operator import optimize
  real = optimize.tiling
  out = []
  for workers in (4, 8):
    for name, build, expected, tol in programs.programs():
      rec = {'program': name, 'workers': workers}
      for alg in ('mincost', 'best'):
        start_cluster(sp, workers)
        FLAGS.opt_auto_tiling = True
        FLAGS.tiling_alg = alg
        FLAGS.num_workers = workers
        optimize._tiled_exprlist.clear() if hasattr(optimize, '_tiled_exprlist') else None
        calls = []

        class Spy(object):
          def __getattr__(self, fn):
            def call(t, edges, groups):
              if alg == 'best' and len(groups) > 9:
                raise OverflowError('4^%d assignments' % len(groups))
              # (costs reach the solver as C longs: Python 2 truncated a float cost silently, Python 3's
              #  PyArg_ParseTuple refuses it)
              edges = [(int(u), int(v), int(c)) for u, v, c in edges]
              res = getattr(real, fn)(t, edges, groups)
              calls.append({'t': int(t), 'edges': [[int(u), int(v), int(c)] for u, v, c in edges],
                            'groups': [[int(x) for x in g] for g in groups], 'chosen': sorted(int(x) for x in res)})
              return res
            return call
        optimize.tiling = Spy()
        try:
          expr = build(sp)
          if not hasattr(expr, 'optimized'):
            raise TypeError('not an expression')
          opt = expr.optimized()
          val = np.asarray(opt.evaluate().glom())
          want = expected()
          ok = bool(np.allclose(val, want, rtol=1e-5, atol=1e-5, equal_nan=True)) and val.shape == want.shape
          rec[alg] = {'calls': calls, 'value_unchanged': ok}
        except Exception as e:   # the reference's pass cannot handle the program
          rec[alg] = {'skipped': '%s: %s' % (type(e).__name__, str(e)[:160])}
        finally:
          optimize.tiling = real
          FLAGS.opt_auto_tiling = False
      out.append(rec)
  return out


def prepare_examples():
  """lib2to3 over the reference's example drivers that BASELINE benchmarks (k-means, SGD regressions)."""
  os.chdir(SCRATCH)
  files = []
  for d in ('spartan/examples', 'spartan/examples/sklearn', 'spartan/examples/sklearn/cluster'):
    files += [os.path.join(d, f) for f in os.listdir(d) if f.endswith('.py')]
  subprocess.check_call([sys.executable, '-m', 'lib2to3', '-w', '-n', '-x', 'map', '-x', 'filter',
                         '-x', 'reduce', '-x', 'zip', '-x', 'import'] + files,
                        stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
  # `np.int` (removed alias) as a dtype; `new_centers / new_counts` etc. are true divisions already
  for f in files:
    t = open(f).read()
    t2 = t.replace('np.int)', 'np.int64)').replace('dtype=np.int,', 'dtype=np.int64,')
    if t2 != t:
      open(f, 'w').write(t2)


def example_inputs():
  """Seeded inputs shared with tests/test_examples.py (kept in the .npz so the tests need no RNG parity)."""
  rng = np.random.RandomState(20150708)
  true_centers = rng.rand(5, 6) * 10
  x = (true_centers[rng.randint(0, 5, size=200)] + rng.randn(200, 6) * 0.1).astype(np.float32)
  init = true_centers + 0.3
  init_empty = init.copy()
  init_empty[4] = 1000.0            # nobody is nearest to it: exercises the empty-cluster re-seed
  xr = rng.rand(96, 8).astype(np.float32)
  yr = rng.rand(96, 1).astype(np.float32)
  return dict(km_x=x, km_init=init, km_init_empty=init_empty, reg_x=xr, reg_y=yr)


def example_goldens(sp, workers):
  from spartan.examples.sklearn.cluster import KMeans
  from spartan.examples import linear_regression, logistic_regression, ridge_regression
  inp = example_inputs()
  out = {}

  def val(v):
    return np.asarray(v.glom() if hasattr(v, 'glom') else v)
  for impl in ('map2', 'outer', 'broadcast', 'shuffle'):
    for tag, init in (('', inp['km_init']), ('_empty', inp['km_init_empty'])):
      start_cluster(sp, workers)
      X = sp.from_numpy(inp['km_x'])
      np.random.seed(4321)
      c0 = init.copy() if impl in ('map2', 'shuffle') else sp.from_numpy(init.copy())
      try:
        centers, labels = KMeans(5, 3).fit(X, c0, implementation=impl)
        out['kmeans_%s%s_centers' % (impl, tag)] = val(centers)
        out['kmeans_%s%s_labels' % (impl, tag)] = val(labels)
      except Exception as e:
        print('   kmeans', impl, tag, 'w%d' % workers, 'failed in the reference:', type(e).__name__, str(e)[:300])
  for name, fn in (('lreg', lambda x, y: linear_regression.linear_regression(x, y, 3)),
                   ('logreg', lambda x, y: logistic_regression.logistic_regression(x, y, 3)),
                   ('ridge', lambda x, y: ridge_regression.ridge_regression(x, y, 1, 2))):
    start_cluster(sp, workers)
    np.random.seed(1234)
    try:
      out[name + '_w'] = np.asarray(fn(sp.from_numpy(inp['reg_x']), sp.from_numpy(inp['reg_y'])))
    except Exception as e:
      print('  ', name, 'w%d' % workers, 'failed in the reference:', type(e).__name__, str(e)[:300])
  return out


def sparse_goldens(sp, workers):
  """Sparse-tile programs (tests/sparse_programs.py) through the reference; values recorded dense."""
  sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
  from tests import sparse_programs
  arrays, meta = {}, {}
  for name, build, tol in sparse_programs.programs():
    start_cluster(sp, workers)
    try:
      expr = build(sp)
      res = expr.evaluate() if hasattr(expr, 'evaluate') else expr
      val, was_sparse = sparse_programs.to_dense(res.glom())
    except Exception as e:  # programs the reference itself cannot run
      meta[name] = {'skipped': '%s: %s' % (type(e).__name__, str(e)[:200])}
      continue
    arrays[name] = val
    meta[name] = {'sparse': bool(was_sparse), 'dtype': val.dtype.str, 'shape': list(val.shape)}
  return arrays, meta


def region_goldens(sp, workers):
  """map2(update_region=...) -- region_join_mapper, map.py:208-241 -- run by the reference."""
  sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
  from tests import region_programs
  arrays = {}
  for name, build in region_programs.programs():
    start_cluster(sp, workers)
    res = build(sp).evaluate()
    arrays[name] = np.asarray(res.glom())
    arrays[name + '__tiles'] = np.asarray(sorted([list(ex.ul) + list(ex.lr) + [int(tid.worker)] for ex, tid in res.tiles.items()]))
  return arrays


def join_goldens(sp, workers):
  """map2 / outer / shuffle with user tile functions (tests/join_programs.py), run by the reference."""
  sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
  from tests import join_programs
  arrays = {}
  for name, build in join_programs.programs():
    start_cluster(sp, workers)
    res = build(sp).evaluate()
    arrays[name] = np.asarray(res.glom())
    arrays[name + '__tiles'] = np.asarray(sorted([list(ex.ul) + list(ex.lr) + [int(tid.worker)] for ex, tid in res.tiles.items()]))
  return arrays


FUZZ_SEEDS = range(5000, 5300)
FUZZ_KEEP = 1024
FUZZ_DOT_SEEDS = range(7000, 7100)


def fuzz_sample(val):
  """What is kept of one program's output: all of it up to FUZZ_KEEP elements, else every k-th element of its ravel
  (k = ceil(size / FUZZ_KEEP)); tests/test_fuzz_reference.py takes the same sample of what the product computes."""
  flat = np.ascontiguousarray(val).ravel()
  if flat.size > FUZZ_KEEP:
    flat = flat[::-(-flat.size // FUZZ_KEEP)]
  return flat


def fuzz_goldens(sp, workers):
  """The random expression DAGs of tests/test_fuzz_gpu.py (element-wise trees with broadcasting and dtype mixes,
  views, reductions and arg-reductions over every axis, fused or not) built over the REFERENCE's builders and run by
  it: the programs it can run (about two thirds: it has no >= / <= on expressions, and its updates assert equal
  dtypes) become fixtures -- shape, dtype, float64 sum and a sample of the values per seed."""
  sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
  import importlib
  fz = importlib.import_module('tests.test_fuzz_gpu')
  arrays, meta = {}, {}
  for seed in FUZZ_SEEDS:
    start_cluster(sp, workers)
    try:
      with np.errstate(all='ignore'):
        val = np.asarray(fz._program(seed, sp))
    except Exception as e:   # noqa: BLE001 -- a program the reference cannot run is not a fixture
      meta[str(seed)] = {'skipped': '%s: %s' % (type(e).__name__, str(e)[:120])}
      continue
    arrays['s%d' % seed] = fuzz_sample(val)
    with np.errstate(all='ignore'):
      total = float(np.nansum(val.astype(np.float64))) if val.size else 0.0
    meta[str(seed)] = {'shape': list(val.shape), 'dtype': val.dtype.str, 'sum': total if np.isfinite(total) else None}
  # ... and its random dots (every dispatch branch of dot.py:243-299: matrix.matrix, matrix.vector, vector.vector, a
  # NumPy right-hand side, tile hints; integer-valued operands, so every product is exact)
  for seed in FUZZ_DOT_SEEDS:
    start_cluster(sp, workers)
    try:
      val = np.asarray(fz._dot_case(seed, sp))
    except Exception as e:   # noqa: BLE001
      meta['d%d' % seed] = {'skipped': '%s: %s' % (type(e).__name__, str(e)[:120])}
      continue
    arrays['d%d' % seed] = fuzz_sample(val)
    meta['d%d' % seed] = {'shape': list(val.shape), 'dtype': val.dtype.str, 'sum': float(val.astype(np.float64).sum())}
  return arrays, meta


def dot_grid_goldens(sp):
  """spartan.dot on operands cut into a 2-D GRID of tiles -- the tiling of the reference's own tests/benchmark_dot.py
  (tile_hint=(T, T)) -- as the reference computes it.  (What it computes is NOT the matrix product: the join turns
  grid cell number b into a slab ONE index thick, extent.pyx:545-552, so only the first <number of cells> indices
  of the contraction take part; the benchmark never looks at the values.  Recorded so that the build reproduces the
  reference here as everywhere else, and says so.)"""
  rng = np.random.RandomState(20150708)
  out = {}
  for name, (m, k, n, t) in (('sq16', (16, 16, 16, (8, 8))), ('wide', (12, 24, 8, (6, 8)))):
    a = rng.randint(-3, 4, size=(m, k)).astype(np.float64)
    b = rng.randint(-3, 4, size=(k, n)).astype(np.float64)
    out[name + '__a'], out[name + '__b'] = a, b
    out[name + '__hint'] = np.asarray(t)
    for workers in (1, 4):
      start_cluster(sp, workers)
      A = sp.from_numpy(a, tile_hint=t)
      B = sp.from_numpy(b, tile_hint=(t[1], t[1]) if name == 'wide' else t)
      out['%s__w%d' % (name, workers)] = np.asarray(sp.dot(A, B).glom())
  return out


if __name__ == '__main__':
  if '--dotgrid' in sys.argv:
    if not os.path.exists(os.path.join(SCRATCH, 'spartan')):
      prepare_tree()
      build_cython()
    prepare_examples()
    install_stubs()
    sp = import_reference()
    res = dot_grid_goldens(sp)
    np.savez_compressed(os.path.join(OUT, 'dot_grid.npz'), **res)
    print('dot on grid tiles:', sorted(res))
    sys.stdout.flush()
    os._exit(0)
  if '--fuzz' in sys.argv:
    if not os.path.exists(os.path.join(SCRATCH, 'spartan')):
      prepare_tree()
      build_cython()
    install_stubs()
    sp = import_reference()
    allmeta = {}
    for n in (1, 3, 4, 8):
      arrays, meta = fuzz_goldens(sp, n)
      np.savez_compressed(os.path.join(OUT, 'fuzz_w%d.npz' % n), **arrays)
      allmeta[str(n)] = meta
      print('workers', n, ':', len(arrays), 'of', len(meta), 'programs run by the reference;',
            sum(1 for k in arrays if k.startswith('d')), 'dots')
    json.dump(allmeta, open(os.path.join(OUT, 'fuzz_meta.json'), 'w'), indent=0, sort_keys=True)
    sys.stdout.flush()
    os._exit(0)
  if '--joins' in sys.argv:
    if not os.path.exists(os.path.join(SCRATCH, 'spartan')):
      prepare_tree()
      build_cython()
    install_stubs()
    sp = import_reference()
    for n in (1, 3, 4, 8):
      res = join_goldens(sp, n)
      np.savez_compressed(os.path.join(OUT, 'joins_w%d.npz' % n), **res)
      print('workers', n, ':', {k: res[k].tolist() for k in sorted(res) if not k.endswith('__tiles') and res[k].size <= 3})
    sys.stdout.flush()
    os._exit(0)
  if '--region' in sys.argv:
    if not os.path.exists(os.path.join(SCRATCH, 'spartan')):
      prepare_tree()
      build_cython()
    install_stubs()
    sp = import_reference()
    res = region_goldens(sp, 4)
    np.savez_compressed(os.path.join(OUT, 'region_w4.npz'), **res)
    print('workers 4 :', sorted(k for k in res if not k.endswith('__tiles')))
    sys.stdout.flush()
    os._exit(0)
  if '--sparse' in sys.argv:
    if not os.path.exists(os.path.join(SCRATCH, 'spartan')):
      prepare_tree()
      build_cython()
    install_stubs()
    sp = import_reference()
    allmeta = {}
    for n in (1, 3, 4, 8):
      arrays, meta = sparse_goldens(sp, n)
      np.savez_compressed(os.path.join(OUT, 'sparse_w%d.npz' % n), **arrays)
      allmeta[str(n)] = meta
      print('workers', n, ':', len(arrays), 'arrays')
      for k, m in meta.items():
        if 'skipped' in m:
          print('   skipped', k, m['skipped'][:200])
    json.dump(allmeta, open(os.path.join(OUT, 'sparse_meta.json'), 'w'), indent=0, sort_keys=True)
    sys.stdout.flush()
    os._exit(0)
  if '--tiling' in sys.argv:
    if not os.path.exists(os.path.join(SCRATCH, 'spartan')):
      prepare_tree()
      build_cython()
    print('built', build_tiling_ext())
    # optimize.py calls the Python-2 builtin reduce() (optimize.py:914; lib2to3's `reduce` fixer is off because it
    # would also rewrite Spartan's own reduce)
    opt_py = os.path.join(SCRATCH, 'spartan', 'expr', 'operator', 'optimize.py')
    text = open(opt_py).read()
    if 'from functools import reduce' not in text:
      open(opt_py, 'w').write('from functools import reduce\n' + text)
    install_stubs()
    sp = import_reference()
    res = tiling_goldens(sp)
    json.dump(res, open(os.path.join(OUT, 'tiling_golden.json'), 'w'), indent=0, sort_keys=True)
    n_graphs = sum(len(r[a].get('calls', [])) for r in res for a in ('mincost', 'best') if a in r)
    print('programs x workers:', len(res), ' solver calls recorded:', n_graphs)
    for r in res:
      for a in ('mincost', 'best'):
        if 'skipped' in r.get(a, {}):
          print('   skipped', r['program'], r['workers'], a, r[a]['skipped'])
        elif not r[a]['value_unchanged']:
          print('   VALUE CHANGED', r['program'], r['workers'], a)
    sys.stdout.flush()
    os._exit(0)
  if '--examples' in sys.argv:
    # only the example-driver goldens, re-using an existing scratch build of the reference
    if not os.path.exists(os.path.join(SCRATCH, 'spartan')):
      prepare_tree()
      build_cython()
    prepare_examples()
    install_stubs()
    sp = import_reference()
    np.savez_compressed(os.path.join(OUT, 'examples_inputs.npz'), **example_inputs())
    for n in (1, 3, 4, 8):
      res = example_goldens(sp, n)
      np.savez_compressed(os.path.join(OUT, 'examples_w%d.npz' % n), **res)
      print('workers', n, ':', sorted(res))
    sys.stdout.flush()
    os._exit(0)
  prepare_tree()
  build_cython()
  prepare_examples()
  install_stubs()
  sp = import_reference()
  print('imported reference:', sp)
  json.dump(extent_goldens(), open(os.path.join(OUT, 'extent_golden.json'), 'w'), indent=0, sort_keys=True)
  json.dump(merge_goldens(), open(os.path.join(OUT, 'merge_golden.json'), 'w'), indent=0, sort_keys=True)
  json.dump(fusion_goldens(sp), open(os.path.join(OUT, 'fusion_golden.json'), 'w'), indent=0, sort_keys=True)
  allmeta = {}
  for n in (1, 3, 4, 8):
    arrays, meta = program_goldens(sp, n)
    np.savez_compressed(os.path.join(OUT, 'programs_w%d.npz' % n), **arrays)
    allmeta[str(n)] = meta
    print('workers', n, ':', len(arrays), 'arrays,', sum(1 for m in meta.values() if 'skipped' in m), 'skipped')
    for k, m in meta.items():
      if 'skipped' in m:
        print('   skipped', k, m['skipped'][:150])
  json.dump(json.loads(json.dumps(allmeta, default=lambda o: int(o) if isinstance(o, np.integer) else float(o))),
            open(os.path.join(OUT, 'programs_meta.json'), 'w'), indent=0, sort_keys=True)
  sys.stdout.flush()
  os._exit(0)
