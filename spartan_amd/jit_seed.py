"""Pre-populate the run-time specialisation cache that travels with the tree (csrc/jit_seed/*.spco).

A fused program outside the prebuilt kernel library runs its first launches on the interpreter tier while hipRTC
compiles its specialisation in the background (csrc/sp_jit.hip) -- on a fresh machine that is the first ~0.4 s of
every such program.  `__graft_entry__.build()` calls `seed()`: the expressions the workloads are known to force are
pushed through the REAL host path (builders -> optimiser -> lowering -> sp_map_fused / sp_reduce) with the library in
seed mode (sp_jit_seed_begin): every specialisation a launch asks for is compiled on the spot and written next to
the library instead of being loaded.  No GPU is needed: without one the tiles are HostStorage stand-ins and the launches
themselves fail and are ignored here; on a machine WITH a GPU the tiles are ordinary device tiles (a host pointer handed
to a launch that really runs is a memory fault, which aborts the process) and the launches simply run.  At run time a code object found in jit_seed is loaded instead of compiled, so those programs
start specialised from their first launch; the file name carries a hash of the kernel headers' contents, so stale
seeds are never used.
"""
import contextlib
import ctypes as C
import os

import numpy as np

from . import _hip, devarray, kernels, sparse
from .backend_hip import HipBackend


class _SeedBackend(HipBackend):
  name = 'hip'

  def __init__(self):   # no device check: nothing is launched successfully in seed mode
    import collections
    self.device = 'seed'
    self._np_cache = collections.OrderedDict()
    self.launches = self.gemms = self.host_round_trips = 0
    self._warned_host = set()
    self.gemm_events = None
    self._rng_seed, self._rng_offset = 1, 0
    self._lowered = collections.OrderedDict()
    self.lowering_hits = 0
    self._side_copies, self._pinned_free = None, {}
    self._fixed_points = None

  def _lowering_key(self, op, inputs, ex, extra):
    return None          # every program is lowered (and its specialisation requested) afresh


@contextlib.contextmanager
def _seed_mode(directory=None):
  """Launch errors ignored, tiles in host memory unless a GPU would really run the launches, library in seed mode."""
  lib = _hip.lib()
  count = C.c_int(0)
  have_gpu = lib.sp_device_count(C.byref(count)) == 0 and count.value > 0
  saved = [(m, m.check) for m in (_hip, kernels, devarray, sparse)]
  for m, _ in saved:
    m.check = lambda rc: None
  if not have_gpu:
    devarray._storage_cls[0] = devarray.HostStorage
  # code objects of earlier builds (other source hashes in their names) would only be loaded for nothing
  target = directory or os.path.join(os.path.dirname(os.path.abspath(_hip.__file__)), 'csrc', 'jit_seed')
  if os.path.isdir(target):
    for name in os.listdir(target):
      if name.endswith('.spco'):
        os.remove(os.path.join(target, name))
  ok = lib.sp_jit_seed_begin(directory.encode() if directory else None)
  try:
    yield bool(ok)
  finally:
    for m, fn in saved:
      m.check = fn
    devarray._storage_cls[0] = devarray.Storage


def default_programs(sp):
  """The fused trees bench.py and the workload drivers force on large tiles, plus the common shapes of user code
  ((x - c)^2, a * b + c, x * s + t, where, ...), as builders over a tile `x` (and `y`, `z` of the same shape)."""
  def progs(x, y, z):
    return [
        ((x * x + x) * 0.5 - x) / (x + 2.0),                 # bench.py map_5op_chain
        x * x + x, x + 1.0, x * y, x + y, x - y, x / y,       # (fp32: in the prebuilt library; fp64 / int: not)
        sp.sum(x, axis=0), sp.sum(x, axis=1), sp.sum(x), sp.max(x, axis=0), sp.max(x), sp.min(x, axis=1),
        (x - 0.5) * (x - 0.5),
        x * y + z, x * y - z, (x + y) * z, x * 2.0 + 1.0, x * 0.5 - 3.0, (x - y) * (x - y),
        sp.sqrt(sp.abs(x)), sp.exp(x * -1.0), sp.ln(x + 1.0), sp.maximum(x, 0.0) + y, sp.square(x - y) + z,
        sp.sum((x - 0.5) * (x - 0.5), axis=0),               # bench.py sum_sq_dev_axis0
        sp.sum((x - 0.5) * (x - 0.5), axis=1), sp.sum((x - 0.5) * (x - 0.5)),
        sp.sum(x * y, axis=0), sp.sum(x * y, axis=1), sp.sum(x * y),
        sp.sum((x - y) * (x - y), axis=1), sp.sum(sp.abs(x), axis=0), sp.max(x * y, axis=0), sp.min(x - y, axis=1),
    ]
  return progs


def seed(directory=None, verbose=False):
  """Compile and store the specialisations of default_programs for fp32 and fp64 tiles, small enough to stay in the
  L2 and larger (the two differ in their load / store hints).  Returns the number of code objects written."""
  import spartan_amd as sp
  from . import context as _context
  from .expr import base as _base
  shapes = ((2048, 2048 + 64), (8192, 4096 + 64))          # 4.3 M and 34 M elements: below / above the L2
  with _seed_mode(directory) as have_rtc:
    if not have_rtc:
      return 0
    prev = _context._ctx
    _context.set(_context.Context(_SeedBackend(), None, 1))
    _base.eval_cache.clear()
    try:
      progs = default_programs(sp)
      for dtype in (np.float32, np.float64):
        for shape in shapes:
          mk = lambda: sp.from_tile_fn(shape, dtype, lambda ex: devarray.empty(ex.shape, dtype))   # noqa: E731
          x, y, z = mk(), mk(), mk()
          for e in progs(x, y, z):
            try:
              e.optimized().force()
            except Exception as err:   # noqa: BLE001  (a program this build cannot lower is simply not seeded)
              if verbose:
                print('jit_seed: skipped %s: %s' % (type(err).__name__, err))
    finally:
      _context.set(prev)
      _base.eval_cache.clear()
      written = _hip.lib().sp_jit_seed_end()
  return written


if __name__ == '__main__':
  print('code objects written:', seed(verbose=True))
