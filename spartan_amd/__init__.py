"""spartan_amd: an MI355X-native tile-execution backend behind the Spartan
expression-builder API.

    import spartan_amd as spartan
    spartan.initialize()                       # one process per GPU (RANK/WORLD_SIZE aware)
    x = spartan.ones((1000, 1000)) + 1
    y = x.force()                              # == x.evaluate(): a DistArray of HBM tiles
    y.glom()                                   # NumPy array

The public names mirror the reference's `spartan` / `spartan.expr` namespaces
(reference spartan/__init__.py, spartan/expr/__init__.py:26-93).  See DESIGN.md.
"""
__version__ = '0.1.0'

import os as _os
_os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # the host driver only supports dmabuf IPC (RCCL across processes)

import numpy as _np

from . import context as _context
from .array import distarray, extent, tile  # noqa: F401
from .comm import World
from .expr.base import (Expr, NotShapeable, Val, as_array, eager, evaluate, glom, lazify,  # noqa: F401
                        newaxis, optimized_dag)
from .expr.broadcast import broadcast  # noqa: F401
from .expr.builtins import *  # noqa: F401,F403
from .expr.builtins import (abs, all, any, max, min, sum)  # noqa: F401  (shadow the builtins, like spartan.expr)
from .expr.dot import dot  # noqa: F401
from .expr.map import map, map2, map_with_location  # noqa: F401
from .expr.manip import bincount, concatenate, diag, diagflat, diagonal, norm, normalize  # noqa: F401
from .expr.ndarray import ndarray  # noqa: F401
from .expr.optimize import optimize  # noqa: F401
from .expr.outer import outer  # noqa: F401
from .expr.reduce import reduce  # noqa: F401
from .expr.shuffle import shuffle  # noqa: F401
from .expr.views import ravel, reshape, transpose  # noqa: F401  (also installs Expr.__getitem__/.T/.reshape)
from .expr.write import write  # noqa: F401

# Operators next to the tile path (outside section 8): not part of the default import; the names
# resolve on first use (module __getattr__ below), their kernels live in libspartan_hip_extras.so.
_EXTRAS = {
    'scan': 'expr.scan', 'sort': 'expr.sort', 'argsort': 'expr.sort', 'partition': 'expr.sort',
    'argpartition': 'expr.sort', 'from_file': 'expr.fio', 'from_file_parallel': 'expr.fio', 'load': 'expr.fio',
    'partial_load': 'expr.fio', 'partial_unpickle': 'expr.fio', 'pickle': 'expr.fio', 'save': 'expr.fio',
    'unpickle': 'expr.fio', 'tile_operation': 'expr.tile_operation', 'checkpoint': 'expr.checkpoint',
    'stencil': 'expr.stencil', 'maxpool': 'expr.stencil', '_convolve': 'expr.stencil',
    'assign': 'expr.region', 'region_map': 'expr.region', 'retile': 'expr.region',
}


def __getattr__(name):
  where = _EXTRAS.get(name)
  if where is None:
    raise AttributeError('module %r has no attribute %r' % (__name__, name))
  import importlib
  value = getattr(importlib.import_module('.' + where, __name__), name)
  globals()[name] = value
  from . import expr as _expr
  setattr(_expr, name, value)          # the flat spartan.expr namespace (over the submodule of the same name)
  return value


# ndarray-style methods on expressions (spartan/expr/__init__.py:66-92)
Expr.all = all
Expr.any = any
Expr.argmax = argmax  # noqa: F405
Expr.argmin = argmin  # noqa: F405
Expr.astype = astype  # noqa: F405
Expr.dot = dot
Expr.max = max
Expr.mean = mean  # noqa: F405
Expr.min = min
Expr.prod = prod  # noqa: F405
Expr.std = std  # noqa: F405
Expr.sum = sum
distarray.DistArray.evaluate = evaluate
distarray.DistArray.force = evaluate
# evaluated arrays take the same operators / methods (spartan/expr/__init__.py:97-139).  `==` / `!=` are
# NOT patched (the reference maps them to equal / not_equal): arrays are compared by identity inside the
# framework; use spartan.equal(a, b).
for _name, _fn in dict(
    __add__=add, __sub__=sub, __mul__=multiply, __mod__=mod, __truediv__=divide, __lt__=less,  # noqa: F405
    __gt__=greater, __and__=logical_and, __or__=logical_or, __pow__=power, __neg__=negative,  # noqa: F405
    __radd__=add, __rmul__=multiply, __rsub__=lambda a, b: sub(b, a), __rtruediv__=lambda a, b: divide(b, a),  # noqa: F405
    all=all, any=any, argmax=argmax, argmin=argmin, astype=astype, dot=dot,  # noqa: F405
    fill=full_like, flatten=ravel, max=max, mean=mean, min=min, prod=prod, ravel=ravel, reshape=reshape,  # noqa: F405
    std=std, sum=sum, transpose=transpose).items():  # noqa: F405
  setattr(distarray.DistArray, _name, _fn)
distarray.DistArray.T = property(transpose)
Expr.fill = full_like  # noqa: F405
Expr.flatten = ravel
Expr.diagonal = diagonal
# (spartan/expr/__init__.py:70-85 binds these names too: three of them to None, the sort family to its builders)
Expr.flat = None
Expr.outer = None
Expr.nonzero = None
for _name in ('argsort', 'argpartition', 'partition'):
  setattr(Expr, _name, (lambda _n: lambda self, *a, **kw: __getattr__(_n)(self, *a, **kw))(_name))


def _export_expr_namespace():
  """`from spartan import expr; expr.rand(...)`: the reference's spartan.expr is a flat
  namespace of builders (spartan/expr/__init__.py:26-93); re-export ours on the package."""
  import types
  from . import expr as _expr
  for name, value in list(globals().items()):
    if not name.startswith('_') and not isinstance(value, types.ModuleType):
      setattr(_expr, name, value)


_export_expr_namespace()


def initialize(backend='hip', num_workers=None, world=None):
  """Create the process-wide worker context (reference spartan.initialize,
  spartan/__init__.py:42-56: start_cluster + blob_ctx).

  backend: 'hip' (the product path: HIP kernels on the local MI355X; raises if
    the GPU or the built library is missing -- there is no CPU fallback), or a
    backend OBJECT (the test-suite injects oracle.np_backend.NumpyBackend to
    exercise the host logic without a GPU).
  num_workers: logical workers (default: one per process).
  world: a spartan_amd.World (default: from RANK/WORLD_SIZE, else 1 process).
  """
  if world is None:
    world = World.from_env()
  if backend == 'hip':
    from .backend_hip import HipBackend
    backend = HipBackend()
  elif isinstance(backend, str):
    raise ValueError("unknown backend %r: the product backend is 'hip'" % backend)
  ctx = _context.Context(backend, world, num_workers)
  _context.set(ctx)
  from .expr import base as _base
  _base.eval_cache.clear()
  return ctx


def shutdown():
  if _context.initialized():
    ctx = _context.get()
    if ctx.heartbeat is not None:
      ctx.heartbeat.stop()
      ctx.heartbeat = None
    release = getattr(ctx.backend, 'release_pinned', None)
    if release is not None:
      release()
  _context.set(None)


def get_context():
  return _context.get()
