"""Differential fuzz of the builders that close the reference's namespace (spartan/expr/__init__.py:26-38): bincount,
normalize, norm, diagonal, diag, diagflat, concatenate -- random shapes, dtypes and worker counts, against NumPy where
the reference's function IS NumPy's (its quirks -- weighted bincount on several tiles, normalize along an axis -- are
pinned by the reference-generated goldens in tests/programs.py, not here).  The host framework on the NumPy oracle
backend runs everywhere; the same cases on the HIP backend are the `gpu` half."""
import numpy as np
import pytest

import spartan_amd as sp

CASES = 60


def _cases(seed):
  rng = np.random.RandomState(seed)
  for _ in range(CASES):
    kind = rng.randint(7)
    fdt = [np.float32, np.float64][rng.randint(2)]
    if kind == 0:                                     # bincount of positive integers
      n = int(rng.randint(1, 4000))
      v = rng.randint(1, int(rng.randint(2, 300)), size=n).astype([np.int64, np.int32][rng.randint(2)])
      ml = [None, int(rng.randint(1, 400))][rng.randint(2)]
      yield ('bincount', lambda api, v=v, ml=ml: api.bincount(api.from_numpy(v), minlength=ml),
             np.bincount(v, minlength=0 if ml is None else ml), 'exact')
    elif kind == 1:                                   # normalize(axis=None)
      shape = tuple(int(s) for s in rng.randint(1, 90, size=rng.randint(1, 3)))
      a = (rng.rand(*shape) + 0.1).astype(fdt)
      yield ('normalize', lambda api, a=a: api.normalize(api.from_numpy(a)), a / a.sum(), 'close')
    elif kind == 2:                                   # norm: 1-norm of a matrix / vector, 2-norm of a vector
      if rng.rand() < 0.5:
        a = (rng.rand(int(rng.randint(1, 200)), int(rng.randint(1, 60))) - 0.5).astype(fdt)
        yield ('norm1', lambda api, a=a: api.norm(api.from_numpy(a), 1), np.linalg.norm(a, 1), 'close')
      else:
        a = (rng.rand(int(rng.randint(1, 3000))) - 0.5).astype(fdt)
        yield ('norm2', lambda api, a=a: api.norm(api.from_numpy(a), 2), np.linalg.norm(a, 2), 'close')
    elif kind == 3:                                   # diagonal of a matrix (tall, wide, square)
      a = rng.rand(int(rng.randint(1, 150)), int(rng.randint(1, 150))).astype(fdt)
      yield ('diagonal', lambda api, a=a: api.diagonal(api.from_numpy(a)), np.diagonal(a), 'exact')
    elif kind == 4:                                   # diagflat / diag of a vector
      a = rng.rand(int(rng.randint(1, 120))).astype(fdt)
      f = ['diagflat', 'diag'][rng.randint(2)]
      yield (f, lambda api, a=a, f=f: getattr(api, f)(api.from_numpy(a)), np.diagflat(a), 'exact')
    elif kind == 5:                                   # diag of a matrix
      a = rng.randint(-9, 9, size=(int(rng.randint(1, 80)), int(rng.randint(1, 80)))).astype(np.int64)
      yield ('diag2', lambda api, a=a: api.diag(api.from_numpy(a)), np.diag(a), 'exact')
    else:                                             # concatenate along either axis, or of vectors
      if rng.rand() < 0.3:
        # (vectors of ONE length: the reference's join gives tile [lo, hi) of `a` the slab [lo, hi) of `b`,
        #  manipulation.py:44-57 with map2's axes (0, 0) -- a longer `b` loses its end, a shorter one is out of bounds)
        m = int(rng.randint(1, 500))
        a, b = rng.rand(m).astype(fdt), rng.rand(m).astype(fdt)
        axis = 0
      else:
        axis = int(rng.randint(2))
        r, c = int(rng.randint(1, 90)), int(rng.randint(1, 90))
        other = int(rng.randint(1, 90))
        a = rng.rand(r, c).astype(fdt)
        b = rng.rand(other, c).astype(fdt) if axis == 0 else rng.rand(r, other).astype(fdt)
      yield ('concatenate', lambda api, a=a, b=b, axis=axis: api.concatenate(api.from_numpy(a), api.from_numpy(b), axis),
             np.concatenate((a, b), axis), 'exact')


def _run(seed):
  bad = []
  for i, (name, build, want, how) in enumerate(_cases(seed)):
    got = build(sp)
    got = np.asarray(got.glom() if hasattr(got, 'glom') else got)
    want = np.asarray(want)
    if got.shape != want.shape:
      bad.append((i, name, 'shape', got.shape, want.shape))
    elif how == 'exact':
      # (diagflat's blocks are float64 whenever the array has several tiles -- creation.py:247-250 -- the VALUES are
      #  the operand's, exactly)
      if not np.array_equal(got.astype(np.float64), want.astype(np.float64)):
        bad.append((i, name, 'values'))
    elif not np.allclose(got, want, rtol=2e-5 if want.dtype == np.float32 or got.dtype == np.float32 else 1e-12, atol=0):
      bad.append((i, name, 'values', float(np.abs(got - want).max())))
  return bad


@pytest.mark.parametrize('workers', [1, 3, 4, 8])
def test_builders_on_the_oracle_backend(workers):
  from oracle.np_backend import NumpyBackend
  sp.initialize(backend=NumpyBackend(), num_workers=workers)
  try:
    assert _run(4000 + workers) == []
  finally:
    sp.shutdown()


@pytest.mark.gpu
@pytest.mark.parametrize('workers', [1, 3, 8])
def test_builders_on_the_hip_backend(workers):
  sp.initialize('hip', num_workers=workers)
  try:
    assert _run(4000 + workers) == []
  finally:
    sp.shutdown()
