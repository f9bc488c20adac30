"""spartan.dot on operands cut into a 2-D grid of tiles -- the tiling of the reference's own tests/benchmark_dot.py --
against outputs RECORDED FROM THE REFERENCE (tests/golden/make_golden.py --dotgrid -> dot_grid.npz).

What the reference computes there is not the matrix product: its join turns grid cell number b into a slab one index
thick (spartan/array/extent.pyx:545-552), so only as many indices of the contraction as there are cells take part,
and its benchmark never looks at the values.  The build reproduces the reference bit for bit here as everywhere else
(same extents, same slabs, same merge) -- this test pins that, and that it is NOT a @ b, so that nobody quotes a
GEMM rate for that tiling; row tiling (the default) is the product."""
import os

import numpy as np
import pytest

import spartan_amd as sp

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'dot_grid.npz'))


import contextlib


def _first_call():
  """The warning is given once per process (expr/map._warned_grid)."""
  import importlib
  return not importlib.import_module('spartan_amd.expr.map')._warned_grid


def _nothing():
  return contextlib.nullcontext()


def _check(workers):
  for name in ('sq16', 'wide'):
    a, b, hint = GOLD[name + '__a'], GOLD[name + '__b'], tuple(int(v) for v in GOLD[name + '__hint'])
    A = sp.from_numpy(a, tile_hint=hint)
    B = sp.from_numpy(b, tile_hint=(hint[1], hint[1]) if name == 'wide' else hint)
    with pytest.warns(RuntimeWarning, match='not the matrix product') if _first_call() else _nothing():
      got = sp.dot(A, B).glom()
    want = GOLD['%s__w%d' % (name, workers)]
    assert got.dtype == want.dtype
    np.testing.assert_array_equal(got, want, err_msg='%s, %d workers' % (name, workers))
    assert not np.array_equal(want, a.dot(b))                       # (the reference's own answer is not the product)
    np.testing.assert_array_equal(sp.dot(sp.from_numpy(a), sp.from_numpy(b)).glom(), a.dot(b))   # row tiles: it is


@pytest.mark.parametrize('workers', [1, 4])
def test_dot_of_grid_tiled_operands_is_the_references(workers):
  from oracle.np_backend import NumpyBackend
  sp.initialize(backend=NumpyBackend(), num_workers=workers)
  try:
    _check(workers)
  finally:
    sp.shutdown()


@pytest.mark.gpu
@pytest.mark.parametrize('workers', [1, 4])
def test_dot_of_grid_tiled_operands_is_the_references_hip(workers):
  sp.initialize('hip', num_workers=workers)
  try:
    _check(workers)
  finally:
    sp.shutdown()
