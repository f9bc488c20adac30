"""map2(update_region=...) (reference region_join_mapper, map.py:208-241,325-333,359-362) against the reference's
own outputs (tests/golden/region_w4.npz, recorded by make_golden.py --region): values, dtypes and the target's tile
table, on the NumPy tile backend, on the HIP backend and across two gloo ranks."""
import os

import numpy as np
import pytest

import spartan_amd as sp
from tests import region_programs

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
GOLD = np.load(os.path.join(HERE, 'region_w4.npz'))


def check_all():
  n = 0
  for name, build in region_programs.programs():
    res = build(sp).force()
    got = res.glom()
    want = GOLD[name]
    assert got.dtype == want.dtype and got.shape == want.shape, (name, got.dtype, want.dtype)
    np.testing.assert_array_equal(got, want, err_msg=name)
    tiles = sorted([list(ex.ul) + list(ex.lr) + [int(tid.worker)] for ex, tid in res.tiles.items()])
    np.testing.assert_array_equal(np.asarray(tiles), GOLD[name + '__tiles'], err_msg=name + ': tile table')
    n += 1
  # what the goldens say, restated: only the boxes change, by the tile body's arithmetic
  a = region_programs._a()
  want = a.copy()
  want[32:, 32:] = a[32:, 32:] * 2 + 1
  np.testing.assert_array_equal(GOLD['diag_cell'], want)
  want = a.copy()
  want[10:40, 5:20] = -7
  want[50:60, 40:64] = -7
  np.testing.assert_array_equal(GOLD['unaligned_box_scalar'], want)
  return n


def test_region_join_cpu():
  from oracle.np_backend import NumpyBackend
  sp.initialize(backend=NumpyBackend(), num_workers=4)
  try:
    assert check_all() == 5
  finally:
    sp.shutdown()


def test_region_join_normalises_a_single_box():
  from oracle.np_backend import NumpyBackend
  sp.initialize(backend=NumpyBackend(), num_workers=4)
  try:
    box = sp.extent.create((0, 0), (8, 8), (64, 64))
    e = sp.map2(sp.ones((64, 64)), ((0, 1),), fn=region_programs._minus_seven, shape=(64, 64), update_region=box)
    assert e.update_region == (box,)                       # map.py:359-362
    e = sp.map2(sp.ones((64, 64)), ((0, 1),), fn=region_programs._minus_seven, shape=(64, 64), update_region=[box, box])
    assert e.update_region == (box, box)
  finally:
    sp.shutdown()


@pytest.mark.gpu
def test_region_join_hip():
  ctx = sp.initialize('hip', num_workers=4)
  try:
    before = ctx.backend.launches
    assert check_all() == 5
    assert ctx.backend.launches > before
  finally:
    sp.shutdown()


def test_region_join_two_gloo_ranks():
  from tests.test_multiprocess import _run_ranks
  _run_ranks(2, 'mp_region_worker.py', ['4'])


@pytest.mark.gpu
def test_region_join_two_ranks_hip_shared_gpu():
  from tests.test_multiprocess import _run_ranks
  _run_ranks(2, 'mp_region_worker.py', ['4', 'hip'])
