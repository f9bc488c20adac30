"""map2 / outer / shuffle with user tile functions (tests/join_programs.py) against the reference's OWN outputs
(tests/golden/joins_w{1,3,4,8}.npz, recorded by make_golden.py --joins): values, dtypes and the result's tile table,
on the NumPy tile backend and on the HIP backend.  Half of the programs depend on the order the reference's kernels run
in (which write stays without a reducer; in which order float32 partials meet with one): see the module's docstring."""
import os

import numpy as np
import pytest

import spartan_amd as sp
from tests import join_programs

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
WORKERS = (1, 3, 4, 8)
GOLD = {w: np.load(os.path.join(HERE, 'joins_w%d.npz' % w)) for w in WORKERS}


def check_all(workers):
  gold = GOLD[workers]
  n = 0
  for name, build in join_programs.programs():
    res = build(sp).force()
    got, want = np.asarray(res.glom()), gold[name]
    assert got.dtype == want.dtype and got.shape == want.shape, (name, got.dtype, want.dtype, got.shape, want.shape)
    np.testing.assert_array_equal(got, want, err_msg='%s on %d workers' % (name, workers))
    tiles = sorted([list(ex.ul) + list(ex.lr) + [int(tid.worker)] for ex, tid in res.tiles.items()])
    np.testing.assert_array_equal(np.asarray(tiles), gold[name + '__tiles'], err_msg=name + ': tile table')
    n += 1
  return n


def test_the_recordings_show_the_kernel_order():
  """What the files say, restated: with one worker the FIRST listed tile writes last (a worker pops its tiles from
  the end), with three the tile of worker 2, with four and eight the last tile; the float sums follow."""
  for w, last_row, total in ((1, 0, 2.0), (3, 100, 2.0), (4, 150, 0.0), (8, 150, 0.0)):
    g = GOLD[w]
    assert g['map2_whole_target_last_write'].tolist() == [last_row + 1.0] * 3
    assert g['shuffle_target_last_write'].tolist() == [[last_row + 1.0] * 3]
    assert g['map2_whole_target_add'].tolist() == [total] * 3 and g['shuffle_target_add'].tolist() == [[total] * 3]


@pytest.mark.parametrize('workers', WORKERS)
def test_join_programs_cpu(workers):
  from oracle.np_backend import NumpyBackend
  sp.initialize(backend=NumpyBackend(), num_workers=workers)
  try:
    assert check_all(workers) == 10
  finally:
    sp.shutdown()


@pytest.mark.gpu
@pytest.mark.parametrize('workers', WORKERS)
def test_join_programs_hip(workers):
  ctx = sp.initialize('hip', num_workers=workers)
  try:
    before = ctx.backend.launches
    assert check_all(workers) == 10
    assert ctx.backend.launches > before
  finally:
    sp.shutdown()
