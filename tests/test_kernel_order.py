"""distarray.kernel_order: the order the mappers of one kernel run in, restated from the reference's worker
(worker.py:246-256: collect own tiles in list order, stable sort by np.size of the data, pop from the end) and its
`_send_all` (blob_ctx.py:270-271: worker after worker in the serial runs the recordings come from).  The recorded
programs pin its EFFECT (tests/test_golden.py, tests/test_join_programs.py); this file pins the rule itself."""
import numpy as np
import pytest

import spartan_amd as sp
from spartan_amd.array import distarray


@pytest.fixture
def ctx3():
  from oracle.np_backend import NumpyBackend
  c = sp.initialize(backend=NumpyBackend(), num_workers=3)
  yield c
  sp.shutdown()


def _rows(array, order):
  ex_of = {tid: ex for ex, tid in array.tiles.items()}
  return [(ex_of[t].ul[0], t.worker) for t in order]


def test_worker_by_worker_and_last_listed_first(ctx3):
  a = sp.from_numpy(np.zeros((200, 3), np.float32), tile_hint=(50, 3)).force()      # tiles on workers 0, 1, 2, 0
  order = distarray.kernel_order(a, list(a.tiles.values()), ctx3)
  assert _rows(a, order) == [(150, 0), (0, 0), (50, 1), (100, 2)]


def test_a_smaller_tile_runs_after_the_larger_ones_of_its_worker(ctx3):
  a = sp.from_numpy(np.zeros((203, 3), np.float32), tile_hint=(50, 3)).force()      # workers 0, 1, 2, 0, 1; last tile 3 rows
  order = distarray.kernel_order(a, list(a.tiles.values()), ctx3)
  assert _rows(a, order) == [(150, 0), (0, 0), (50, 1), (200, 1), (100, 2)]


def test_tiles_nothing_was_written_to_count_as_one_element(ctx3):
  a = distarray.create((203, 3), np.float32, tile_hint=(50, 3))                        # unwritten: np.size(None) == 1
  order = distarray.kernel_order(a, list(a.tiles.values()), ctx3)
  assert _rows(a, order) == [(150, 0), (0, 0), (200, 1), (50, 1), (100, 2)]           # (pure reverse list order per worker)


def test_a_subset_and_a_single_tile(ctx3):
  a = sp.from_numpy(np.zeros((200, 3), np.float32), tile_hint=(50, 3)).force()
  tids = list(a.tiles.values())
  assert distarray.kernel_order(a, tids[:1], ctx3) == tids[:1]
  assert _rows(a, distarray.kernel_order(a, [tids[3], tids[1]], ctx3)) == [(150, 0), (50, 1)]


def test_across_ranks_the_extent_is_the_size(ctx3):
  class TwoRanks(object):
    size = 2
  a = distarray.create((203, 3), np.float32, tile_hint=(50, 3))                        # unwritten tiles, ragged last one
  real = ctx3.world
  ctx3.world = TwoRanks()
  try:
    order = distarray.kernel_order(a, list(a.tiles.values()), ctx3)
  finally:
    ctx3.world = real
  # every rank must walk ONE order and cannot see a remote tile's allocation: sizes are the extents'
  assert _rows(a, order) == [(150, 0), (0, 0), (50, 1), (200, 1), (100, 2)]
