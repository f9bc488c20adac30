"""The merge branches of sparse tiles the product does NOT implement say so, with the reason, instead of computing
something else (reference: spartan/array/tile.pyx:236-247 applies `reducer(old, update)` to two scipy matrices --
only np.add has a sparse meaning there -- and :284-297 converts the tile to LIL and assigns a dense slice, marked
"this is SLOW").  Pinned here so that a caller who hits one of them reads what to do."""
import numpy as np
import pytest
import scipy.sparse

import spartan_amd as sp
from spartan_amd.array import extent, tile


@pytest.fixture
def ctx():
  from oracle.np_backend import NumpyBackend
  c = sp.initialize(backend=NumpyBackend(), num_workers=2)
  yield c
  sp.shutdown()


def test_dense_update_of_a_sparse_tile_is_refused_with_the_way_out(ctx):
  t = tile.from_shape((4, 6), np.float32, tile.TYPE_SPARSE)
  with pytest.raises(NotImplementedError, match='dense update of a sparse tile is not supported; make the target dense'):
    tile.merge(ctx.backend, t, None, np.ones((4, 6), np.float32), np.add)
  # ... and one level up, before anything travels between workers
  a = sp.sparse_rand((8, 6), density=0.5, format='csr', dtype=np.float32).evaluate()
  assert a.sparse
  with pytest.raises(NotImplementedError, match='dense update of a sparse array is not supported; yield a sparse block'):
    a.update(extent.create((0, 0), (4, 6), (8, 6)), ctx.backend.from_numpy(np.ones((4, 6), np.float32)))


def test_hip_backend_combines_sparse_tiles_with_add_only():
  """(no device needed: the refusal comes before any launch)"""
  from spartan_amd import backend_hip
  be = backend_hip.HipBackend.__new__(backend_hip.HipBackend)
  be.launches = 0
  for reducer in (np.maximum, np.minimum, np.multiply):
    with pytest.raises(NotImplementedError, match='sparse tiles combine with np.add only'):
      be.sparse_reduce(object(), object(), reducer)
    with pytest.raises(NotImplementedError, match='sparse tiles combine with np.add only'):
      be.sparse_update(object(), (0, 0), (1, 1), object(), reducer)
  assert be.launches == 0


def test_sparse_tiles_hold_floating_point_values_on_the_device():
  from spartan_amd import sparse
  with pytest.raises(NotImplementedError, match='float32 / float64 values'):
    sparse._check_dtype(np.int32)
  big = scipy.sparse.coo_matrix((np.ones(1, np.float32), (np.zeros(1, np.int64), np.zeros(1, np.int64))), shape=(2 ** 31, 1))
  with pytest.raises(NotImplementedError, match='int32 index range'):
    sparse.from_scipy(big, None)
