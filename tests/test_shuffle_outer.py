"""shuffle (target / no-target) and outer with a partitioned second operand:
the remaining operators of the path (reference operator/shuffle.py,
operator/outer.py:30-57), on the NumPy backend (CPU) and the HIP backend (GPU).
User functions are written against NumPy arrays, as the reference's mappers are (shuffle.py:41-96,
outer.py:12-59, map.py:243-286: `data.T`, `data.sum(axis, keepdims=True)`, `tiles[0] * 2 + tiles[1]`,
`np.maximum(...)`, `tile_a.dot(tile_b)`): they receive np.ndarray tiles on the NumPy backend and device arrays
that answer the same calls with HIP kernels on the HIP backend -- the same function, unchanged, on both."""
import numpy as np
import pytest

import spartan_amd as sp
from spartan_amd.array import extent


def _backends():
  from oracle.np_backend import NumpyBackend
  yield pytest.param(lambda: NumpyBackend(), id='numpy')
  yield pytest.param('hip', id='hip', marks=pytest.mark.gpu)


@pytest.fixture(params=list(_backends()))
def make_backend(request):
  return request.param


def _init(make_backend, workers):
  be = make_backend if isinstance(make_backend, str) else make_backend()
  return sp.initialize(backend=be, num_workers=workers)


def _transpose_fn(source, ex):
  """tile -> its transposed block (the body of the reference's tests/test_shuffle-style mappers)."""
  data = source.fetch(ex)
  tex = extent.create(ex.ul[::-1], ex.lr[::-1], source.shape[::-1])
  return [(tex, data.T)]


@pytest.mark.parametrize('workers', [1, 3, 4])
def test_shuffle_notarget_transpose(make_backend, workers):
  _init(make_backend, workers)
  a = np.arange(60 * 28, dtype=np.float32).reshape(60, 28)
  r = sp.shuffle(sp.from_numpy(a), _transpose_fn, shape_hint=(28, 60))
  np.testing.assert_array_equal(r.glom(), a.T)
  sp.shutdown()


@pytest.mark.parametrize('workers', [1, 3, 4])
def test_shuffle_target_accumulates(make_backend, workers):
  _init(make_backend, workers)
  a = (np.arange(60 * 28, dtype=np.float32).reshape(60, 28) % 11) - 5

  def colsum_fn(source, ex):
    data = source.fetch(ex)
    tex = extent.create((0, ex.ul[1]), (1, ex.lr[1]), (1, source.shape[1]))
    return [(tex, data.sum(axis=0, keepdims=True))]
  target = sp.ndarray((1, 28), dtype=np.float32, reduce_fn=np.add)
  r = sp.shuffle(sp.from_numpy(a), colsum_fn, target=target)
  np.testing.assert_array_equal(r.glom(), a.sum(0, keepdims=True))
  sp.shutdown()


@pytest.mark.parametrize('workers', [1, 3, 4])
def test_outer_partitioned_rhs(make_backend, workers):
  """outer((a, b), (0, 1), fn): every row block of a against every column block of b
  (outer.py:30-57, axes[1] not None)."""
  ctx = _init(make_backend, workers)
  a = (np.arange(48 * 20, dtype=np.float32).reshape(48, 20) % 7) - 3
  b = (np.arange(20 * 36, dtype=np.float32).reshape(20, 36) % 5) - 2

  def block_dot(ex_a, tile_a, ex_b, tile_b):
    tex = extent.create((ex_a.ul[0], ex_b.ul[1]), (ex_a.lr[0], ex_b.lr[1]), (48, 36))
    yield tex, tile_a.dot(tile_b)
  r = sp.outer((sp.from_numpy(a), sp.from_numpy(b)), (0, 1), block_dot, shape=(48, 36), reducer=np.add,
               tile_hint=(16, 36))
  np.testing.assert_array_equal(r.glom(), a.dot(b))
  sp.shutdown()


def test_user_map2_join(make_backend):
  """map2 with a user join function over matching row blocks (map.py:243-286)."""
  _init(make_backend, 3)
  a = np.arange(90 * 8, dtype=np.float32).reshape(90, 8)
  b = np.arange(90 * 8, dtype=np.float32).reshape(90, 8)[::-1].copy()

  def join(extents, tiles):
    ex = extents[0]
    yield ex, np.maximum(tiles[0] * 2.0 + tiles[1], tiles[1] - 1e9)
  r = sp.map2((sp.from_numpy(a), sp.from_numpy(b)), (0, 0), fn=join, shape=(90, 8))
  np.testing.assert_array_equal(r.glom(), a * 2 + b)
  sp.shutdown()
