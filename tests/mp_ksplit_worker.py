"""One rank of an N-process run (NumPy tile backend on CPU; with the argument `hip`: the HIP backend, all ranks
sharing GPU 0 over the staged transport) of the K-split dot pipeline and the collective
combines at world sizes other than 2: dot.ksplit_plan with p blocks per chunk, reduce-scatter into p pieces, the
reductions' reduce-scatter + all-gather, a row-tiled map with a broadcast operand.  Launched by
tests/test_multiprocess.py::test_ksplit_pipeline_more_ranks."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import spartan_amd as sp  # noqa: E402
from oracle.np_backend import NumpyBackend  # noqa: E402


def main():
  world = sp.World.from_env(backend=os.environ.get('SPARTAN_TEST_BACKEND', 'socket'))
  p = world.size
  use_hip = len(sys.argv) > 1 and sys.argv[1] == 'hip'
  if use_hip:
    world.staged = True
    ctx = sp.initialize('hip', num_workers=p, world=world)
    assert ctx.backend.name == 'hip'
  else:
    ctx = sp.initialize(backend=NumpyBackend(), num_workers=p, world=world)
  rng = np.random.RandomState(5)
  m, k, n = 6 * p, 8 * p, 40            # (a wide or square left operand takes the map2 join; a tall one the outer path)
  a = rng.randint(-4, 5, size=(m, k)).astype(np.float32)
  b = rng.randint(-4, 5, size=(k, n)).astype(np.float32)
  os.environ['SPARTAN_DOT_CHUNK_COLS'] = '16'          # whole chunks only: 40 columns -> 4 chunks of 10
  for dtype in (np.float32, np.float64):
    before = dict(world.stats)
    gemms = getattr(ctx.backend, 'gemms', 0)
    A = sp.from_numpy(a.astype(dtype), tile_hint=(m // p, k))
    B = sp.from_numpy(b.astype(dtype), tile_hint=(k // p, n))
    got = sp.dot(A, B, tile_hint=(m // p, n)).force()
    np.testing.assert_array_equal(got.glom(), a.dot(b).astype(dtype))
    assert got.dtype == dtype and sorted(ex.shape for ex in got.tiles) == [(m // p, n)] * p
    # the plan ran: p - 1 blocks sent, one reduce-scatter per column chunk (+ the glom's gather); chunk 0 is my own
    # block + the row ranges above / below it, every later chunk ONE GEMM on the whole slab
    assert world.stats['p2p_msgs'] - before['p2p_msgs'] == p - 1, world.stats
    assert world.stats['collectives'] - before['collectives'] == 4 + 1, world.stats
    assert ctx.backend.gemms - gemms == 1 + (world.rank > 0) + (world.rank < p - 1) + 3, ctx.backend.gemms - gemms
  del os.environ['SPARTAN_DOT_CHUNK_COLS']
  # reductions over p row tiles, both axes, and the index reductions
  x = rng.randint(-9, 10, size=(12 * p, 20)).astype(np.float32)
  x[5, 3] = x[7 * p, 3] = 99.0                          # duplicate maximum in two ranks' tiles: first one wins
  X = sp.from_numpy(x, tile_hint=(12, 20))
  np.testing.assert_array_equal(sp.sum(X, axis=0).glom(), x.sum(0))
  np.testing.assert_array_equal(sp.sum(X, axis=1).glom(), x.sum(1))
  np.testing.assert_array_equal(sp.sum(X).glom(), x.sum())
  np.testing.assert_array_equal(sp.argmax(X, axis=0).glom(), np.argmax(x, 0))
  np.testing.assert_array_equal(sp.argmax(X, axis=1).glom(), np.argmax(x, 1))
  row = rng.randint(-3, 4, size=(20,)).astype(np.float32)
  np.testing.assert_array_equal((X * sp.from_numpy(row) + X).optimized().glom(), x * row + x)
  # one k-means iteration with the points over p ranks against NumPy
  from scipy.spatial.distance import cdist
  from spartan_amd.examples.sklearn.cluster import KMeans
  pts = rng.rand(16 * p, 6).astype(np.float32 if use_hip else np.float64)
  c0 = pts[:5].copy()
  centers, labels = KMeans(5, 1).fit(sp.from_numpy(pts, tile_hint=(16, 6)), c0, implementation='map2', reducer=np.add)
  want_labels = np.argmin(cdist(pts, c0), axis=1)
  np.testing.assert_array_equal(labels.glom().reshape(-1), want_labels)
  want = np.stack([pts[want_labels == c].sum(0) / max(1, (want_labels == c).sum()) for c in range(5)])
  np.testing.assert_allclose(centers, want, rtol=2e-6 if use_hip else 1e-12)
  if use_hip:
    assert ctx.backend.launches > 0
  if world.control.name == 'socket':
    assert 'torch' not in sys.modules, 'the socket control plane must not bring torch in'
  world.barrier()
  print('RANK %d OK' % world.rank)
  sys.stdout.flush()


if __name__ == '__main__':
  main()
