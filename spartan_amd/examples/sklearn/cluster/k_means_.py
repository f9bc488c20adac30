"""Lloyd's k-means over a row-tiled point array: the workload of BASELINE configs[3].

Interface of the reference's spartan/examples/sklearn/cluster/k_means_.py (`KMeans(n_clusters, n_iter)
.fit(X, centers, implementation)` -> (centers, labels)), including its four ways of phrasing one iteration,
because each of them exercises a different part of the tile path:

  'map2'       assign = map2 join of every point tile with driver-side centers; accumulate = two more map2
               joins (counts, per-cluster sums) merged into k-row targets
  'outer'      the same with the centers as a distributed array fetched whole by every tile (outer join)
  'broadcast'  pure expressions: (N,1,D) - (1,K,D), square, sum, argmin, one-hot sums -- map / reduce fusion
  'shuffle'    one shuffle whose mapper updates three reducer targets

Per tile the work is three kernels behind the backend:
  nearest_center  labels = argmin_c |x - c|   sp_nearest_center (fp32 MFMA, candidate re-check in fp64)
  bincount        counts of each label         sp_bincount_i64
  segment_sum     per-cluster sums of rows     sp_segment_sum

An iteration is  assign -> accumulate -> `_finish` (driver: re-seed empty clusters, divide).  Two behaviours of
the reference are kept because its recorded outputs (tests/golden/examples_w*.npz) depend on them: the join
variants create their count / sum targets WITHOUT a reducer (k_means_.py:133-141), so with several tiles the last
tile's partial replaces the others -- pass `reducer=np.add` for the real update, as bench.py does -- and the
'shuffle' variant divides by a (k, 1) count column.  Everything the driver draws at random is drawn by rank 0
and sent to the other ranks: with one process per GPU every rank runs this loop and must step the same centers.
"""
import numpy as np

from .... import context, expr
from ....array import distarray, extent


def _replicated(draw):
  """A driver-level random value, identical on every rank."""
  value = draw()
  if context.initialized():
    value = context.get().world.broadcast_object(value, 0)
  return value


def _on_tile(method, out_shape, out_dtype, *tiles, **kw):
  """backend.<method>(*tiles) where the tile lives; a shape/dtype placeholder on the other ranks."""
  if any(isinstance(t, distarray.Absent) for t in tiles):
    return distarray.Absent(tuple(out_shape), np.dtype(out_dtype))
  return getattr(context.get().backend, method)(*tiles, **kw)


def _nearest(points, centers):
  return _on_tile('nearest_center', (points.shape[0],), np.int64, points, centers)


def _rows_of(ex):
  """Target region (rows of the point tile) in the 1-D label array."""
  return extent.create((ex.ul[0],), (ex.lr[0],), (ex.array_shape[0],))


# ---- tile bodies of the join variants (the mapper protocols of map2 / outer) ---------------------------------
def kmeans_map2_dist_mapper(extents, tiles, centers=None):
  yield _rows_of(extents[0]), _nearest(tiles[0], centers)


def kmeans_outer_dist_mapper(ex_points, points, ex_centers, centers):
  yield _rows_of(ex_points), _nearest(points, centers)


def kmeans_count_mapper(extents, tiles, centers_count):
  whole = extent.create((0,), (centers_count,), (centers_count,))
  yield whole, _on_tile('bincount', (centers_count,), np.int64, tiles[0], k=centers_count)


def kmeans_center_mapper(extents, tiles, centers_count):
  points, labels = tiles
  dim = points.shape[1]
  whole = extent.create((0, 0), (centers_count, dim), (centers_count, dim))
  yield whole, _on_tile('segment_sum', (centers_count, dim), context.get().backend.dtype_of(points),
                        points, labels, k=centers_count)


for _m in (kmeans_map2_dist_mapper, kmeans_outer_dist_mapper, kmeans_count_mapper, kmeans_center_mapper):
  _m.yields_fresh_tensors = True      # kernel outputs, never input tiles (see expr/map.join_mapper)


def _find_cluster_mapper(inputs, ex, d_pts, old_centers, new_centers, new_counts, labels):
  """Tile body of the 'shuffle' variant: all three products of one tile, pushed into reducer targets."""
  points = d_pts.fetch(ex)
  k, dim = old_centers.shape
  nearest = _nearest(points, old_centers)
  new_centers.update(extent.from_shape(new_centers.shape),
                     _on_tile('segment_sum', (k, dim), d_pts.dtype, points, nearest, k=k))
  new_counts.update(extent.from_shape(new_counts.shape),
                    _on_tile('bincount', (k,), np.int64, nearest, k=k).reshape(k, 1))
  labels.update(extent.create(ex.ul, (ex.lr[0], 1), labels.shape), nearest.reshape(points.shape[0], 1))
  return []


class KMeans(object):
  def __init__(self, n_clusters=8, n_iter=100):
    self.n_clusters = n_clusters
    self.n_iter = n_iter

  # ---- driver side of an iteration --------------------------------------------------------------------------
  def _finish(self, sums, counts):
    """Empty clusters get a fresh standard-normal center; the others the mean of their points."""
    empty = (counts == 0).reshape(self.n_clusters)
    n_empty = int(np.count_nonzero(empty))
    if n_empty:
      counts[empty] = 1
      sums[empty, :] = _replicated(lambda: np.random.randn(n_empty, sums.shape[1]))
    return sums, counts

  def _launch_join(self, X, labels, reducer):
    """The two accumulate joins of an iteration, launched; nothing waits for the device."""
    k, dim = self.n_clusters, X.shape[1]
    # (both targets as ONE tile: every worker's partial covers the whole target, so a target cut over the workers --
    #  the default, the reference's -- turns each partial into one merge per target tile: 2 x workers^2 small launches
    #  per iteration where several workers share a GPU, 1.5 MB of reduce-scatter + all-gather instead of reduce +
    #  broadcast where they do not.  Values are the same; with reducer=None the reference's own quirk -- the last
    #  partial replaces the others -- is too.)
    counts = expr.map2(labels, 0, fn=kmeans_count_mapper, fn_kw={'centers_count': k}, shape=(k,),
                       reducer=reducer, tile_hint=(k,))
    sums = expr.map2((X, labels), (0, 0), fn=kmeans_center_mapper, fn_kw={'centers_count': k},
                     shape=(k, dim), reducer=reducer, tile_hint=(k, dim))
    # Evaluated as built: the reference optimises both joins here (k_means_.py:144-145), which fuses nothing (a join
    # has no fusable body) but lets its auto-tiling pass cut the targets by rows again.
    # (the sums first: a backend's segment sum has the counts of the same labels for nothing, and hands them to the
    #  count join -- HipBackend.segment_sum / bincount inside fixed_points(); the two joins are independent)
    sums_arr = sums.evaluate()
    return counts.evaluate(), sums_arr

  def _centers_on_host(self, counts_arr, sums_arr):
    """The reference's driver step: glom both, re-seed empty clusters, divide."""
    counts, sums = counts_arr.glom(), sums_arr.glom()
    sums, counts = self._finish(sums, counts)
    return sums / counts.reshape(self.n_clusters, 1)

  def _accumulate_join(self, X, labels, reducer):
    return self._centers_on_host(*self._launch_join(X, labels, reducer))

  def _centers_on_worker(self, counts_arr, sums_arr):
    """(centers, check) with the division sums / counts done where the sums are -- NumPy's own arithmetic on the
    backend's tiles, float32 / int64 -> float64 like the host's -- so that the centers never visit the driver; only
    the k counts do, and LATER: `check()` says whether every cluster had points (an empty one is re-seeded from the
    driver's random stream, which takes the host route).  (None, None) if the tiles are not plain."""
    k = self.n_clusters
    be = context.get().backend
    count_t = counts_arr.fetch(extent.from_shape(counts_arr.shape))     # replicated: every rank holds all of it
    sum_t = sums_arr.fetch(extent.from_shape(sums_arr.shape))
    if any(type(t).__name__ in ('MaskedBlob', 'EmptyBlob') for t in (count_t, sum_t)):
      return None, None
    with np.errstate(all='ignore'):                # (an empty cluster divides by zero: that result is not used)
      centers = sum_t / count_t.reshape(k, 1)
    later = getattr(be, 'to_numpy_later', None)
    if later is None or isinstance(count_t, np.ndarray):
      ok = not np.any(be.to_numpy(count_t) == 0)
      return centers, (lambda: ok)
    handle = later(count_t)
    return centers, (lambda: not np.any(handle.get() == 0))

  def _fit_map2(self, X, centers, reducer):
    """The 'map2' loop with the centers kept on the workers between iterations and the empty-cluster check of an
    iteration made one iteration LATE: iteration i + 1 is launched with the centers iteration i produced before the
    driver has seen i's counts, so the device queue never drains (configs[3]: the chain download 1 MB -> divide on
    the host -> upload 2 MB -> build and launch cost 0.43 of 6.1 ms per iteration with the device idle).  If the late
    check finds an empty cluster -- rare -- what was launched on those centers is dropped and the loop continues from
    the reference's host step for iteration i: same centers, same draws from the driver's random stream, same order."""
    labels = None
    checked = None          # (check, counts_arr, sums_arr) of the iteration whose centers are in use, not yet verified
    it = 0
    while it < self.n_iter:
      labels_try = expr.map2(X, 0, fn=kmeans_map2_dist_mapper, fn_kw={'centers': centers}, shape=(X.shape[0],))
      counts_arr, sums_arr = self._launch_join(X, labels_try, reducer)
      if checked is not None and not checked[0]():
        # iteration it - 1 had an empty cluster: redo its driver step on the host, then this iteration again
        centers = self._centers_on_host(checked[1], checked[2])
        checked = None
        continue
      labels = labels_try
      ahead, check = self._centers_on_worker(counts_arr, sums_arr)
      if ahead is None:
        centers, checked = self._centers_on_host(counts_arr, sums_arr), None
      else:
        centers, checked = ahead, (check, counts_arr, sums_arr)
      it += 1
    if checked is not None and not checked[0]():
      centers = self._centers_on_host(checked[1], checked[2])
    if not isinstance(centers, np.ndarray):
      centers = context.get().backend.to_numpy(centers)
    return centers, labels

  # ---- one iteration per implementation: (X, centers) -> (centers, labels) -----------------------------------
  def _step_map2(self, X, centers, reducer):
    labels = expr.map2(X, 0, fn=kmeans_map2_dist_mapper, fn_kw={'centers': centers}, shape=(X.shape[0],))
    return self._accumulate_join(X, labels, reducer), labels

  def _step_outer(self, X, centers, reducer):
    labels = expr.outer((X, centers), (0, None), fn=kmeans_outer_dist_mapper, shape=(X.shape[0],))
    return expr.from_numpy(self._accumulate_join(X, labels, reducer)), labels

  def _step_broadcast(self, X, centers, reducer):
    k, dim = centers.shape
    x3 = expr.reshape(X, (X.shape[0], 1, dim))
    c3 = expr.reshape(centers, (1, k, dim))
    labels = expr.argmin(expr.sum(expr.square(x3 - c3), axis=2), axis=1)
    onehot = (expr.reshape(labels, (labels.shape[0], 1)) == expr.arange((1, k))).astype(np.int64)
    counts = expr.sum(onehot, axis=0)
    sums = expr.sum(x3 * expr.reshape(onehot, (onehot.shape[0], k, 1)), axis=0)
    counts = counts.optimized().glom()
    sums = sums.optimized().glom()
    sums, counts = self._finish(sums, counts)
    return expr.from_numpy(sums / counts.reshape(k, 1)), labels

  def _step_shuffle(self, X, centers, labels):
    k, dim = self.n_clusters, X.shape[1]
    sums = expr.ndarray((k, dim), reduce_fn=np.add)
    counts = expr.ndarray((k, 1), dtype=np.int64, reduce_fn=np.add)
    expr.shuffle(X, _find_cluster_mapper,
                 kw={'d_pts': X, 'old_centers': centers, 'new_centers': sums, 'new_counts': counts,
                     'labels': labels},
                 shape_hint=(1,),
                 cost_hint={hash(labels): {'00': 0, '01': np.prod(labels.shape)}}).evaluate()
    sums, counts = self._finish(sums.glom(), counts.glom())
    return sums / counts

  def fit(self, X, centers=None, implementation='map2', reducer=None):
    """Run `n_iter` iterations from `centers` (uniform random when None).

    X: expression / array of shape (n_samples, n_features), tiled by rows.
    centers: start centers; a NumPy array for 'map2' / 'shuffle', an expression for 'outer' / 'broadcast'.
    reducer: how the join variants combine the per-tile counts and sums (None: the reference's behaviour,
      see the module docstring; np.add: the real update).
    Returns (centers, labels); `labels` is an expression of the last assignment.
    """
    # the points are not written between the iterations of a fit: a backend may derive what it needs of them once
    fixed = getattr(context.get().backend, 'fixed_points', None) if context.initialized() else None
    if fixed is not None:
      with fixed():
        return self._fit(X, centers, implementation, reducer)
    return self._fit(X, centers, implementation, reducer)

  def _fit(self, X, centers, implementation, reducer):
    k, dim = self.n_clusters, X.shape[1]
    labels = expr.zeros((X.shape[0], 1), dtype=np.int64)
    if implementation in ('map2', 'shuffle'):
      if centers is None:
        centers = _replicated(lambda: np.random.rand(k, dim))
      if implementation == 'map2':
        return self._fit_map2(X, centers, reducer)
      for _ in range(self.n_iter):
        centers = self._step_shuffle(X, centers, labels)
      return centers, labels
    if implementation in ('outer', 'broadcast'):
      if centers is None:
        centers = expr.rand(k, dim)
      step = self._step_outer if implementation == 'outer' else self._step_broadcast
      for _ in range(self.n_iter):
        centers, labels = step(X, centers, reducer)
      return centers, labels
    raise ValueError('unknown implementation %r' % (implementation,))
