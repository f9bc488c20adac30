"""K-Means (reference spartan/examples/sklearn/cluster/k_means_.py), BASELINE configs[3].

The per-tile bodies of the reference's mappers are HIP kernels behind the backend:

  np.argmin(cdist(points, centers), axis=1)   -> backend.nearest_center  (sp_nearest_center:
        fp32 MFMA GEMM with the argmin fused into the epilogue, exact fp64 re-check of near ties)
  np.bincount(labels, minlength=k)            -> backend.bincount        (sp_bincount_i64)
  new_centers[i] = points[labels == i].sum(0) -> backend.segment_sum     (sp_segment_sum)

`KMeans.fit` keeps the reference's four implementations and their argument meaning.
One faithful quirk is kept as the default and made switchable: the reference's map2 /
outer variants create the `counts` / `new_centers` targets WITHOUT a reducer
(k_means_.py:133-141), so with more than one tile the value of the last tile to arrive
replaces the others instead of being added.  `fit(..., reducer=np.add)` combines the
per-tile partials (reduce / reduce-scatter over RCCL), which is the algorithm the
'shuffle' variant implements with its `reduce_fn` (k_means_.py:236-239).
"""
import numpy as np

from .... import context, expr
from ....array import distarray, extent


def _tile_op(method, out_shape, out_dtype, *tiles, **kw):
  """backend.<method>(*tiles) on the executing rank, an Absent placeholder elsewhere."""
  if any(isinstance(t, distarray.Absent) for t in tiles):
    return distarray.Absent(tuple(out_shape), np.dtype(out_dtype))
  return getattr(context.get().backend, method)(*tiles, **kw)


def _find_closest(pts, centers):
  """k_means_.py:11-28 (first strict minimum) == argmin of the squared distances."""
  return _tile_op('nearest_center', (pts.shape[0],), np.int64, pts, centers)


def _find_cluster_mapper(inputs, ex, d_pts, old_centers, new_centers, new_counts, labels):
  """k_means_.py:31-49: the 'shuffle' variant's per-tile body."""
  centers = old_centers
  pts = d_pts.fetch(ex)
  k = centers.shape[0]
  closest = _find_closest(pts, centers)
  l_counts = _tile_op('bincount', (k,), np.int64, closest, k=k).reshape(k, 1)
  l_centers = _tile_op('segment_sum', (k, centers.shape[1]), d_pts.dtype, pts, closest, k=k)
  new_centers.update(extent.from_shape(new_centers.shape), l_centers)
  new_counts.update(extent.from_shape(new_counts.shape), l_counts)
  labels.update(extent.create(ex.ul, (ex.lr[0], 1), labels.shape), closest.reshape(pts.shape[0], 1))
  return []


def kmeans_outer_dist_mapper(ex_a, tile_a, ex_b, tile_b):
  """k_means_.py:52-58."""
  target_ex = extent.create((ex_a[0].ul[0],), (ex_a[0].lr[0],), (ex_a[0].array_shape[0],))
  yield target_ex, _tile_op('nearest_center', (tile_a.shape[0],), np.int64, tile_a, tile_b)


def kmeans_map2_dist_mapper(ex, tile, centers=None):
  """k_means_.py:61-66."""
  points = tile[0]
  target_ex = extent.create((ex[0].ul[0],), (ex[0].lr[0],), (ex[0].array_shape[0],))
  yield target_ex, _tile_op('nearest_center', (points.shape[0],), np.int64, points, centers)


def kmeans_count_mapper(extents, tiles, centers_count):
  """k_means_.py:69-72."""
  target_ex = extent.create((0,), (centers_count,), (centers_count,))
  yield target_ex, _tile_op('bincount', (centers_count,), np.int64, tiles[0], k=centers_count)


def kmeans_center_mapper(extents, tiles, centers_count):
  """k_means_.py:75-97."""
  points, labels = tiles[0], tiles[1]
  target_ex = extent.create((0, 0), (centers_count, points.shape[1]), (centers_count, points.shape[1]))
  yield target_ex, _tile_op('segment_sum', (centers_count, points.shape[1]),
                            context.get().backend.dtype_of(points), points, labels, k=centers_count)


class KMeans(object):
  def __init__(self, n_clusters=8, n_iter=100):
    """k_means_.py:100-115."""
    self.n_clusters = n_clusters
    self.n_iter = n_iter

  def _reseed_and_divide(self, centers, counts, num_dim):
    """k_means_.py:145-157: empty clusters are re-seeded from randn, then sums / counts."""
    zcount_indices = (counts == 0).reshape(self.n_clusters)
    if np.any(zcount_indices):
      n_points = np.count_nonzero(zcount_indices)
      counts[zcount_indices] = 1
      centers[zcount_indices, :] = np.random.randn(n_points, num_dim)
    return centers / counts.reshape(centers.shape[0], 1)

  def fit(self, X, centers=None, implementation='map2', reducer=None):
    """Compute k-means clustering (k_means_.py:117-267).

    X: spartan matrix (n_samples, n_features), tiled by rows.
    centers: initial centers (numpy.ndarray); random if None.
    reducer: combine function of the per-tile counts / center sums of the 'map2' and
      'outer' variants; None reproduces the reference (see the module docstring).
    Returns (centers, labels).
    """
    num_dim = X.shape[1]
    num_points = X.shape[0]
    labels = expr.zeros((num_points, 1), dtype=np.int64)

    if implementation == 'map2':
      if centers is None:
        centers = np.random.rand(self.n_clusters, num_dim)
      for i in range(self.n_iter):
        labels = expr.map2(X, 0, fn=kmeans_map2_dist_mapper, fn_kw={"centers": centers},
                           shape=(X.shape[0],))
        counts = expr.map2(labels, 0, fn=kmeans_count_mapper,
                           fn_kw={'centers_count': self.n_clusters},
                           shape=(centers.shape[0],), reducer=reducer)
        new_centers = expr.map2((X, labels), (0, 0), fn=kmeans_center_mapper,
                                fn_kw={'centers_count': self.n_clusters},
                                shape=(centers.shape[0], centers.shape[1]), reducer=reducer)
        counts = counts.optimized().glom()
        centers = new_centers.optimized().glom()
        centers = self._reseed_and_divide(centers, counts, num_dim)
      return centers, labels

    elif implementation == 'outer':
      if centers is None:
        centers = expr.rand(self.n_clusters, num_dim)
      for i in range(self.n_iter):
        labels = expr.outer((X, centers), (0, None), fn=kmeans_outer_dist_mapper,
                            shape=(X.shape[0],))
        counts = expr.map2(labels, 0, fn=kmeans_count_mapper,
                           fn_kw={'centers_count': self.n_clusters},
                           shape=(centers.shape[0],), reducer=reducer)
        new_centers = expr.map2((X, labels), (0, 0), fn=kmeans_center_mapper,
                                fn_kw={'centers_count': self.n_clusters},
                                shape=(centers.shape[0], centers.shape[1]), reducer=reducer)
        counts = counts.optimized().glom()
        centers = new_centers.optimized().glom()
        centers = self._reseed_and_divide(centers, counts, num_dim)
        centers = expr.from_numpy(centers)
      return centers, labels

    elif implementation == 'broadcast':
      if centers is None:
        centers = expr.rand(self.n_clusters, num_dim)
      for i in range(self.n_iter):
        X_broadcast = expr.reshape(X, (X.shape[0], 1, X.shape[1]))
        centers_broadcast = expr.reshape(centers, (1, centers.shape[0], centers.shape[1]))
        distances = expr.sum(expr.square(X_broadcast - centers_broadcast), axis=2)
        labels = expr.argmin(distances, axis=1)
        center_idx = expr.arange((1, centers.shape[0]))
        matches = expr.reshape(labels, (labels.shape[0], 1)) == center_idx
        matches = matches.astype(np.int64)
        counts = expr.sum(matches, axis=0)
        centers = expr.sum(X_broadcast * expr.reshape(matches, (matches.shape[0], matches.shape[1], 1)),
                           axis=0)
        counts = counts.optimized().glom()
        centers = centers.optimized().glom()
        centers = self._reseed_and_divide(centers, counts, num_dim)
        centers = expr.from_numpy(centers)
      return centers, labels

    elif implementation == 'shuffle':
      if centers is None:
        centers = np.random.rand(self.n_clusters, num_dim)
      for i in range(self.n_iter):
        # (the reference passes `lambda a, b: a + b`; np.add is the same function with a combine kernel)
        new_centers = expr.ndarray((self.n_clusters, num_dim), reduce_fn=np.add)
        new_counts = expr.ndarray((self.n_clusters, 1), dtype=np.int64, reduce_fn=np.add)
        _ = expr.shuffle(X, _find_cluster_mapper,
                         kw={'d_pts': X, 'old_centers': centers, 'new_centers': new_centers,
                             'new_counts': new_counts, 'labels': labels},
                         shape_hint=(1,),
                         cost_hint={hash(labels): {'00': 0, '01': np.prod(labels.shape)}})
        _.evaluate()
        new_counts = new_counts.glom()
        new_centers = new_centers.glom()
        zcount_indices = (new_counts == 0).reshape(self.n_clusters)
        if np.any(zcount_indices):
          n_points = np.count_nonzero(zcount_indices)
          new_counts[zcount_indices] = 1
          new_centers[zcount_indices, :] = np.random.randn(n_points, num_dim)
        new_centers = new_centers / new_counts
        centers = new_centers
      return centers, labels

    raise ValueError('unknown implementation %r' % (implementation,))
