"""Least-squares fit by full-batch gradient steps: the workload of BASELINE configs[4]
(reference: tests/benchmark_lreg.py:8-32 drives spartan/examples/linear_regression.py).

One step, for X (N, D) row-tiled over the workers, y (N, 1) and driver-side weights w (D, 1):

    r    = X . w - y          dot with a driver array: every row tile computes its own rows of X.w
                              (one multiply-reduce launch per tile, w broadcast, no combine)
    g    = sum(X * r, axis=0) fused map -> column reduce, one launch per tile; the (D,) partials of
                              the tiles are combined by reduce-scatter, the result is all-gathered
    w   -= alpha * g          every rank holds the same w -- as the backend's tensor between steps (see _whole)

so X is streamed twice per step as stated; on the HIP backend the optimizer rewrites the gradient's DAG into ONE
pass over X (expr/rowdot.py: the rows stay in registers between the two uses, sp_rowdot_colsum_f32) -- the same
sums in another order.  Nothing of size N ever leaves HBM.  The arithmetic is the
reference's (`w - grad * alpha` with grad the glommed float sum), which the committed goldens pin.
"""
import numpy as np

from .. import context, expr
from ..array import extent


def initial_weights(n_features):
  """The reference draws the start vector from the driver's np.random stream.  With one process per GPU
  every rank runs the driver, so rank 0 draws and the others receive it: all ranks step the same w."""
  w = np.random.rand(n_features, 1)
  if context.initialized():
    w = context.get().world.broadcast_object(w, 0)
  return w


def gradient(x, y, w):
  """Expr of shape (D,): sum over the rows of x * (x.w - y)."""
  residual = expr.dot(x, w) - y
  return expr.sum(x * residual, axis=0)


def fit(x, y, steps, alpha=1e-6, w=None):
  """`steps` gradient steps from `w` (drawn by initial_weights when None); returns w, shape (D, 1)."""
  n_features = x.shape[1]
  if w is None:
    w = initial_weights(n_features)
  be = context.get().backend if context.initialized() else None
  for _ in range(steps):
    g = _whole(gradient(x, y, w).optimized().evaluate())
    w = w - g.reshape((n_features, 1)) * alpha
  if not isinstance(w, np.ndarray):
    w = be.to_numpy(w)
  return w


def _whole(array):
  """The (D,) gradient as ONE tensor where this rank keeps its tiles -- the reference gloms it to the driver
  (sgd.py:37) and updates `w` there with NumPy; here NumPy's same arithmetic (`w - g * alpha`: float32 * Python float,
  then the subtraction in w's precision) runs on the backend's tensors, so that the weights of the next step are
  where the tiles are and no step waits for a transfer: on the HIP backend the loop is then bound by its kernels
  (0.49 -> 0.35 ms per step on BASELINE configs[4]'s per-GPU tile).  On a NumPy backend this IS the glom."""
  return array.fetch(extent.from_shape(array.shape))


def run(n_rows, n_features, steps):
  """benchmark_lreg.py's shape of program: uniform random x and y generated on the workers."""
  return fit(expr.rand(n_rows, n_features), expr.rand(n_rows, 1), steps)
