"""Linear regression by SGD (reference spartan/examples/linear_regression.py:6-28).
Per step and tile: `dot(x, w)` is one fused multiply-reduce pass over X (matrix.vector,
HBM-bound), `sum(x * (yp - y), axis=0)` one fused map->column-reduce pass."""
from .. import expr
from . import sgd


class LinearRegression(sgd.SGDRegressor):
  def __init__(self, x, y, iterations, alpha=1e-6):
    super(LinearRegression, self).__init__(x, y, iterations, alpha)

  def update(self):
    """gradient_update = (h(w) - y) * x,  h(w) = x * w  (linear_regression.py:10-16)."""
    yp = expr.dot(self.x, self.w)
    return self.x * (yp - self.y)


def linear_regression(x, y, iterations):
  lreg = LinearRegression(x, y, iterations)
  return lreg.train()


def run(N_EXAMPLES, N_DIM, iterations):
  x = expr.rand(N_EXAMPLES, N_DIM)
  y = expr.rand(N_EXAMPLES, 1)
  return linear_regression(x, y, iterations)
