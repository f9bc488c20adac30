"""Ridge regression by SGD (reference spartan/examples/ridge_regression.py:7-33):
the x^T x product runs on the MFMA GEMM through the Transpose view."""
from .. import expr
from . import sgd


class RidgeRegression(sgd.SGDRegressor):
  def __init__(self, x, y, ridge_lambda, iterations, alpha=1e-6):
    super(RidgeRegression, self).__init__(x, y, iterations, alpha)
    self.ridge_lambda = ridge_lambda

  def update(self):
    """gradient_update = xTxw + xTy + lambda * w, as written at ridge_regression.py:12-22."""
    xT = expr.transpose(self.x)
    g1 = expr.dot(expr.dot(xT, self.x), self.w)
    g2 = expr.dot(xT, self.y)
    g3 = self.ridge_lambda * self.w
    g4 = (g1 + g2 + g3)
    return expr.reshape(g4, (1, self.N_DIM))


def ridge_regression(x, y, ridge_lambda, iterations):
  ridge_reg = RidgeRegression(x, y, ridge_lambda, iterations)
  return ridge_reg.train()


def run(N_EXAMPLES, N_DIM, iterations):
  x = expr.rand(N_EXAMPLES, N_DIM)
  y = expr.rand(N_EXAMPLES, 1)
  return ridge_regression(x, y, 1, iterations)
