"""Logistic regression by SGD (reference spartan/examples/logistic_regression.py:6-28)."""
from .. import expr
from . import sgd


class LogisticRegression(sgd.SGDRegressor):
  def __init__(self, x, y, iterations, alpha=1e-6):
    super(LogisticRegression, self).__init__(x, y, iterations, alpha)

  def update(self):
    """gradient_update = (h(w) - y) * x,  h(w) = 1 / (1 + e^-(x*w))  (logistic_regression.py:10-17)."""
    g = expr.exp(expr.dot(self.x, self.w))
    yp = g / (g + 1)
    return self.x * (yp - self.y)


def logistic_regression(x, y, iterations):
  logreg = LogisticRegression(x, y, iterations)
  return logreg.train()


def run(N_EXAMPLES, N_DIM, iterations):
  x = expr.rand(N_EXAMPLES, N_DIM)
  y = expr.rand(N_EXAMPLES, 1)
  return logistic_regression(x, y, iterations)
