"""SGDRegressor: W = W - alpha * sum(gradient_update, axis=0).
Mirror of the reference's spartan/examples/sgd.py:6-40."""
import numpy as np

from .. import expr


class SGDRegressor(object):
  def __init__(self, x, y, iterations, alpha=1e-6):
    """x, y: Exprs (N, D) and (N, 1); sgd.py:14-30 (w drawn with np.random.rand)."""
    self.x = x
    self.y = y
    self.iterations = iterations
    self.alpha = alpha
    self.N_DIM = self.x.shape[1]
    self.w = np.random.rand(self.N_DIM, 1)

  def update(self):
    raise NotImplementedError("Should be overrided by the child regression")

  def train(self):
    """sgd.py:35-40: one fused map->reduce launch per tile and step, (D,) gathered to the driver."""
    for i in range(self.iterations):
      diff = self.update()
      grad = expr.sum(diff, axis=0).optimized().glom().reshape((self.N_DIM, 1))
      self.w = self.w - grad * self.alpha
    return self.w
