"""Jacobi iteration: mirror of the reference's spartan/examples/jacobi.py."""
from .. import expr, util


def jacobi_init(size):
  """jacobi.py:6-27."""
  av = expr.arange(start=2, stop=size + 2)
  bv = expr.arange(start=4, stop=size + 4).reshape((size, 1))
  A = av * bv
  return A, A[:, -1:].reshape((size,))


def jacobi_method(A, b, _iter=100):
  """jacobi.py:29-56: x <- (b - R x) / D with D = diag(A), R = A - diagflat(D)."""
  util.Assert.eq(A.shape[0], b.shape[0])
  x = expr.zeros((A.shape[0],))
  D = expr.diag(A)
  R = A - expr.diagflat(D)
  for _ in range(_iter):
    x = (b - expr.dot(R, x)) / D
  return x
