"""Conjugate gradient (NAS CG style): mirror of the reference's spartan/examples/conj_gradient.py.
Per iteration: one matrix x vector product (dense GEMV-like reduce, or sp_csr_spmm when A is sparse), three fused
map -> sum reductions and three fused maps."""
from .. import expr


def cgit(A, x):
  """conj_gradient.py:4-30: 15 CG iterations for A z = x."""
  z = expr.zeros(x.shape)
  r = x
  rho = expr.sum(r * r).optimized().glom()
  p = r
  for _ in range(15):
    q = expr.dot(A, p)
    alpha = rho / expr.sum(p * q).optimized().glom()
    z = z + p * alpha
    rho0 = rho
    r = r - q * alpha
    rho = expr.sum(r * r).optimized().glom()
    beta = rho / rho0
    p = r + p * beta
  return z


def conj_gradient(A, num_iter=15):
  """conj_gradient.py:32-51."""
  x = expr.ones((A.shape[1], 1))
  for _ in range(num_iter):
    z = cgit(A, x)
    x = z / expr.norm(z)
  return x
