"""The one-pass gradient rewrite (spartan_amd/expr/rowdot.py, optimize.RowDotColSumFusion): which DAGs it takes, that
it never fires on a backend without the kernel, and that the rewritten node gives the reduction's result -- host logic
on the NumPy oracle backend with a stand-in for the kernel; the kernel itself is tested on the GPU."""
import numpy as np
import pytest

import spartan_amd as sp
from oracle.np_backend import NumpyBackend
from spartan_amd.examples import lreg
import importlib

optimize = importlib.import_module('spartan_amd.expr.optimize')     # (the package attribute is the function)
from spartan_amd.expr.rowdot import RowDotColSumExpr


class _WithKernel(NumpyBackend):
  """The oracle backend plus a NumPy statement of sp_rowdot_colsum_f32 (test stand-in for the HIP kernel)."""
  calls = 0

  def rowdot_colsum(self, x, w, y):
    type(self).calls += 1
    t = x.astype(np.float32).dot(np.asarray(w, np.float32).reshape(-1, 1))
    r = t if y is None else t - np.asarray(y).reshape(t.shape)
    return (x * r).sum(0).astype(np.float32)


def _data(n=203, d=64, seed=0):
  rng = np.random.RandomState(seed)
  return rng.rand(n, d).astype(np.float32), rng.rand(n, 1).astype(np.float32), rng.rand(d, 1).astype(np.float32)


@pytest.mark.parametrize('workers', [1, 4])
def test_gradient_dag_is_rewritten_and_gives_the_same_values(workers):
  xh, yh, w = _data()
  want = (xh * (xh.dot(w) - yh)).sum(0)
  sp.initialize(backend=_WithKernel(), num_workers=workers)
  try:
    _WithKernel.calls = 0
    x, y = sp.Val(val=sp.from_numpy(xh).force()), sp.Val(val=sp.from_numpy(yh).force())
    g = lreg.gradient(x, y, w).optimized()
    assert isinstance(g, RowDotColSumExpr)
    got = g.glom()
    assert got.shape == (64,) and got.dtype == np.float32
    np.testing.assert_allclose(got, want, rtol=2e-5)
    assert _WithKernel.calls == len(x.val.tiles)                 # one call per row tile
    g2 = sp.sum(x * sp.dot(x, w), axis=0).optimized()            # without y
    assert isinstance(g2, RowDotColSumExpr)
    np.testing.assert_allclose(g2.glom(), (xh * xh.dot(w)).sum(0), rtol=2e-5)
    # a whole fit through the rewrite equals the fit without it to rounding
    w1 = lreg.fit(x, y, 5, alpha=1e-3, w=w)
    optimize.FLAGS['opt_rowdot_fusion'] = False
    try:
      w0 = lreg.fit(x, y, 5, alpha=1e-3, w=w)
    finally:
      optimize.FLAGS['opt_rowdot_fusion'] = True
    np.testing.assert_allclose(w1, w0, rtol=1e-5, atol=1e-6)
  finally:
    sp.shutdown()


def test_rewrite_leaves_everything_else_alone():
  xh, yh, w = _data(d=64)
  sp.initialize(backend=_WithKernel(), num_workers=2)
  try:
    x, y = sp.Val(val=sp.from_numpy(xh).force()), sp.Val(val=sp.from_numpy(yh).force())
    keep = [
        sp.sum(x * (sp.dot(x, w) - y), axis=1),                            # row sums
        sp.sum(x * (sp.dot(x, w.astype(np.float64)) - y), axis=0),         # a float64 weight vector
        sp.sum(x + (sp.dot(x, w) - y), axis=0),                            # not a product
        sp.sum(x * (sp.dot(x, w) + y), axis=0),                            # not a difference
        sp.max(x * (sp.dot(x, w) - y), axis=0),                            # not a sum
    ]
    x2 = sp.Val(val=sp.from_numpy(xh).force())
    keep.append(sp.sum(x2 * (sp.dot(x, w) - y), axis=0))                   # the product's x is another array
    xc = sp.Val(val=sp.from_numpy(xh, tile_hint=(203, 16)).force())
    keep.append(sp.sum(xc * (sp.dot(xc, w) - y), axis=0))                  # x tiled by columns
    x6 = sp.Val(val=sp.from_numpy(xh[:, :62].copy()).force())
    keep.append(sp.sum(x6 * (sp.dot(x6, w[:62]) - y), axis=0))             # 62 columns: not a multiple of 4
    for e in keep:
      assert not isinstance(e.optimized(), RowDotColSumExpr), e
    np.testing.assert_allclose(keep[0].glom(), (xh * (xh.dot(w) - yh)).sum(1), rtol=2e-5)
  finally:
    sp.shutdown()


def test_rewrite_needs_the_backend_kernel():
  xh, yh, w = _data()
  sp.initialize(backend=NumpyBackend(), num_workers=2)
  try:
    x, y = sp.Val(val=sp.from_numpy(xh).force()), sp.Val(val=sp.from_numpy(yh).force())
    assert not isinstance(lreg.gradient(x, y, w).optimized(), RowDotColSumExpr)
  finally:
    sp.shutdown()
