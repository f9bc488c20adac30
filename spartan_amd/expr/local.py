"""Local expressions: the per-tile kernel IR.

Same node types as the reference's spartan/expr/operator/local.py.  A fused
tree of these (after spartan_amd.expr.optimize) is the *specification* of one
HIP kernel launch; the backend lowers it (spartan_amd/lower.py).  There is no
NumPy `evaluate` here: the product path has no CPU fallback (the NumPy
evaluation of local.py:115-127 lives in oracle/ as test infrastructure).
"""
import itertools

_var_id = itertools.count()


def make_var():
  """local.py:30-32."""
  return 'key_%d' % next(_var_id)


def indent(s):
  return s.replace('\n', '\n  ')


class LocalExpr(object):
  """local.py:39-55."""

  def __init__(self, deps=None):
    self.deps = list(deps) if deps is not None else []

  def __repr__(self):
    return self.pretty_str()

  def add_dep(self, v):
    self.deps.append(v)

  def input_names(self):
    out = []
    for d in self.deps:
      for n in d.input_names():
        if n not in out:
          out.append(n)
    return out


class LocalInput(LocalExpr):
  """local.py:58-73: an externally supplied input (a fetched tile, 'extent', 'axis')."""

  def __init__(self, idx):
    LocalExpr.__init__(self)
    assert idx != ''
    self.idx = idx

  def pretty_str(self):
    return '%s' % self.idx

  def input_names(self):
    return [self.idx]


class FnCallExpr(LocalExpr):
  """local.py:76-127."""

  def __init__(self, fn, kw=None, pretty_fn=None, deps=None):
    LocalExpr.__init__(self, deps)
    assert fn is not None
    self.fn = fn
    self.kw = kw if kw is not None else {}
    self.pretty_fn = pretty_fn

  def fn_name(self):
    if self.pretty_fn:
      return self.pretty_fn
    if hasattr(self.fn, '__module__') and hasattr(self.fn, '__name__'):
      return '%s.%s' % (self.fn.__module__, self.fn.__name__)
    return getattr(self.fn, '__name__', repr(self.fn))

  def pretty_str(self):
    # local.py:94-100
    pretty_fn = self.fn_name().split('.')[-1]
    return '%s(%s,kw=%s)' % (
        pretty_fn,
        indent(','.join([v.pretty_str() for v in self.deps if not isinstance(v, LocalInput)])),
        indent(','.join(['(k=%s v=%s)' % (k, v) for k, v in self.kw.items()])))


class LocalMapExpr(FnCallExpr):
  _op_type = 'map'


class LocalMapLocationExpr(LocalMapExpr):
  """local.py:137-149: the mapper also receives the extent as (ul, lr, array_shape)."""
  _op_type = 'map_location'


class LocalReduceExpr(FnCallExpr):
  _op_type = 'reduce'
