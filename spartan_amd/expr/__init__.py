"""The expression layer.  `spartan_amd.expr` doubles as the reference's flat builder namespace (`expr.rand`, `expr.dot`,
...: spartan/expr/__init__.py:26-93): spartan_amd/__init__.py re-exports its public names here, and names of the
operators outside the default import resolve on first use."""


def __getattr__(name):
  import importlib
  top = importlib.import_module(__name__.rsplit('.', 1)[0])
  if name in getattr(top, '_EXTRAS', ()):
    return getattr(top, name)
  raise AttributeError('module %r has no attribute %r' % (__name__, name))
