"""Small helpers mirroring the bits of spartan/util.py the tile path uses."""
import math

import numpy as np


def divup(a, b):
  """Reference spartan/util.py:404-408 (float ceil, kept bit-for-bit)."""
  if isinstance(a, tuple):
    return tuple([divup(ta, b) for ta in a])
  return int(math.ceil(float(a) / b))


def is_iterable(x):
  return hasattr(x, '__iter__') and not isinstance(x, (str, bytes, np.ndarray))


class Assert(object):
  """Subset of spartan/util.py:222-326 (same failure type: AssertionError)."""

  @staticmethod
  def eq(a, b, msg='', *args):
    same = a == b
    if same is True:
      return
    if not np.all(np.asarray(same)):
      raise AssertionError('%s != %s %s' % (a, b, (msg % args) if args else msg))

  @staticmethod
  def le(a, b, msg=''):
    if not a <= b:
      raise AssertionError('%s > %s %s' % (a, b, msg))

  @staticmethod
  def isinstance(v, types):
    if not isinstance(v, types):
      raise AssertionError('%s (%s) is not an instance of %s' % (v, type(v), types))

  @staticmethod
  def not_null(v):
    if v is None:
      raise AssertionError('unexpected None')

  @staticmethod
  def no_duplicates(seq):
    seq = list(seq)
    if len(set(seq)) != len(seq):
      raise AssertionError('duplicates in %s' % (seq,))

  @staticmethod
  def all_eq(a, b, tolerance=0):
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape:
      raise AssertionError('shape mismatch %s vs %s' % (a.shape, b.shape))
    if tolerance == 0:
      ok = np.all(a == b)
    else:
      ok = np.all(np.abs(a.astype(np.float64) - b.astype(np.float64)) <= tolerance)
    if not ok:
      raise AssertionError('arrays differ:\n%s\n%s' % (a, b))

  @staticmethod
  def all_close(a, b):
    if not np.allclose(a, b):
      raise AssertionError('arrays not close')
