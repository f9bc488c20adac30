"""TileExtent and the integer index arithmetic of the tile path.

Host-side mirror of the reference's Cython module spartan/array/extent.pyx
(same names, argument meaning and None/assert behaviour); every function cites
the lines it follows.  All arithmetic is on Python ints: results are bit-exact
with the reference (which uses int64 coordinates, extent.pyx:19-22), including
its Python-2 floor division (extent.pyx:207,242,523,547).
"""
import math

import numpy as np

from ..util import Assert, divup

MAX_DIM = 32  # extent.pyx:21-22


class TileExtent(object):
  """Half-open box [ul, lr) inside an array of `array_shape` (extent.pyx:23-136)."""
  __slots__ = ('ul', 'lr', 'array_shape', '_shape')

  def __init__(self, ul, lr, array_shape):
    # (an extent never changes: the tuples are taken as they are when they already are tuples of ints -- `create`
    #  and the algebra below build them that way -- and the derived shape is computed once)
    self.ul = ul if type(ul) is tuple and all(type(v) is int for v in ul) else tuple(int(v) for v in ul)
    self.lr = lr if type(lr) is tuple and all(type(v) is int for v in lr) else tuple(int(v) for v in lr)
    if array_shape is None or (type(array_shape) is tuple and all(type(v) is int for v in array_shape)):
      self.array_shape = array_shape
    else:
      self.array_shape = tuple(int(v) for v in array_shape)
    self._shape = None

  @property
  def size(self):
    return math.prod(self.shape)

  @property
  def shape(self):
    # extent.pyx:66-72 -- a zero-length dimension reports 1
    shape = self._shape
    if shape is None:
      shape = self._shape = tuple([1 if (l - u) == 0 else (l - u) for u, l in zip(self.ul, self.lr)])
    return shape

  @property
  def ndim(self):
    return len(self.ul)

  def to_slice(self):
    return tuple(slice(u, l) for u, l in zip(self.ul, self.lr))

  def to_tuple(self):
    return (self.ul, self.lr, self.array_shape)

  def __getitem__(self, axis):
    """The 1-D extent of one axis (negative axes count from the end, like the reference's `ul[idx]`)."""
    axis %= len(self.ul)
    return create(self.ul[axis:axis + 1], self.lr[axis:axis + 1], self.array_shape[axis:axis + 1])

  def __repr__(self):
    return 'extent(%s)' % ','.join('%s:%s' % bounds for bounds in zip(self.ul, self.lr))

  def __hash__(self):
    return hash(self.ul)  # (by corner only, like extent.pyx:93-94: equal extents hash alike either way)

  def __eq__(self, other):
    return isinstance(other, TileExtent) and self.ul == other.ul and self.lr == other.lr

  def __ne__(self, other):
    return not self.__eq__(other)

  def __lt__(self, other):
    # extent.pyx:96-105: lexicographic on ul, "smaller" defaults to True on ties
    for a, b in zip(self.ul, other.ul):
      if a < b:
        return True
      if a > b:
        return False
    return True

  def __gt__(self, other):
    return not self.__lt__(other)

  def to_global(self, idx, axis):
    """A tile-local index as an index of the whole array: along `axis`, or (axis None) a flat index into the
    tile as the flat index of the same cell in the array (extent.pyx:121-127)."""
    if axis is None:
      inside = unravelled_pos(idx, self.shape)
      return ravelled_pos(tuple(u + i for u, i in zip(self.ul, inside)), self.array_shape)
    return idx + self.ul[axis]

  def ravelled_pos(self):
    return ravelled_pos(self.ul, self.array_shape)

  def add_dim(self):
    return create(self.ul + (0,), self.lr + (1,), self.array_shape + (1,))

  def clone(self):
    return create(self.ul, self.lr, self.array_shape)


def create(ul, lr, array_shape):
  """extent.pyx:141-182 -- returns None for an unrealistic box (any ul >= lr)."""
  ul = tuple([int(v) for v in ul])
  lr = tuple([int(v) for v in lr])
  if len(ul) > MAX_DIM:
    raise AssertionError('more than %d dimensions' % MAX_DIM)
  for u, l in zip(ul, lr):
    if u >= l:
      return None
  return TileExtent(ul, lr, array_shape)


def from_shape(shp):
  """extent.pyx:184-193."""
  return create([0] * len(shp), shp, shp)


def from_tuple(tup):
  return create(tup[0], tup[1], tup[2])


def unravelled_pos(idx, array_shape):
  """extent.pyx:195-209 (C division on the index)."""
  idx = int(idx)
  unravelled = []
  for dim in reversed(array_shape):
    dim = int(dim)
    unravelled.append(idx % dim)
    idx //= dim
  return tuple(reversed(unravelled))


def ravelled_pos(idx, array_shape):
  """extent.pyx:211-219."""
  rpos = 0
  mul = 1
  for i in range(len(array_shape) - 1, -1, -1):
    rpos += mul * int(idx[i])
    mul *= int(array_shape[i])
  return rpos


def all_nonzero_shape(shape):
  """extent.pyx:221-231."""
  for i in shape:
    if i == 0:
      return False
  return True


def find_rect(ravelled_ul, ravelled_lr, shape):
  """extent.pyx:233-252."""
  if shape[-1] == 1 or ravelled_ul // shape[-1] == ravelled_lr // shape[-1]:
    return (ravelled_ul, ravelled_lr)
  div = 1
  for i in shape[1:]:
    div = div * i
  rect_ul = ravelled_ul - (ravelled_ul % div)
  rect_lr = ravelled_lr + (div - ravelled_lr % div) % div - 1
  return (rect_ul, rect_lr)


def find_overlapping(extents, region):
  """extent.pyx:254-264."""
  for ex in extents:
    overlap = intersection(ex, region)
    if overlap is not None:
      yield (ex, overlap)


def compute_slice(base, idx):
  """extent.pyx:266-296: the extent of base[idx]."""
  if np.isscalar(idx):
    assert isinstance(idx, (int, np.integer))
    idx = slice(idx, idx + 1)
  if not isinstance(idx, tuple):
    idx = (idx,)
  ul, lr = [], []
  for i in range(base.ndim):
    if i >= len(idx):
      ul.append(base.ul[i])
      lr.append(base.lr[i])
    else:
      axis_idx = idx[i]
      if np.isscalar(axis_idx):
        axis_idx = slice(axis_idx, axis_idx + 1)
      start, stop, _ = axis_idx.indices(base.shape[i])
      ul.append(base.ul[i] + start)
      lr.append(base.ul[i] + stop)
  return create(ul, lr, base.array_shape)


def offset_from(base, other):
  """extent.pyx:298-314."""
  ul, lr = [], []
  for i in range(base.ndim):
    if other.ul[i] < base.ul[i] or other.lr[i] > base.lr[i]:
      raise AssertionError('%s is not inside %s' % (other, base))
    ul.append(other.ul[i] - base.ul[i])
    lr.append(other.lr[i] - base.ul[i])
  return create(ul, lr, other.array_shape)


def offset_slice(base, other):
  """extent.pyx:316-324."""
  return tuple(slice(other.ul[i] - base.ul[i], other.lr[i] - base.ul[i], None) for i in range(base.ndim))


def from_slice(idx, shape):
  """extent.pyx:326-361."""
  if not isinstance(idx, tuple):
    idx = (idx,)
  if len(idx) < len(shape):
    idx = tuple(list(idx) + [slice(None, None, None) for _ in range(len(shape) - len(idx))])
  ul, lr = [], []
  for i in range(len(shape)):
    dim = shape[i]
    slc = idx[i]
    if np.isscalar(slc):
      slc = int(slc)
      slc = slice(slc, slc + 1, None)
    # py2 `None > 0` is False: the asserts only fire for explicit positive bounds
    if slc.start is not None and slc.start > 0:
      assert slc.start <= dim
    if slc.stop is not None and slc.stop > 0:
      assert slc.stop <= dim
    indices = slc.indices(dim)
    ul.append(indices[0])
    lr.append(indices[1])
  return create(ul, lr, shape)


def intersection(a, b):
  """extent.pyx:367-387.  Touching boxes produce a degenerate box that `create`
  turns into None (the reference compares with `<`, not `<=`)."""
  if a is None:
    return None
  if a.array_shape != b.array_shape:
    Assert.eq(a.array_shape, b.array_shape, 'Tiles must have compatible shapes!')
  ul, lr = [], []
  for i in range(a.ndim):
    if b.lr[i] < a.ul[i]:
      return None
    if a.lr[i] < b.ul[i]:
      return None
    ul.append(a.ul[i] if a.ul[i] >= b.ul[i] else b.ul[i])
    lr.append(a.lr[i] if a.lr[i] < b.lr[i] else b.lr[i])
  return create(ul, lr, a.array_shape)


def shape_for_reduction(input_shape, axis):
  """extent.pyx:390-400 (returns a list for axis != None, like the reference)."""
  if axis is None:
    return ()
  input_shape = list(input_shape)
  del input_shape[axis]
  return input_shape


def shapes_match(offset, data):
  return np.all(offset.shape == data.shape)


def drop_axis(ex, axis):
  """extent.pyx:411-429."""
  if axis is None:
    return create((), (), ())
  if axis < 0:
    axis = ex.ndim + axis
  shape = list(ex.array_shape)
  del shape[axis]
  ul = list(ex.ul[:axis]) + list(ex.ul[axis + 1:])
  lr = list(ex.lr[:axis]) + list(ex.lr[axis + 1:])
  return create(ul, lr, shape)


def index_for_reduction(index, axis):
  """extent.pyx:431-432."""
  return drop_axis(index, axis)


def find_shape(extents):
  """extent.pyx:434-443."""
  corners = [ex.lr for ex in extents]
  return tuple([max(axis) or 1 for axis in zip(*corners)]) if corners and corners[0] else ()


def is_complete(shape, slices):
  """extent.pyx:446-464."""
  if len(shape) != len(slices):
    return False
  for dim, slc in zip(shape, slices):
    if slc.start is not None and slc.start > 0:
      return False
    if slc.stop is not None and slc.stop < dim:
      return False
  return True


def largest_dim_axis(shape, exclude_axes=None):
  """First axis of maximal length among those not excluded (0 when nothing is left or every length is 0)."""
  allowed = [i for i in range(len(shape)) if not exclude_axes or i not in exclude_axes]
  if not allowed:
    return 0
  best = max(allowed, key=lambda i: (shape[i], -i))
  return best if shape[best] > 0 else 0


def partition_axes(ex):
  """Axes along which the extent is a proper part of its array."""
  return [i for i in range(len(ex.shape)) if ex.shape[i] != ex.array_shape[i]]


def _to_grid(ex, n_dim):
  """A tile of a 1-D partition as the cell with the same ordinal number of an n x n (x ...) grid over the first
  n_dim axes, cells numbered in row-major order (extent.pyx:502-530; its float n-th root and Python-2 integer
  division kept)."""
  cut = partition_axes(ex)
  if len(cut) > 1:
    return ex
  step = ex.lr[cut[0]] - ex.ul[cut[0]]
  ordinal = int(ex.ul[cut[0]] // step)
  per_side = int(math.pow(divup(ex.array_shape[cut[0]], step), 1.0 / n_dim))
  cell = []
  for _ in range(n_dim):                       # base-`per_side` digits of the ordinal, last axis first
    cell.insert(0, ordinal % per_side)
    ordinal = (ordinal - cell[0]) // per_side
  sides = [divup(ex.array_shape[i], per_side) for i in range(n_dim)]
  ul = [side * c for side, c in zip(sides, cell)]
  lr = [side * (c + 1) for side, c in zip(sides, cell)]
  if any(hi > n for hi, n in zip(lr, ex.array_shape)):
    return None
  return create(ul, lr, ex.array_shape)


def change_partition_axis(ex, axis):
  """The tile with the same ordinal number when the array is cut along `axis` instead (the re-tiling step of the
  joins: worker i's row tile of A becomes A's i-th column slab).  Integer results are the reference's
  (extent.pyx:501-570).  `axis` may be a list of axes: a 1-D partition is mapped onto a grid over them."""
  if isinstance(axis, (list, tuple)):
    return _to_grid(ex, len(axis))
  shape = ex.array_shape
  if axis < 0:
    axis += len(shape)
  if len(ex.shape) == 1:                       # a vector has one way to be cut; "axis 1" means all of it
    return create((0,), shape, shape) if axis == 1 else ex
  cut = partition_axes(ex)
  if len(cut) > 1:
    # a cell of a 2-D grid becomes the slab with the cell's row-major number, one index thick
    number = (ex.ul[0] // ex.shape[0]) * divup(shape[1], ex.shape[1]) + ex.ul[1] // ex.shape[1]
    ul, lr = [0, 0], list(shape)
    ul[axis], lr[axis] = number, number + 1
    return create(ul, lr, shape)
  if not cut or cut[0] == axis:
    return ex
  old = cut[0]
  ul, lr = list(ex.ul), list(ex.lr)
  # the same FRACTION of the new axis, rounded up at both ends
  ul[axis] = divup(ex.ul[old] * shape[axis], shape[old])
  lr[axis] = divup(ex.lr[old] * shape[axis], shape[old])
  ul[old], lr[old] = 0, shape[old]
  return create(ul, lr, shape)
