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
  __slots__ = ('ul', 'lr', 'array_shape')

  def __init__(self, ul, lr, array_shape):
    self.ul = tuple(int(v) for v in ul)
    self.lr = tuple(int(v) for v in lr)
    self.array_shape = None if array_shape is None else tuple(int(v) for v in array_shape)

  @property
  def size(self):
    return int(np.prod(self.shape, dtype=np.int64)) if len(self.shape) else 1

  @property
  def shape(self):
    # extent.pyx:66-72 -- a zero-length dimension reports 1
    return tuple(1 if (l - u) == 0 else (l - u) for u, l in zip(self.ul, self.lr))

  @property
  def ndim(self):
    return len(self.ul)

  def to_slice(self):
    return tuple(slice(u, l) for u, l in zip(self.ul, self.lr))

  def to_tuple(self):
    return (self.ul, self.lr, self.array_shape)

  def __repr__(self):
    return 'extent(' + ','.join('%s:%s' % (a, b) for a, b in zip(self.ul, self.lr)) + ')'

  def __getitem__(self, idx):
    return create((self.ul[idx],), (self.lr[idx],), (self.array_shape[idx],))

  def __hash__(self):
    return hash(self.ul)  # extent.pyx:93-94

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

  def ravelled_pos(self):
    return ravelled_pos(self.ul, self.array_shape)

  def to_global(self, idx, axis):
    """extent.pyx:121-127."""
    if axis is not None:
      return idx + self.ul[axis]
    local_idx = unravelled_pos(idx, self.shape)
    return ravelled_pos(tuple(u + l for u, l in zip(self.ul, local_idx)), self.array_shape)

  def add_dim(self):
    return create(self.ul + (0,), self.lr + (1,), self.array_shape + (1,))

  def clone(self):
    return create(self.ul, self.lr, self.array_shape)


def create(ul, lr, array_shape):
  """extent.pyx:141-182 -- returns None for an unrealistic box (any ul >= lr)."""
  ul = tuple(int(v) for v in ul)
  lr = tuple(int(v) for v in lr)
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
  shape = np.max([ex.lr for ex in extents], axis=0)
  shape[shape == 0] = 1
  return tuple(int(v) for v in shape)


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
  """extent.pyx:466-476."""
  largest_dim = 0
  largest_axis = 0
  for i in range(len(shape)):
    if exclude_axes is not None and i in exclude_axes:
      continue
    if largest_dim < shape[i]:
      largest_dim = shape[i]
      largest_axis = i
  return largest_axis


def partition_axes(ex):
  """extent.pyx:493-499."""
  return [i for i in range(len(ex.shape)) if ex.shape[i] != ex.array_shape[i]]


def change_partition_axis(ex, axis):
  """extent.pyx:501-570: re-map a 1-D-partitioned extent onto another axis."""
  if isinstance(axis, (list, tuple)):
    old_axes = partition_axes(ex)
    if len(old_axes) > 1:
      return ex
    old_axis = old_axes[0]
    n_dim = len(axis)
    step = ex.lr[old_axis] - ex.ul[old_axis]
    ntiles = divup(ex.array_shape[old_axis], step)
    original_index = int(ex.ul[old_axis] // step)
    n = int(math.pow(ntiles, 1.0 / n_dim))
    grid_index = [0 for _ in range(n_dim)]
    for i in reversed(range(n_dim)):
      grid_index[i] = original_index % n
      original_index -= grid_index[i]
      original_index //= n
    steps = [divup(ex.array_shape[i], n) for i in range(n_dim)]
    ul = [steps[i] * grid_index[i] for i in range(n_dim)]
    lr = [steps[i] * (grid_index[i] + 1) for i in range(n_dim)]
    for i in range(len(lr)):
      if lr[i] > ex.array_shape[i]:
        return None
    return create(ul, lr, ex.array_shape)

  if axis < 0:
    axis += len(ex.array_shape)
  if len(ex.shape) == 1:  # vector special case, extent.pyx:537-542
    if axis == 1:
      return create((0,), ex.array_shape, ex.array_shape)
    return ex
  old_axes = partition_axes(ex)
  if len(old_axes) > 1:  # grid -> 1-D, extent.pyx:545-552
    blk_idx = (ex.ul[0] // ex.shape[0]) * divup(ex.array_shape[1], ex.shape[1]) + ex.ul[1] // ex.shape[1]
    ul = [0, 0]
    lr = list(ex.array_shape)
    ul[axis] = blk_idx
    lr[axis] = blk_idx + 1
    return create(ul, lr, ex.array_shape)
  if len(old_axes) == 0 or old_axes[0] == axis:
    return ex
  old_axis = old_axes[0]
  new_ul = list(ex.ul)
  new_lr = list(ex.lr)
  new_ul[axis] = divup(new_ul[old_axis] * ex.array_shape[axis], ex.array_shape[old_axis])
  new_ul[old_axis] = 0
  new_lr[axis] = divup(new_lr[old_axis] * ex.array_shape[axis], ex.array_shape[old_axis])
  new_lr[old_axis] = ex.array_shape[old_axis]
  return create(new_ul, new_lr, ex.array_shape)
