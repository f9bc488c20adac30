"""Tile: the unit of storage (an HBM blob) and of the combine step.

Mirror of the reference's spartan/array/tile.pyx for DENSE tiles.  `data` is a
backend tensor (torch tensor in HBM for the HIP backend) instead of a NumPy
array; the mask is kept as a *state* (all-clear / all-set) and only
materialised as a byte array in HBM when a sub-slice update makes it
non-uniform -- the reference allocates a 1 B/element bool array for every tile
(tile.pyx:145-159), which would cost 25 % extra HBM traffic on fp32 tiles.
"""
import itertools

import numpy as np

from ..util import Assert

TYPE_EMPTY = 0
TYPE_DENSE = 1
TYPE_MASKED = 2
TYPE_SPARSE = 3

MASK_ALL_CLEAR = 0
MASK_ALL_SET = 1

_ID = itertools.count()


class EmptyBlob(object):
  """What `Tile.get()` returns for a tile that has no data yet: the reference
  returns an uninitialised np.ndarray (tile.pyx:72-79) that creation mappers
  only inspect for .shape/.dtype; no HBM is touched here."""
  __slots__ = ('shape', 'dtype')

  def __init__(self, shape, dtype):
    self.shape = tuple(shape)
    self.dtype = np.dtype(dtype)


class Tile(object):
  """tile.pyx:24-62."""

  def __init__(self, shape, dtype, data, mask, tile_type):
    self.id = next(_ID)
    self.shape = tuple(int(s) for s in shape)
    self.dtype = np.dtype(dtype)
    self.type = tile_type
    self.mask = mask      # MASK_ALL_CLEAR | MASK_ALL_SET | backend byte tensor
    self.data = data      # backend tensor or None
    self.refcnt = 1

  def __repr__(self):
    return 'tile(%s, %s) [%s]' % (self.shape, self.dtype, 'empty' if self.data is None else 'data')

  def mask_is_uniform(self):
    return isinstance(self.mask, int)

  def get(self, backend, subslice=None):
    """tile.pyx:64-113.  Returns a backend tensor (a view when possible)."""
    if subslice is not None and not isinstance(subslice, tuple):
      subslice = (subslice,)
    if self.data is None:
      if subslice is None or len(self.shape) == 0:
        return EmptyBlob(self.shape, self.dtype)
      shp = []
      for slc, n in zip(subslice, self.shape):
        start, stop, _ = slc.indices(n)
        shp.append(max(stop - start, 0))
      shp += list(self.shape[len(subslice):])
      return EmptyBlob(shp, self.dtype)
    if len(self.shape) == 0:
      return self.data
    if subslice is None:
      subslice = tuple(slice(None) for _ in self.shape)
    Assert.le(len(subslice), len(self.shape), 'Selector has more dimensions than data!')
    if not self.mask_is_uniform():
      if not backend.mask_all_set(self.mask, subslice):
        # the reference returns a numpy.ma.MaskedArray here (tile.pyx:104-112);
        # masked arrays do not exist on the device
        raise NotImplementedError('reading a region of a tile with unset cells '
                                  '(reference: MaskedArray) is not supported on the GPU backend')
    elif self.mask == MASK_ALL_CLEAR:
      raise NotImplementedError('reading an initialised-but-unwritten tile (reference: MaskedArray)')
    return self.data[subslice]

  def update(self, backend, subslice, data, reducer, owned=False):
    return merge(backend, self, subslice, data, reducer, owned)


def from_data(data, dtype=None, shape=None):
  """tile.pyx:145-159 (mask = all set, kept as a state)."""
  return Tile(shape=tuple(data.shape) if shape is None else shape,
              data=data, dtype=dtype, mask=MASK_ALL_SET, tile_type=TYPE_DENSE)


def from_shape(shape, dtype, tile_type=TYPE_DENSE):
  """tile.pyx:162-176: an empty tile carries no data."""
  if tile_type != TYPE_DENSE:
    raise NotImplementedError('sparse tiles are outside the GPU tile path (SURVEY 8f.2)')
  return Tile(shape=shape, data=None, dtype=dtype, tile_type=tile_type, mask=MASK_ALL_CLEAR)


def merge(backend, old_tile, subslice, update, reducer, owned=False):
  """tile.pyx:200-297, dense->dense branch, executed by backend.update_box
  (sp_update on the GPU).  `update` is a backend tensor; `owned` says the
  caller hands the (contiguous, freshly produced) tensor over, so a first
  full-tile write can adopt it instead of copying."""
  Assert.isinstance(old_tile, Tile)
  nd = len(old_tile.shape)

  if nd == 0:
    # tile.pyx:212-217: data None acts as the mask
    if old_tile.data is None or reducer is None:
      data = backend.astype(update.reshape(()), old_tile.dtype)
      if not owned and data.data_ptr() == update.data_ptr():
        data = backend.copy(data)     # never alias the caller's tensor
      old_tile.data = data
    else:
      backend.update_box(old_tile.data, (), (), update.reshape(()), reducer, MASK_ALL_SET, None)
    old_tile.mask = MASK_ALL_SET
    return old_tile

  ushape = tuple(update.shape)
  full = ushape == old_tile.shape
  if subslice is None:
    subslice = tuple(slice(0, n) for n in old_tile.shape)
  ul, lr = [], []
  for slc, n in zip(subslice, old_tile.shape):
    start, stop, _ = slc.indices(n)
    ul.append(start)
    lr.append(stop)
  for d in range(len(subslice), nd):
    ul.append(0)
    lr.append(old_tile.shape[d])

  if full:
    # tile.pyx:261-268: whole-tile fast path keyed on mask[0...]
    if old_tile.data is None:
      first_set = False
    elif old_tile.mask_is_uniform():
      first_set = old_tile.mask == MASK_ALL_SET
    else:
      first_set = backend.mask_first(old_tile.mask)
    if reducer is not None and first_set:
      backend.update_box(old_tile.data, [0] * nd, old_tile.shape, update, reducer, MASK_ALL_SET, None)
    else:
      if old_tile.data is None and backend.same_dtype(update, old_tile.dtype):
        old_tile.data = update if owned else backend.copy(update)
      else:
        if old_tile.data is None:
          old_tile.data = backend.empty(old_tile.shape, old_tile.dtype)
        backend.update_box(old_tile.data, [0] * nd, old_tile.shape, update, None, MASK_ALL_CLEAR, None)
    old_tile.mask = MASK_ALL_SET
    return old_tile

  # sub-slice update (tile.pyx:270-283)
  if old_tile.data is None:
    # _initialize(): zeros + an all-clear mask (tile.pyx:115-127)
    old_tile.data = backend.zeros(old_tile.shape, old_tile.dtype)
    old_tile.mask = MASK_ALL_CLEAR
  if old_tile.mask_is_uniform():
    if old_tile.mask == MASK_ALL_SET:
      backend.update_box(old_tile.data, ul, lr, update, reducer, MASK_ALL_SET, None)
      return old_tile
    # all clear -> becomes non-uniform: materialise the byte mask
    mask = backend.zeros(old_tile.shape, np.uint8)
    backend.update_box(old_tile.data, ul, lr, update, reducer, MASK_ALL_CLEAR, mask)
    old_tile.mask = mask
    return old_tile
  backend.update_box(old_tile.data, ul, lr, update, reducer, 2, old_tile.mask)
  return old_tile
