"""Tile: the unit of storage (an HBM blob) and of the combine step.

Mirror of the reference's spartan/array/tile.pyx.  `data` is a backend tensor
(a device array in HBM for the HIP backend, spartan_amd/devarray.py) instead of a NumPy array; the mask is
kept as a *state* (all-clear / all-set) and only materialised as a byte array in
HBM when a sub-slice update makes it non-uniform -- the reference allocates a
1 B/element bool array for every tile (tile.pyx:145-159), which would cost 25 %
extra HBM traffic on fp32 tiles.

Sparse tiles (TYPE_SPARSE) hold a backend sparse blob: canonical CSR in HBM for
the HIP backend (spartan_amd/sparse.py) where the reference holds a scipy.sparse
matrix; they have no mask (tile.pyx:129-132).
"""
import itertools

import numpy as np

from ..util import Assert

TYPE_EMPTY, TYPE_DENSE, TYPE_MASKED, TYPE_SPARSE = range(4)       # tile.pyx:11-14

MASK_ALL_CLEAR, MASK_ALL_SET = 0, 1                              # uniform mask states (see the module docstring)

_ID = itertools.count()


class EmptyBlob(object):
  """What `Tile.get()` returns for a tile that has no data yet: the reference
  returns an uninitialised np.ndarray (tile.pyx:72-79) that creation mappers
  only inspect for .shape/.dtype; no HBM is touched here."""
  __slots__ = ('shape', 'dtype')

  def __init__(self, shape, dtype):
    self.shape = tuple(shape)
    self.dtype = np.dtype(dtype)


class MaskedBlob(object):
  """What `Tile.get()` returns for a region in which some cells were never written: the values together with
  the written-cells mask, both in HBM -- the device form of the numpy.ma.MaskedArray the reference builds there
  (tile.pyx:100-113: `masked_all` + the written cells copied in).  `data` and `valid` are backend tensors of the
  same shape (`valid` uint8, 1 = written); `to_host` gives the reference's MaskedArray.  Kernels do not take
  masked operands: a program that computes on a partially written array is refused (MaskedOperandError)."""
  __slots__ = ('data', 'valid')

  def __init__(self, data, valid):
    self.data = data
    self.valid = valid

  @property
  def shape(self):
    return tuple(self.data.shape)

  @property
  def dtype(self):
    return self.data.dtype

  def to_host(self, backend):
    data = backend.to_numpy(self.data)
    valid = backend.to_numpy(self.valid).astype(bool)
    out = np.ma.masked_all(data.shape, dtype=data.dtype)
    out[valid] = data[valid]
    return out


class MaskedOperandError(TypeError):
  """A kernel was asked to compute on a region with never-written cells."""


def reject_masked(values, what):
  """Kernels do not take operands with never-written cells (MaskedBlob -- the reference would hand a
  numpy.ma.MaskedArray to the NumPy function): one explicit error instead of whatever the first attribute access
  on the blob would raise."""
  for v in (values.values() if isinstance(values, dict) else values):
    if isinstance(v, MaskedBlob):
      raise MaskedOperandError('%s of an array with never-written cells (a masked region, shape %s): write the '
                               'whole region first, or read it with glom() / fetch()' % (what, tuple(v.shape)))


def is_sparse_blob(x):
  """A backend sparse blob (device CSR) or a scipy.sparse matrix (what user mappers yield)."""
  if getattr(x, 'is_sparse_tile', False):
    return True
  return type(x).__module__.startswith('scipy.sparse')


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
    if self.type == TYPE_SPARSE:
      # tile.pyx:74-77, :91-99 (the reference warns that slicing a sparse tile "will likely fail";
      # here a box of a CSR tile is a well-defined device operation)
      if subslice is None:
        shp = self.shape
      else:
        shp = tuple(len(range(*slc.indices(n))) for slc, n in zip(subslice, self.shape)) + self.shape[len(subslice):]
      if self.data is None:
        # tile.pyx:77 `coo_matrix(shape, self.dtype)`: the dtype lands in coo_matrix's `shape` parameter, so
        # the blob of a never-written sparse tile is float64 whatever the array's dtype -- and so is
        # everything a creation mapper derives from it (sparse_diagonal is float64 in the reference's
        # outputs, tests/golden/sparse_meta.json).  Kept.
        return backend.sparse_empty(shp, np.float64)
      if tuple(shp) == self.shape:
        return self.data
      return backend.sparse_slice(self.data, subslice)
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
        return MaskedBlob(self.data[subslice], self.mask[subslice])     # tile.pyx:104-112
    elif self.mask == MASK_ALL_CLEAR:
      # initialised (zeros) but nothing written yet: everything masked
      return MaskedBlob(self.data[subslice], backend.zeros(tuple(self.data[subslice].shape), np.uint8))
    return self.data[subslice]

  def update(self, backend, subslice, data, reducer, owned=False):
    return merge(backend, self, subslice, data, reducer, owned)


def from_data(data, dtype=None, shape=None):
  """tile.pyx:145-159 (mask = all set, kept as a state)."""
  if is_sparse_blob(data):
    return Tile(shape=tuple(data.shape), data=data, dtype=data.dtype if dtype is None else dtype,
                mask=MASK_ALL_SET, tile_type=TYPE_SPARSE)
  return Tile(shape=tuple(data.shape) if shape is None else shape,
              data=data, dtype=dtype, mask=MASK_ALL_SET, tile_type=TYPE_DENSE)


def from_shape(shape, dtype, tile_type=TYPE_DENSE):
  """tile.pyx:162-176: an empty tile carries no data."""
  if tile_type == TYPE_SPARSE:
    return Tile(shape=shape, data=None, dtype=dtype, tile_type=TYPE_SPARSE, mask=None)
  assert tile_type == TYPE_DENSE, 'Unknown tile type %s' % tile_type
  return Tile(shape=shape, data=None, dtype=dtype, tile_type=tile_type, mask=MASK_ALL_CLEAR)


def merge(backend, old_tile, subslice, update, reducer, owned=False):
  """tile.pyx:200-297, dense->dense branch, executed by backend.update_box
  (sp_update on the GPU).  `update` is a backend tensor; `owned` says the
  caller hands the (contiguous, freshly produced) tensor over, so a first
  full-tile write can adopt it instead of copying."""
  Assert.isinstance(old_tile, Tile)
  nd = len(old_tile.shape)

  if is_sparse_blob(update) or old_tile.type == TYPE_SPARSE:
    return _merge_sparse(backend, old_tile, subslice, update, reducer)

  if nd == 0:
    # tile.pyx:212-217: data None acts as the mask
    if old_tile.data is None or reducer is None:
      data = backend.astype(update.reshape(()), old_tile.dtype)
      if not owned and backend.same_memory(data, update):
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


def _box_of(subslice, shape):
  if subslice is None:
    return tuple(0 for _ in shape), tuple(shape)
  ul, lr = [], []
  for slc, n in zip(subslice, shape):
    start, stop, _ = slc.indices(n)
    ul.append(start)
    lr.append(stop)
  for d in range(len(subslice), len(shape)):
    ul.append(0)
    lr.append(shape[d])
  return tuple(ul), tuple(lr)


def _merge_sparse(backend, old_tile, subslice, update, reducer):
  """tile.pyx:226-252 (sparse update) and :283-295 (dense update of a sparse tile)."""
  Assert.eq(len(old_tile.shape), 2, 'sparse tiles are two-dimensional')
  ul, lr = _box_of(subslice, old_tile.shape)
  if is_sparse_blob(update):
    update = backend.sparse_blob(update, old_tile.dtype)
    if old_tile.type == TYPE_DENSE:
      # tile.pyx:229-243: sparse_to_dense_update(..., REDUCE_ADD) -- first write where the mask is clear,
      # add where it is set -- then mask[subslice] = True.  (The reference passes the update's own
      # coordinates without adding the box origin; the box origin is honoured here.)
      if old_tile.data is None:
        old_tile.data = backend.zeros(old_tile.shape, old_tile.dtype)
        old_tile.mask = MASK_ALL_CLEAR
      full = (ul == (0, 0) and lr == old_tile.shape)
      if old_tile.mask_is_uniform():
        if old_tile.mask == MASK_ALL_SET:
          backend.sparse_scatter(old_tile.data, ul, update, 1, None)
          return old_tile
        if full:
          backend.sparse_scatter(old_tile.data, ul, update, 0, None)
          old_tile.mask = MASK_ALL_SET
          return old_tile
        old_tile.mask = backend.zeros(old_tile.shape, np.uint8)
      backend.sparse_scatter(old_tile.data, ul, update, 2, old_tile.mask)
      backend.assign_box(old_tile.mask, tuple(slice(u, l) for u, l in zip(ul, lr)), 1)
      return old_tile
    # sparse update of a sparse tile (tile.pyx:244-252)
    if tuple(update.shape) == old_tile.shape:
      if reducer is not None and old_tile.data is not None:
        old_tile.data = backend.sparse_reduce(old_tile.data, update, reducer)
      else:
        old_tile.data = update
      return old_tile
    if old_tile.data is None:
      old_tile.data = backend.sparse_empty(old_tile.shape, old_tile.dtype)
    old_tile.data = backend.sparse_update(old_tile.data, ul, lr, update, reducer)
    return old_tile
  # dense update of a sparse tile: the reference converts the tile to lil and assigns the slice
  # (tile.pyx:283-295, marked "this is SLOW").  Here the block's non-zero cells merge as a sparse block: the same
  # result for reducer None (the box is replaced, zeros of the block included) and for np.add.
  return _merge_sparse(backend, old_tile, subslice, backend.dense_to_sparse(update, old_tile.dtype), reducer)
