"""Stretched views of distributed arrays (NumPy broadcasting without copies).

Role of the reference's spartan/expr/operator/broadcast.py (`Broadcast`, `broadcast`).  Nothing is ever
replicated in HBM: a `Broadcast` only translates regions between the stretched index space and the array
underneath, and the kernels read the un-stretched slab through zero strides (lower.py builds the strides from
the slab's size-1 axes).  The whole translation is one table, built once per view:

    axis_of[i]    the view axis that base axis i is aligned with (right alignment: i + lead)
    stretched[i]  base axis i has length 1 but the view axis is longer

* view region -> base region (`project`): a stretched axis always reads [0, 1), any other axis keeps the
  region's bounds; the `lead` leading view axes have no counterpart and drop out.
* base tile -> view region (`lift`): an aligned, un-stretched axis keeps the tile's bounds, every other view
  axis is covered completely -- the region of the result that the tile contributes to.
"""
import numpy as np

from ..array import distarray, extent


class Broadcast(distarray.DistArray):
  """`base` seen with shape `shape` (which `base.shape` broadcasts to)."""

  def __init__(self, base, shape):
    if not isinstance(base, (np.ndarray, distarray.DistArray)):
      raise TypeError('Broadcast of %r' % type(base))
    if isinstance(base, Broadcast):          # stretch of a stretch: translate from the innermost array
      base = base.base
    shape = tuple(int(s) for s in shape)
    lead = len(shape) - len(base.shape)
    if lead < 0:
      raise ValueError('cannot broadcast %s to %s' % (base.shape, shape))
    self.base = base
    self.shape = shape
    self.dtype = base.dtype
    self.sparse = base.sparse
    self.tiles = base.tiles
    self.bad_tiles = []
    self.axis_of = tuple(i + lead for i in range(len(base.shape)))
    self.stretched = tuple(n == 1 and shape[a] != 1 for n, a in zip(base.shape, self.axis_of))
    for n, a, st in zip(base.shape, self.axis_of, self.stretched):
      if not st and n != shape[a]:
        raise ValueError('cannot broadcast %s to %s' % (base.shape, shape))

  def __repr__(self):
    return 'Broadcast(%s -> %s)' % (self.base, self.shape)

  # the array with the most elements drives a map; a view must lose a tie against a real array of its size
  def real_size(self):
    return self.base.real_size() - 1

  def extent_for_blob(self, tile_id):
    return self.base.extent_for_blob(tile_id)

  # -- region translation ----------------------------------------------------------------------------
  def project(self, region):
    """The part of `base` that the view region `region` reads."""
    bounds = [(0, 1) if st else (region.ul[a], region.lr[a]) for a, st in zip(self.axis_of, self.stretched)]
    return extent.create([b[0] for b in bounds], [b[1] for b in bounds], self.base.shape)

  def lift(self, base_region):
    """The view region that the tile `base_region` of `base` spans."""
    ul = [0] * len(self.shape)
    lr = list(self.shape)
    for i, (a, st) in enumerate(zip(self.axis_of, self.stretched)):
      if not st:
        ul[a], lr[a] = base_region.ul[i], base_region.lr[i]
    return extent.create(ul, lr, self.shape)

  # -- DistArray interface ---------------------------------------------------------------------------
  def fetch_base_tile(self, region):
    """The un-stretched slab under `region` (size-1 axes intact, leading axes absent)."""
    return self.base.fetch(self.project(region))

  fetch = fetch_base_tile

  def foreach_tile(self, mapper_fn, kw=None):
    return distarray.run_kernel(self, list(self.tiles.values()), mapper_fn, kw or {})

  def _invoke_mapper(self, tile_id, blob, mapper_fn, kw):
    return mapper_fn(self.lift(self.base.extent_for_blob(tile_id)), **kw)


def common_shape(shapes):
  """NumPy's broadcasting rule on a list of shapes (right-aligned; per axis all lengths equal or 1)."""
  nd = max(len(s) for s in shapes)
  out = [1] * nd
  for s in shapes:
    at = nd - len(s)
    for n in s:
      n = int(n)
      have = out[at]
      if n != have:
        if have == 1:
          out[at] = n
        elif n != 1:
          raise AssertionError('Mismatched shapes for broadcast: %s' % [list(x) for x in shapes])
      at += 1
  return tuple(out)


def broadcast(args):
  """Bring the arrays to one shape: arrays that already have it are passed through, the others are wrapped."""
  if len(args) < 2:
    return args
  target = common_shape([tuple(a.shape) for a in args])
  return [a if tuple(a.shape) == target else Broadcast(a, target) for a in args]
