"""File I/O in the reference's on-disk layout (spartan/expr/fio.py): `<path>/<prefix>/<prefix>_dist.spf`
describes the array (shape / tile shape / dtype / DENSITY), every tile is one file
`<prefix>_<ul>_<lr>_spf[bz2]` = an npy-style header (magic, 2-byte little-endian length, a dict with
ul/lr/shape/dtype/type padded to 16 bytes) followed by the raw C-order tile bytes; `pickle` writes
`..._spp[bz2]` files holding the pickled tile.  Each tile is written / read by the worker that owns it
(HBM <-> host copy of that one tile); sparse tiles are outside the GPU tile path."""
import ast
import bz2
import os
import pickle as _pickle

import numpy as np

from .base import Expr
from .ndarray import ndarray
from .shuffle import shuffle
from .. import context
from ..array import distarray
from ..context import LocalKernelResult

_MAGIC = b"\x93NUMPY\x01\x00"


def save_filename(**kw):
  """fio.py:49-67."""
  fn = kw['path'] + "/" + kw['prefix'] + "/" + kw['prefix'] + "_" + str(kw['ul']) + "_" + str(kw['lr'])
  if kw['suffix'] != "":
    fn += "_" + kw['suffix']
  if not kw['isnp']:
    fn += "_" + "sp"
    fn += "p" if kw['ispickle'] else "f"
    if kw['iszip']:
      fn += "bz2"
  return fn


def _open(fn, mode, iszip):
  return bz2.BZ2File(fn, mode, compresslevel=1) if (iszip and 'w' in mode) else (
      bz2.BZ2File(fn, mode) if iszip else open(fn, mode + 'b'))


def _tile_to_host(array, ex):
  """The tile's data as a C-contiguous NumPy array on the rank that owns it (None elsewhere)."""
  ctx = context.get()
  data = array.fetch(ex)
  if isinstance(data, distarray.Absent) or not ctx.executing:
    return None
  return np.ascontiguousarray(ctx.backend.to_numpy(data))


def _save_tile_mapper(ex, src=None, path=None, prefix=None, iszip=None, ispickle=None):
  """fio.py:70-112 / :235-257, one tile."""
  tile = _tile_to_host(src, ex)
  if tile is not None:
    os.makedirs(path + '/' + prefix, exist_ok=True)
    kw = {'path': path, 'prefix': prefix, 'suffix': '', 'ul': ex.ul, 'lr': ex.lr, 'ispickle': ispickle,
          'isnp': False, 'iszip': bool(iszip)}
    with _open(save_filename(**kw), 'w', iszip) as fp:
      if ispickle:
        _pickle.dump(tile, fp, -1)
      else:
        tile_dict = {'ul': ex.ul, 'lr': ex.lr, 'shape': tile.shape, 'dtype': str(tile.dtype), 'type': "DENSITY"}
        dict_cnt = str(tile_dict)
        if (len(_MAGIC) + 2 + len(dict_cnt)) % 16 != 0:
          dict_cnt += (16 - (len(_MAGIC) + 2 + len(dict_cnt)) % 16) * ' '
        fp.write(_MAGIC + bytes([len(dict_cnt) % 256, len(dict_cnt) // 256]) + dict_cnt.encode('latin-1'))
        fp.write(tile.tobytes())
  return LocalKernelResult(result=[])


def _save(path, prefix, array, iszip):
  """fio.py:115-131: the array-level description file (every rank writes the same text)."""
  path = path + '/' + prefix
  os.makedirs(path, exist_ok=True)
  with open(path + '/' + prefix + "_dist.spf", "w") as fp:
    fp.write("".join(str(dim) + " " for dim in array.shape) + "\n")
    fp.write("".join(str(dim) + " " for dim in array.tile_shape()) + "\n")
    fp.write(str(array.dtype) + "\n")
    fp.write("SPARSE\n" if array.sparse else "DENSITY\n")


def _dump(array, prefix, path, iszip, ispickle):
  if isinstance(array, Expr):
    array = array.evaluate()
  _save(path, prefix, array, iszip)
  array.foreach_tile(mapper_fn=_save_tile_mapper,
                     kw={'src': array, 'path': path, 'prefix': prefix, 'iszip': iszip, 'ispickle': ispickle})
  context.get().world.barrier()
  return True


def save(array, prefix, path='.', iszip=False):
  """Not lazy; True on success (fio.py:134-156)."""
  return _dump(array, prefix, path, iszip, False)


def pickle(array, prefix, path='.', iszip=False):
  """fio.py:260-282."""
  return _dump(array, prefix, path, iszip, True)


def _load(path, prefix, iszip):
  """fio.py:194-211."""
  fn = path + "/" + prefix + "/" + prefix + "_dist.spf"
  if not os.path.exists(fn):
    raise IOError(fn)
  with open(fn) as fp:
    shape = [int(i) for i in fp.readline().strip().split()]
    tile_hint = [int(i) for i in fp.readline().strip().split()]
    dtype = np.dtype("".join(fp.readline().strip()))
    sparse = fp.readline().find("SPARSE") != -1
  if sparse:
    raise NotImplementedError('sparse arrays are outside the GPU tile path (SURVEY 8f.2)')
  return {'shape': shape, 'sparse': sparse, 'dtype': dtype, 'tile_hint': tile_hint}


def _read_tile(ex, path, prefix, dtype, iszip, ispickle):
  kw = {'path': path, 'prefix': prefix, 'suffix': '', 'ul': ex.ul, 'lr': ex.lr, 'ispickle': ispickle,
        'isnp': False, 'iszip': bool(iszip)}
  with _open(save_filename(**kw), 'r', iszip) as fp:
    if ispickle:
      return np.asarray(_pickle.load(fp))
    fp.read(8)                                   # magic number and version
    dlen = fp.read(2)
    ast.literal_eval(fp.read(dlen[0] + dlen[1] * 256).decode('latin-1'))   # (redundant, as in the reference)
    data = np.frombuffer(fp.read(), dtype=dtype).copy()
  data.shape = ex.shape
  return data


def _load_mapper(array, ex, prefix=None, path=None, sparse=None, dtype=None, iszip=None, ispickle=False):
  """fio.py:159-191 / :285-299."""
  ctx = context.get()
  if not ctx.executing:
    return [(ex, distarray.Absent(ex.shape, dtype))]
  return [(ex, ctx.backend.from_numpy(_read_tile(ex, path, prefix, dtype, iszip, ispickle)))]


def load(prefix, path='.', iszip=False):
  """Lazy: a new array with the tiles stored under `prefix` (fio.py:214-232)."""
  info = _load(path, prefix, iszip)
  return shuffle(ndarray(info['shape'], dtype=info['dtype'], tile_hint=info['tile_hint']),
                 fn=_load_mapper,
                 kw={'path': path, 'prefix': prefix, 'sparse': info['sparse'], 'dtype': info['dtype'],
                     'iszip': iszip},
                 shape_hint=info['shape'])


def unpickle(prefix, path='.', iszip=False):
  """fio.py:302-320."""
  info = _load(path, prefix, iszip)
  return shuffle(ndarray(info['shape'], dtype=info['dtype'], tile_hint=info['tile_hint']),
                 fn=_load_mapper,
                 kw={'path': path, 'prefix': prefix, 'sparse': info['sparse'], 'dtype': info['dtype'],
                     'iszip': iszip, 'ispickle': True},
                 shape_hint=info['shape'])


def from_file(fn, file_type='numpy', sparse=True, tile_hint=None):
  """Make an array from a file read on the driver (write_array.py:380-421): `numpy` (.npy / one-array
  .npz) or `mm` (Matrix Market).  Sparse inputs would become sparse tiles in the reference; those are
  outside the GPU tile path, so `sparse=True` (the reference's default) is refused loudly
  unless the Matrix Market file holds a dense array."""
  from .builtins import from_numpy
  if file_type == 'numpy':
    if sparse:
      raise NotImplementedError('from_file(sparse=True): sparse tiles are outside the GPU tile path; '
                                'pass sparse=False for a dense .npy / .npz file')
    npa = np.load(fn)
    if fn.endswith("npz"):
      data = None
      for _k, v in npa.items():      # "we expect only one npy in npz" (write_array.py:409-413)
        data = v
      npa.close()
      npa = data
  elif file_type == 'mm':
    import scipy.io
    import scipy.sparse
    npa = scipy.io.mmread(fn)
    if scipy.sparse.issparse(npa):
      raise NotImplementedError('from_file: %s holds a sparse matrix; sparse tiles are outside the GPU tile path' % fn)
  else:
    raise NotImplementedError("Only support npy and mm now. Got %s" % file_type)
  return from_numpy(np.asarray(npa), tile_hint)
