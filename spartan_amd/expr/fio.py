"""File I/O in the reference's on-disk layout (spartan/expr/fio.py): `<path>/<prefix>/<prefix>_dist.spf`
describes the array (shape / tile shape / dtype / DENSITY), every tile is one file
`<prefix>_<ul>_<lr>_spf[bz2]` = an npy-style header (magic, 2-byte little-endian length, a dict with
ul/lr/shape/dtype/type padded to 16 bytes) followed by the raw C-order tile bytes; `pickle` writes
`..._spp[bz2]` files holding the pickled tile.  Each tile is written / read by the worker that owns it
(HBM <-> host copy of that one tile).  A sparse tile is the same header with type SPARSE plus an
`.npz` holding row / col / data / shape (fio.py:98-104), or the pickled scipy matrix."""
import ast
import bz2
import os
import pickle as _pickle

import numpy as np

from .base import Expr
from .ndarray import ndarray
from .shuffle import shuffle
from .. import context
from ..array import distarray, tile as tile_mod
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
  if tile_mod.is_sparse_blob(data):
    return ctx.backend.sparse_to_host(data)
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
        sparse = tile_mod.is_sparse_blob(tile)
        tile_dict = {'ul': ex.ul, 'lr': ex.lr, 'shape': tile.shape, 'dtype': str(tile.dtype),
                     'type': "SPARSE" if sparse else "DENSITY"}
        dict_cnt = str(tile_dict)
        if (len(_MAGIC) + 2 + len(dict_cnt)) % 16 != 0:
          dict_cnt += (16 - (len(_MAGIC) + 2 + len(dict_cnt)) % 16) * ' '
        fp.write(_MAGIC + bytes([len(dict_cnt) % 256, len(dict_cnt) // 256]) + dict_cnt.encode('latin-1'))
        if sparse:
          coo = tile.tocoo()
          kw['isnp'] = True
          (np.savez_compressed if iszip else np.savez)(save_filename(**kw), row=coo.row, col=coo.col, data=coo.data,
                                                       shape=coo.shape)
        else:
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
  return {'shape': shape, 'sparse': sparse, 'dtype': dtype, 'tile_hint': tile_hint}


def _read_tile(ex, path, prefix, dtype, iszip, ispickle, sparse=False):
  kw = {'path': path, 'prefix': prefix, 'suffix': '', 'ul': ex.ul, 'lr': ex.lr, 'ispickle': ispickle,
        'isnp': False, 'iszip': bool(iszip)}
  if sparse and not ispickle:
    import scipy.sparse
    kw['isnp'] = True
    a = np.load(save_filename(**kw) + '.npz')                  # fio.py:181-184
    return scipy.sparse.coo_matrix((a['data'], (a['row'], a['col'])), tuple(a['shape']))
  with _open(save_filename(**kw), 'r', iszip) as fp:
    if ispickle:
      obj = _pickle.load(fp)
      return obj if tile_mod.is_sparse_blob(obj) else np.asarray(obj)
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
  data = _read_tile(ex, path, prefix, dtype, iszip, ispickle, sparse=bool(sparse))
  if tile_mod.is_sparse_blob(data):
    return [(ex, ctx.backend.sparse_blob(data, dtype))]
  return [(ex, ctx.backend.from_numpy(data))]


def load(prefix, path='.', iszip=False):
  """Lazy: a new array with the tiles stored under `prefix` (fio.py:214-232)."""
  info = _load(path, prefix, iszip)
  return shuffle(ndarray(info['shape'], dtype=info['dtype'], tile_hint=info['tile_hint'], sparse=info['sparse']),
                 fn=_load_mapper,
                 kw={'path': path, 'prefix': prefix, 'sparse': info['sparse'], 'dtype': info['dtype'],
                     'iszip': iszip},
                 shape_hint=info['shape'])


def unpickle(prefix, path='.', iszip=False):
  """fio.py:302-320."""
  info = _load(path, prefix, iszip)
  return shuffle(ndarray(info['shape'], dtype=info['dtype'], tile_hint=info['tile_hint'], sparse=info['sparse']),
                 fn=_load_mapper,
                 kw={'path': path, 'prefix': prefix, 'sparse': info['sparse'], 'dtype': info['dtype'],
                     'iszip': iszip, 'ispickle': True},
                 shape_hint=info['shape'])


def _partial_load(path, prefix, extents, iszip, ispickle):
  """fio.py:353-382: load the tiles named by `extents` ({extent: worker}) onto those workers; returns
  {extent: tile_id}.  Not lazy.  (The reference uses it to re-load the tiles of a failed worker from a
  checkpoint, checkpoint.py:27-37.)"""
  ctx = context.get()
  info = _load(path, prefix, iszip)
  loaded = {}
  for ex, worker in extents.items():
    worker = int(worker) % ctx.num_workers
    with ctx.on_worker(worker):
      t = None
      if ctx.executing:
        data = _read_tile(ex, path, prefix, info['dtype'], iszip, ispickle, sparse=info['sparse'])
        data = ctx.backend.sparse_blob(data, info['dtype']) if tile_mod.is_sparse_blob(data) else ctx.backend.from_numpy(data)
        t = tile_mod.from_data(data, dtype=info['dtype'])
      loaded[ex] = ctx.create(t, hint=worker)
  return loaded


def partial_load(extents, prefix, path=".", iszip=False):
  """fio.py:385-398."""
  return _partial_load(path, prefix, extents, iszip, False)


def partial_unpickle(extents, prefix, path=".", iszip=False):
  """fio.py:401-414."""
  return _partial_load(path, prefix, extents, iszip, True)


def from_file(fn, file_type='numpy', sparse=True, tile_hint=None):
  """Make an array from a file read on the driver (write_array.py:380-421): `numpy` (dense .npy / one-array
  .npz, or with sparse=True the four files <fn>_shape/_row/_col/_data.npy of a COO matrix) or `mm` (Matrix
  Market, dense or sparse; a float64 sparse matrix is narrowed to float32 like the reference does)."""
  from .builtins import from_numpy
  import scipy.sparse
  if file_type == 'numpy':
    if sparse:
      shape = [int(v) for v in np.load(fn + '_shape.npy')]
      npa = scipy.sparse.coo_matrix((np.load(fn + '_data.npy'), (np.load(fn + '_row.npy'), np.load(fn + '_col.npy'))),
                                    shape=shape)
    else:
      npa = np.load(fn)
      if fn.endswith("npz"):
        data = None
        for _k, v in npa.items():      # "we expect only one npy in npz" (write_array.py:409-413)
          data = v
        npa.close()
        npa = data
  elif file_type == 'mm':
    import scipy.io
    npa = scipy.io.mmread(fn)
    if scipy.sparse.issparse(npa) and npa.dtype == np.float64:
      npa = npa.astype(np.float32)
  else:
    raise NotImplementedError("Only support npy and mm now. Got %s" % file_type)
  if scipy.sparse.issparse(npa):
    return from_numpy(npa, tile_hint)
  return from_numpy(np.asarray(npa), tile_hint)


def from_file_parallel(fn, file_format='mm', sparse=True, tile_hint=None):
  """write_array.py:315-377: every worker reads its own part of the file.  Every rank of this SPMD program already
  reads the file itself in `from_file` and keeps only the tiles it owns (nothing passes through a driver), so the two
  entry points share one implementation; `file_format` is 'mm' or 'numpy'."""
  return from_file(fn, file_type=file_format, sparse=sparse, tile_hint=tile_hint)
