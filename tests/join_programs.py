"""User-function joins -- map2 (reference join_mapper, spartan/expr/operator/map.py:243-286), outer (outer.py:12-59)
and shuffle with and without a target (shuffle.py:41-96) -- as programs that BOTH packages can run: the reference
(tests/golden/make_golden.py --joins -> tests/golden/joins_w{1,3,4,8}.npz) and the product on the NumPy tile backend and
on the HIP backend (tests/test_join_programs.py).  The tile functions are written as the reference's users write them
(`tile * 2 + other`, `tile.sum(axis=0, keepdims=True)`, `tile.T`, `tile_a.dot(tile_b)`): they get np.ndarray tiles there
and on the NumPy backend, device arrays that answer the same calls with kernels on the HIP backend.

Half of the programs are ORDER-sensitive on purpose: every source tile writes the WHOLE of a small target -- with a
reducer (float32 partials 2^24, 1, 1, -2^24: the sum depends on the order they meet in) or without one (the last write
stays) -- so the recorded outputs pin the order the reference's kernels run in (worker by worker, a worker's tiles
largest first and last listed first, worker.py:246-256) for every operator that pushes updates, not only for reduce.
"""
import importlib

import numpy as np

F32 = np.float32


def _ext(ex):
  """The extent module of whichever package made `ex`."""
  return importlib.import_module(type(ex).__module__)


def _spikes():
  a = np.zeros((200, 3), F32)
  a[0], a[70], a[140], a[199] = 16777216.0, 1.0, 1.0, -16777216.0
  return a


def _ramp(shape, mod=11, off=5):
  return ((np.arange(int(np.prod(shape)), dtype=F32).reshape(shape) % mod) - off).astype(F32)


# ---- map2 tile functions: fn(extents, tiles, **kw) yields (target extent, data)
def _rows_join(extents, tiles):
  yield extents[0], tiles[0] * 2 + tiles[1]


def _whole_target_colsum(extents, tiles, width=3):
  ex = extents[0]
  yield _ext(ex).create((0,), (width,), (width,)), tiles[0].sum(axis=0)


def _whole_target_first_row_tag(extents, tiles, width=3):
  # which tile wrote last?  every tile writes (its first row number + 1) into all of the target
  ex = extents[0]
  yield _ext(ex).create((0,), (width,), (width,)), tiles[0][0:1, :].reshape(width) * 0 + (ex.ul[0] + 1)


# ---- outer tile function: fn(ex_a, tile_a, ex_b, tile_b) yields (target extent, data)
def _rows_times_all(ex_a, tile_a, ex_b, tile_b):
  yield _ext(ex_a).create((ex_a.ul[0], 0), (ex_a.lr[0], ex_b.lr[1]), (ex_a.array_shape[0], ex_b.array_shape[1])), tile_a.dot(tile_b)


def _pair_partial(ex_a, tile_a, ex_b, tile_b):
  # both operands partitioned: one call per PAIR of tiles (outer.py:30-57); every pair adds into the whole target
  yield _ext(ex_a).create((0,), (3,), (3,)), tile_a.sum(axis=0) * tile_b[0:1, :].reshape(3)


def _pair_tag(ex_a, tile_a, ex_b, tile_b):
  yield _ext(ex_a).create((0,), (3,), (3,)), tile_b[0:1, :].reshape(3) * 0 + (ex_a.ul[0] * 10 + ex_b.ul[0] + 1)


# ---- shuffle tile functions: fn(source, ex, **kw) -> [(target extent, data)]
def _transposed_block(source, ex):
  data = source.fetch(ex)
  return [(_ext(ex).create(ex.ul[::-1], ex.lr[::-1], source.shape[::-1]), data.T)]


def _colsum_row(source, ex):
  data = source.fetch(ex)
  return [(_ext(ex).create((0, ex.ul[1]), (1, ex.lr[1]), (1, source.shape[1])), data.sum(axis=0, keepdims=True))]


def _tag_row(source, ex):
  data = source.fetch(ex)
  return [(_ext(ex).create((0, ex.ul[1]), (1, ex.lr[1]), (1, source.shape[1])), data[0:1, :] * 0 + (ex.ul[0] + 1))]


def programs():
  """(name, build(sp) -> Expr)."""
  P = []
  four = (50, 3)            # four row tiles whatever the worker count
  P.append(('map2_rows_join', lambda sp: sp.map2((sp.from_numpy(_ramp((90, 8))), sp.from_numpy(_ramp((90, 8), 7, 3))), (0, 0),
                                                 fn=_rows_join, shape=(90, 8))))
  P.append(('map2_whole_target_add', lambda sp: sp.map2(sp.from_numpy(_spikes(), tile_hint=four), 0, fn=_whole_target_colsum,
                                                        shape=(3,), reducer=np.add)))
  P.append(('map2_whole_target_last_write', lambda sp: sp.map2(sp.from_numpy(_spikes(), tile_hint=four), 0,
                                                               fn=_whole_target_first_row_tag, shape=(3,))))
  P.append(('map2_whole_target_last_write_default_tiles',
            lambda sp: sp.map2(sp.from_numpy(_ramp((200, 3))), 0, fn=_whole_target_first_row_tag, shape=(3,))))
  P.append(('outer_rows_times_all', lambda sp: sp.outer((sp.from_numpy(_ramp((60, 5))), sp.from_numpy(_ramp((5, 4), 5, 2))), (0, None),
                                                        fn=_rows_times_all, shape=(60, 4))))
  P.append(('outer_pairs_whole_target_add', lambda sp: sp.outer((sp.from_numpy(_spikes(), tile_hint=four), sp.from_numpy(np.ones((8, 3), F32), tile_hint=(4, 3))),
                                                                (0, 0), fn=_pair_partial, shape=(3,), reducer=np.add)))
  P.append(('outer_pairs_whole_target_last_write', lambda sp: sp.outer((sp.from_numpy(_spikes(), tile_hint=four), sp.from_numpy(np.ones((8, 3), F32), tile_hint=(4, 3))),
                                                                       (0, 0), fn=_pair_tag, shape=(3,))))
  P.append(('shuffle_transpose', lambda sp: sp.shuffle(sp.from_numpy(_ramp((60, 28))), _transposed_block, shape_hint=(28, 60))))
  P.append(('shuffle_target_add', lambda sp: sp.shuffle(sp.from_numpy(_spikes(), tile_hint=four), _colsum_row,
                                                        target=sp.ndarray((1, 3), dtype=F32, reduce_fn=np.add))))
  P.append(('shuffle_target_last_write', lambda sp: sp.shuffle(sp.from_numpy(_spikes(), tile_hint=four), _tag_row,
                                                               target=sp.ndarray((1, 3), dtype=F32))))
  return P
