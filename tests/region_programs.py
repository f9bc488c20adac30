"""map2(update_region=...) programs (reference region_join_mapper, spartan/expr/operator/map.py:208-241): a join
that rewrites boxes of its first array, shaped after the three steps of the reference's only user of that branch
(examples/cholesky.py:57-80) with integer-valued tile bodies in place of the LAPACK calls, so results are bit-exact.
Run by the reference (tests/golden/make_golden.py --region -> region_w4.npz) and by the product on the NumPy tile
backend, the HIP backend and two gloo ranks (tests/test_region_join.py).

The branch maps the 1-D tiling of arrays[0] onto a sqrt(W) x sqrt(W) grid (extent.change_partition_axis with a list
of axes) and fails in the reference for an untiled array, so the programs are recorded with 4 workers.
"""
import numpy as np

F32 = np.float32
N, B = 64, 32          # array order; grid cell order with 4 workers


def _a():
  return (np.arange(N * N, dtype=F32).reshape(N, N) % 11) - 5


def _box(sp, ul, lr):
  import importlib
  return importlib.import_module(sp.__name__ + '.array.extent').create(ul, lr, (N, N))


def _double_plus_one(extents, tiles):
  return extents[0], tiles[0] * 2 + 1


def _plus_corner_sum(extents, tiles):
  # tiles[1]: the whole second array (its join axis is None)
  return extents[0], tiles[0] + tiles[1].sum()


def _minus_product(extents, tiles):
  # tiles[1]: columns of the transposed panel under the cell's ROW range; tiles[2]: rows of the panel under the
  # cell's COLUMN range
  return extents[0], tiles[0] - tiles[2].dot(tiles[1]).T


def _minus_seven(extents, tiles):
  return extents[0], -7.0


def programs():
  P = []
  # (name, build(sp) -> Expr)
  P.append(('diag_cell', lambda sp: sp.map2(sp.from_numpy(_a()), ((0, 1),), fn=_double_plus_one, shape=(N, N),
                                            update_region=_box(sp, (B, B), (N, N)))))
  P.append(('column_of_cells_two_arrays',
            lambda sp: sp.map2((sp.from_numpy(_a()), sp.from_numpy(_a()[:B, :B].copy())), ((0, 1), None),
                               fn=_plus_corner_sum, shape=(N, N), update_region=_box(sp, (0, 0), (N, B)))))
  P.append(('three_arrays_int_axes',
            lambda sp: sp.map2((sp.from_numpy(_a()), sp.from_numpy(_a()[:, :B].T.copy()), sp.from_numpy(_a()[:, :B].copy())),
                               ((0, 1), 1, 0), fn=_minus_product, shape=(N, N),
                               update_region=[_box(sp, (B, B), (N, N)), _box(sp, (0, 0), (B, B))])))
  P.append(('unaligned_box_scalar',
            lambda sp: sp.map2(sp.from_numpy(_a()), ((0, 1),), fn=_minus_seven, shape=(N, N),
                               update_region=[_box(sp, (10, 5), (40, 20)), _box(sp, (50, 40), (60, 64))])))
  P.append(('region_misses_every_cell_but_one',
            lambda sp: sp.map2(sp.from_numpy(_a()), ((0, 1),), fn=_double_plus_one, shape=(N, N),
                               update_region=_box(sp, (0, B), (B, N)), reducer=np.add)))
  return P
