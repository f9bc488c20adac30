"""Shared test programs: each builds an expression with the Spartan API and
gives the NumPy value it must produce.  They follow the reference's own
operator tests (tests/test_maptiles.py, test_elementwise.py, test_reduce.py,
test_dot.py, test_matmul.py, test_optimization.py, test_creation.py) and are run
against (a) the NumPy tile backend on CPU and (b) the HIP backend on the GPU.

Entry: (name, build(sp) -> Expr, expected() -> ndarray, tol) where tol is None
for bit-exact results or an (rtol, atol) pair, with the tolerance stated by
SURVEY 8c for that operator class.
"""
import numpy as np

F32 = np.float32
SUM_TOL = (1e-6, 1e-5)      # |d| <= 1e-6 * sum|x| (different summation tree)
ULP = (3e-7, 0)             # <= 2 ulp for exp/log/sqrt/div


def _ar(shape, dtype=F32):
  return np.arange(int(np.prod(shape)), dtype=dtype).reshape(shape)


def programs():
  P = []
  add = P.append
  # ---- creation / plumbing (BASELINE config 1)
  add(('ones_plus_one', lambda sp: sp.ones((1000, 1000)) + 1, lambda: np.full((1000, 1000), 2, F32), None))
  add(('zeros', lambda sp: sp.zeros((33, 7)), lambda: np.zeros((33, 7), F32), None))
  add(('full', lambda sp: sp.full((12, 5), 3.5), lambda: np.full((12, 5), 3.5, F32), None))
  add(('arange_2d_f64', lambda sp: sp.arange((40, 30)), lambda: _ar((40, 30), np.float64), None))
  add(('arange_start_step', lambda sp: sp.arange((13, 5), -1, step=2, dtype=np.int64),
       lambda: np.arange(-1, -1 + 2 * 65, 2, dtype=np.int64).reshape(13, 5), None))
  add(('arange_1d', lambda sp: sp.arange(None, stop=100, dtype=F32), lambda: np.arange(100, dtype=F32), None))
  add(('eye', lambda sp: sp.eye(17, 9, k=1), lambda: np.eye(17, 9, k=1, dtype=F32), None))
  add(('from_numpy', lambda sp: sp.from_numpy(_ar((31, 11))) * 2, lambda: _ar((31, 11)) * 2, None))
  # ---- elementwise (tests/test_maptiles.py, test_elementwise.py)
  add(('add2', lambda sp: sp.ones((100, 10)) + sp.ones((100, 10)), lambda: np.full((100, 10), 2, F32), None))
  add(('add_many', lambda sp: sp.ones((100, 10)) + sp.ones((100, 10)) + sp.ones((100, 10)) + sp.ones((100, 10)),
       lambda: np.full((100, 10), 4, F32), None))
  add(('xx_plus_x', lambda sp: sp.arange((64, 48), dtype=F32) * sp.arange((64, 48), dtype=F32) + sp.arange((64, 48), dtype=F32),
       lambda: _ar((64, 48)) * _ar((64, 48)) + _ar((64, 48)), None))
  add(('sub_rsub_neg', lambda sp: -(3 - sp.arange((20, 7), dtype=F32)) - 1.5,
       lambda: -(3 - _ar((20, 7))) - F32(1.5), None))
  add(('div', lambda sp: sp.arange((50, 4), dtype=F32) / 7, lambda: _ar((50, 4)) / F32(7), ULP))
  add(('ln_exp_sqrt', lambda sp: sp.sqrt(sp.exp(sp.ln(sp.arange((30, 10), dtype=F32) + 1))),
       lambda: np.sqrt(np.exp(np.log(_ar((30, 10)) + 1))), (2e-6, 0)))
  add(('pow_square_abs', lambda sp: sp.abs(sp.square(sp.arange((9, 9), dtype=F32) - 40) ** 2 - 1000),
       lambda: np.abs(np.square(_ar((9, 9)) - 40) ** 2 - 1000), (1e-6, 0)))
  add(('max_min', lambda sp: sp.maximum(sp.arange((10, 10), dtype=F32), 50) + sp.minimum(sp.arange((10, 10), dtype=F32), 20),
       lambda: np.maximum(_ar((10, 10)), 50) + np.minimum(_ar((10, 10)), 20), None))
  add(('compare', lambda sp: (sp.arange((25, 8), dtype=F32) > 50), lambda: _ar((25, 8)) > 50, None))
  add(('logical', lambda sp: (sp.arange((25, 8), dtype=F32) > 50) & (sp.arange((25, 8), dtype=F32) < 100),
       lambda: (_ar((25, 8)) > 50) & (_ar((25, 8)) < 100), None))
  add(('astype', lambda sp: sp.astype(sp.arange((25, 8), dtype=F32) * 0.75, np.int32),
       lambda: (_ar((25, 8)) * F32(0.75)).astype(np.int32), None))
  add(('int_mod_floordiv', lambda sp: (sp.arange((25, 8), dtype=np.int64) - 100) % 7 + (sp.arange((25, 8), dtype=np.int64) - 100) // 7,
       lambda: (_ar((25, 8), np.int64) - 100) % 7 + (_ar((25, 8), np.int64) - 100) // 7, None))
  add(('mixed_f32_i64', lambda sp: sp.arange((25, 8), dtype=F32) + sp.arange((25, 8), dtype=np.int64),
       lambda: _ar((25, 8)) + _ar((25, 8), np.int64), None))
  # ---- broadcasting (tests/test_broadcast.py, test_maptiles.py:test_broadcast)
  add(('bcast_row', lambda sp: sp.arange((64, 10), dtype=F32) + sp.arange((1, 10), dtype=F32),
       lambda: _ar((64, 10)) + _ar((1, 10)), None))
  add(('bcast_col', lambda sp: sp.arange((64, 10), dtype=F32) * sp.arange((64, 1), dtype=F32),
       lambda: _ar((64, 10)) * _ar((64, 1)), None))
  add(('bcast_vec', lambda sp: sp.arange((64, 10), dtype=F32) - sp.arange((10,), dtype=F32),
       lambda: _ar((64, 10)) - _ar((10,)), None))
  add(('bcast_3d', lambda sp: sp.arange((6, 7, 8), dtype=F32) + sp.arange((7, 1), dtype=F32),
       lambda: _ar((6, 7, 8)) + _ar((7, 1)), None))
  add(('bcast_numpy', lambda sp: sp.arange((64, 10), dtype=F32) + _ar((10,)), lambda: _ar((64, 10)) + _ar((10,)), None))
  # ---- reductions (tests/test_reduce.py:14-107)
  for axis in (None, 0, 1):
    add(('sum_2d_%s' % axis, (lambda axis: lambda sp: sp.sum(sp.arange((137, 33), dtype=F32) / 1000, axis))(axis),
         (lambda axis: lambda: (_ar((137, 33)) / F32(1000)).astype(np.float64).sum(axis))(axis), SUM_TOL))
    add(('max_2d_%s' % axis, (lambda axis: lambda sp: sp.max(sp.arange((137, 33), dtype=F32) % 17, axis))(axis),
         (lambda axis: lambda: (_ar((137, 33)) % 17).max(axis))(axis), None))
    add(('min_2d_%s' % axis, (lambda axis: lambda sp: sp.min((sp.arange((137, 33), dtype=F32) - 99) % 17, axis))(axis),
         (lambda axis: lambda: ((_ar((137, 33)) - 99) % 17).min(axis))(axis), None))
    add(('argmax_2d_%s' % axis, (lambda axis: lambda sp: sp.argmax(sp.arange((137, 33), dtype=F32) % 19, axis))(axis),
         (lambda axis: lambda: np.argmax(_ar((137, 33)) % 19, axis))(axis), None))
    add(('argmin_2d_%s' % axis, (lambda axis: lambda sp: sp.argmin((sp.arange((137, 33), dtype=F32) + 5) % 19, axis))(axis),
         (lambda axis: lambda: np.argmin((_ar((137, 33)) + 5) % 19, axis))(axis), None))
    add(('count_nonzero_%s' % axis, (lambda axis: lambda sp: sp.count_nonzero(sp.arange((37, 11), dtype=F32) % 3, axis))(axis),
         (lambda axis: lambda: np.count_nonzero(_ar((37, 11)) % 3, axis))(axis), None))
    add(('count_zero_%s' % axis, (lambda axis: lambda sp: sp.count_zero(sp.arange((37, 11), dtype=F32) % 3, axis))(axis),
         (lambda axis: lambda: (_ar((37, 11)) % 3 == 0).sum(axis))(axis), None))
    add(('all_any_%s' % axis, (lambda axis: lambda sp: sp.all(sp.arange((37, 11), dtype=F32) > 5, axis) | sp.any(sp.arange((37, 11), dtype=F32) > 400, axis))(axis),
         (lambda axis: lambda: np.all(_ar((37, 11)) > 5, axis) | np.any(_ar((37, 11)) > 400, axis))(axis), None))
    add(('mean_%s' % axis, (lambda axis: lambda sp: sp.mean(sp.arange((64, 16), dtype=F32), axis))(axis),
         (lambda axis: lambda: _ar((64, 16)).astype(np.float64).mean(axis))(axis), SUM_TOL))
  for axis in (None, 0, 1, 2):
    add(('sum_3d_%s' % axis, (lambda axis: lambda sp: sp.sum(sp.arange((11, 12, 13), dtype=np.int64), axis))(axis),
         (lambda axis: lambda: _ar((11, 12, 13), np.int64).sum(axis))(axis), None))
    add(('argmax_3d_%s' % axis, (lambda axis: lambda sp: sp.argmax(sp.arange((11, 12, 13), dtype=F32) % 23, axis))(axis),
         (lambda axis: lambda: np.argmax(_ar((11, 12, 13)) % 23, axis))(axis), None))
  add(('sum_1d', lambda sp: sp.sum(sp.arange((1000,), dtype=np.int64)), lambda: _ar((1000,), np.int64).sum(), None))
  add(('prod_int32', lambda sp: sp.prod(sp.astype(sp.arange((3, 4), dtype=F32) % 3 + 1, np.int32), 1),
       lambda: (_ar((3, 4)) % 3 + 1).astype(np.int32).prod(1, dtype=np.int64), None))
  add(('std', lambda sp: sp.std(sp.arange((64, 16), dtype=F32), 0), lambda: _ar((64, 16)).std(0), (1e-6, 1e-6)))  # E[x^2]-E[x]^2 formula (statistics.py:102)
  # ---- fusion (tests/test_optimization.py:124-163)
  add(('opt_map_chain', lambda sp: (sp.ones((50, 50)) + sp.ones((50, 50)) + sp.ones((50, 50)) + sp.ones((50, 50))).optimized(),
       lambda: np.full((50, 50), 4, F32), None))
  add(('opt_reduce_map', lambda sp: sp.sum(sp.arange((137, 33), dtype=F32) * sp.arange((137, 33), dtype=F32) / 1e6 + sp.arange((137, 33), dtype=F32) / 1e3, 0).optimized(),
       lambda: (_ar((137, 33)).astype(np.float64) ** 2 / 1e6 + _ar((137, 33)) / 1e3).sum(0), (1e-5, 1e-5)))
  # (the broadcast operand is loaded, not generated: a fused map_with_location sees
  # the DRIVING tile's extent, in the reference too -- optimize.py:180-181)
  add(('opt_lreg_grad', lambda sp: sp.sum(sp.arange((200, 16), dtype=F32) / 100 * (sp.from_numpy(_ar((200, 1))) / 50 - 1), 0).optimized(),
       lambda: (_ar((200, 16)).astype(np.float64) / 100 * (_ar((200, 1)) / 50 - 1)).sum(0), (1e-5, 1e-4)))
  # ---- dot (tests/test_dot.py:8-103, test_matmul.py)
  small = lambda shape: _ar(shape) % 5 - 2   # small integers: exact in fp32
  sm = lambda sp, shape: sp.arange(shape, dtype=F32) % 5 - 2
  add(('dot_tall', lambda sp: sp.dot(sm(sp, (100, 40)), sm(sp, (40, 60))), lambda: small((100, 40)).dot(small((40, 60))), None))
  add(('dot_wide', lambda sp: sp.dot(sm(sp, (40, 100)), sm(sp, (100, 60))), lambda: small((40, 100)).dot(small((100, 60))), None))
  add(('dot_square', lambda sp: sp.dot(sm(sp, (128, 128)), sm(sp, (128, 128))), lambda: small((128, 128)).dot(small((128, 128))), None))
  add(('dot_numpy_rhs', lambda sp: sp.dot(sm(sp, (100, 40)), small((40, 8))), lambda: small((100, 40)).dot(small((40, 8))), None))
  add(('dot_numpy_vec', lambda sp: sp.dot(sm(sp, (100, 40)), small((40,))), lambda: small((100, 40)).dot(small((40,))), None))
  add(('dot_2d_vec', lambda sp: sp.dot(sm(sp, (100, 40)), sm(sp, (40,))), lambda: small((100, 40)).dot(small((40,))), None))
  add(('dot_vec_2d', lambda sp: sp.dot(sm(sp, (40,)), sm(sp, (40, 60))), lambda: small((40,)).dot(small((40, 60))), None))
  add(('dot_vec_vec', lambda sp: sp.dot(sm(sp, (333,)), sm(sp, (333,))), lambda: np.asarray([small((333,)).dot(small((333,)))]), None))
  add(('dot_tile_hint', lambda sp: sp.dot(sm(sp, (64, 128)), sm(sp, (128, 64)), tile_hint=(8, 64)),
       lambda: small((64, 128)).dot(small((128, 64))), None))
  add(('dot_f64', lambda sp: sp.dot(sp.arange((30, 20)), sp.arange((20, 10))), lambda: _ar((30, 20), np.float64).dot(_ar((20, 10), np.float64)), None))
  add(('lreg_step', lambda sp: sp.sum(sm(sp, (256, 16)) * (sp.dot(sm(sp, (256, 16)), small((16, 1))) - sm(sp, (256, 1))), axis=0).optimized(),
       lambda: (small((256, 16)) * (small((256, 16)).dot(small((16, 1))) - small((256, 1)))).sum(0), None))
  # ---- views: Slice / Transpose / Reshape (tests/test_slice.py, test_transpose.py, test_reshape.py)
  A = lambda sp: sp.arange((40, 30), dtype=F32)
  a = lambda: _ar((40, 30))
  add(('slice_rows', lambda sp: A(sp)[5:30] + 1, lambda: a()[5:30] + 1, None))
  add(('slice_2d', lambda sp: A(sp)[2:30, 3:17] * 2, lambda: a()[2:30, 3:17] * 2, None))
  add(('slice_col_sum', lambda sp: sp.sum(A(sp)[:, 4:20], 0), lambda: a()[:, 4:20].sum(0), None))
  add(('slice_of_slice', lambda sp: A(sp)[4:36][3:20, 1:9] - 3, lambda: a()[4:36][3:20, 1:9] - 3, None))
  add(('int_index', lambda sp: A(sp)[3] * 1, lambda: a()[3:4] * 1, None))   # reference keeps the axis (base.py:437-440)
  add(('tuple_int_index', lambda sp: A(sp)[:, 7] + 0, lambda: a()[:, 7] + 0, None))
  add(('transpose_map', lambda sp: sp.transpose(A(sp)) + 1, lambda: a().T + 1, None))
  add(('transpose_sum', lambda sp: sp.sum(sp.transpose(A(sp)), 1), lambda: a().T.sum(1), None))
  small2 = lambda: _ar((40, 30)) % 5 - 2
  S2 = lambda sp: sp.arange((40, 30), dtype=F32) % 5 - 2
  add(('dot_a_at', lambda sp: sp.dot(S2(sp), sp.transpose(S2(sp))), lambda: small2().dot(small2().T), None))
  add(('dot_at_a', lambda sp: sp.dot(sp.transpose(S2(sp)), S2(sp)), lambda: small2().T.dot(small2()), None))
  add(('reshape_flat', lambda sp: sp.reshape(A(sp), (1200,)) * 2, lambda: a().reshape(1200) * 2, None))
  add(('reshape_2d', lambda sp: sp.reshape(A(sp), (30, 40)) + 1, lambda: a().reshape(30, 40) + 1, None))
  add(('reshape_add_dim', lambda sp: sp.reshape(A(sp), (40, 30, 1)) + 1, lambda: a().reshape(40, 30, 1) + 1, None))
  add(('ravel_sum', lambda sp: sp.sum(sp.ravel(A(sp))), lambda: a().ravel().sum(), None))
  # ---- the rest of the builder namespace (tests/test_statistics.py:10-14, test_creation.py:67-81,
  # test_manipulation.py:19-37; statistics.py:105-219, creation.py:225-330, manipulation.py:44-80)
  lab = lambda: (np.arange(200, dtype=np.int64) * 7) % 11 + 1
  wts = lambda: (np.arange(200, dtype=F32) % 9) / 4
  add(('bincount', lambda sp: sp.bincount(sp.from_numpy(lab())), lambda: np.bincount(lab()), None))
  add(('bincount_minlength', lambda sp: sp.bincount(sp.from_numpy(lab()), minlength=20), lambda: np.bincount(lab(), minlength=20), None))
  add(('bincount_weights', lambda sp: sp.bincount(sp.from_numpy(lab()), sp.from_numpy(wts())),
       lambda: None, (1e-12, 0)))      # (the target's dtype is the labels': what the reference returns is pinned by the goldens)
  pos = lambda: _ar((40, 30)) % 13 + 1
  add(('normalize_all', lambda sp: sp.normalize(sp.from_numpy(pos())), lambda: pos() / pos().sum(), (1e-6, 0)))
  # (axis 0 / 1 divide the first column / row of every TILE only, statistics.py:157-160: the value depends on the
  # tiling, so only the reference's outputs can say what it is)
  add(('normalize_axis0', lambda sp: sp.normalize(sp.from_numpy(pos()), 0), lambda: None, (1e-6, 0)))
  add(('normalize_axis1', lambda sp: sp.normalize(sp.from_numpy(pos()), 1), lambda: None, (1e-6, 0)))
  # norm returns a NumPy value, not an expression: wrapped so that every program yields an array
  sgn = lambda: (_ar((40, 30)) % 7 - 3) / 8
  add(('norm1_matrix', lambda sp: sp.from_numpy(np.atleast_1d(sp.norm(sp.from_numpy(sgn()), 1))) * 1,
       lambda: np.atleast_1d(np.abs(sgn()).sum(0).max()), SUM_TOL))
  add(('norm1_vector', lambda sp: sp.from_numpy(np.atleast_1d(sp.norm(sp.from_numpy(sgn().ravel()), 1))) * 1,
       lambda: np.atleast_1d(np.abs(sgn()).sum()), SUM_TOL))
  add(('norm2_vector', lambda sp: sp.from_numpy(np.atleast_1d(sp.norm(sp.from_numpy(sgn().ravel())))) * 1,
       lambda: np.atleast_1d(np.sqrt(np.square(sgn()).sum())), SUM_TOL))
  add(('norm2_column', lambda sp: sp.from_numpy(np.atleast_1d(sp.norm(sp.from_numpy(sgn()[:, :1])))) * 1,
       lambda: np.atleast_1d(np.sqrt(np.square(sgn()[:, :1]).sum())), SUM_TOL))
  for tag, shape in (('square', (16, 16)), ('tall', (15, 10)), ('wide', (10, 15)), ('big', (64, 48))):
    add(('diagonal_' + tag, (lambda shape: lambda sp: sp.diagonal(sp.from_numpy(_ar(shape))))(shape),
         (lambda shape: lambda: np.diagonal(_ar(shape)))(shape), None))
  add(('diagonal_method', lambda sp: (sp.arange((20, 20), dtype=F32) + 1).diagonal(), lambda: np.diagonal(_ar((20, 20)) + 1), None))
  add(('diag_1d', lambda sp: sp.diag(sp.from_numpy(_ar((24,)) + 1)), lambda: None, None))
  add(('diag_2d', lambda sp: sp.diag(sp.from_numpy(_ar((12, 18)))), lambda: np.diag(_ar((12, 18))), None))
  add(('diagflat_2d', lambda sp: sp.diagflat(sp.from_numpy(_ar((16, 2)) + 1)), lambda: None, None))
  add(('concatenate_1d', lambda sp: sp.concatenate(sp.from_numpy(_ar((40,))), sp.from_numpy(_ar((40,)) * 2)),
       lambda: np.concatenate((_ar((40,)), _ar((40,)) * 2)), None))
  add(('concatenate_2d_axis0', lambda sp: sp.concatenate(sp.from_numpy(_ar((32, 32))), sp.from_numpy(_ar((32, 32)) + 5)),
       lambda: np.concatenate((_ar((32, 32)), _ar((32, 32)) + 5)), None))
  add(('concatenate_2d_axis1', lambda sp: sp.concatenate(sp.from_numpy(_ar((32, 32))), sp.from_numpy(_ar((32, 32)) + 5), 1),
       lambda: np.concatenate((_ar((32, 32)), _ar((32, 32)) + 5), 1), None))
  add(('concatenate_ragged', lambda sp: sp.concatenate(sp.from_numpy(_ar((15, 5))), sp.from_numpy(_ar((15, 7)) - 3), 1),
       lambda: np.concatenate((_ar((15, 5)), _ar((15, 7)) - 3), 1), None))
  add(('concatenate_exprs', lambda sp: sp.concatenate(sp.arange((30, 8), dtype=F32) * 2, sp.ones((12, 8)), 0) + 1,
       lambda: np.concatenate((_ar((30, 8)) * 2, np.ones((12, 8), F32)), 0) + 1, None))
  # ---- the ORDER in which the partials of a reduction meet (round 6): four row tiles whatever the worker count
  # (tile_hint), ONE non-zero row each -- every per-tile partial is exact in any order --, magnitudes chosen so that
  # float32 addition of the four partials gives different answers in different orders: 2^24 + 1 == 2^24.  The tiles
  # sit on workers 0, 1, 2, 0 with three workers, and the reference's kernels run worker by worker (blob_ctx.py:
  # 270-271): (t0 + t3) + t1 + t2 there, t0 + t1 + t2 + t3 with 1, 4 and 8 workers.  Values: the recorded ones.
  def spikes(scale=1.0):
    a = np.zeros((200, 3), F32)
    a[0], a[70], a[140], a[199] = 16777216.0 * scale, 1.0 * scale, 1.0 * scale, -16777216.0 * scale
    return a
  add(('sum_axis0_partials_meet_in_worker_order', lambda sp: sp.sum(sp.from_numpy(spikes(), tile_hint=(50, 3)), 0),
       lambda: None, None))
  add(('sum_all_partials_meet_in_worker_order', lambda sp: sp.sum(sp.from_numpy(spikes(), tile_hint=(50, 3))) * sp.ones((2,)),
       lambda: None, None))
  add(('max_of_scaled_sum_in_worker_order', lambda sp: sp.max(sp.sum(sp.from_numpy(spikes(0.5), tile_hint=(50, 3)) * 2, 0).optimized()) * sp.ones((2,)),
       lambda: None, None))
  # the same for the K-split join of spartan.dot (dot.py:195-217): four K slabs of a wide matrix, one non-zero product
  # per slab and output element, 2^12 * 2^12 + 1 + 1 - 2^12 * 2^12 in the order the slabs' partial products meet
  def wide():
    a = np.zeros((4, 200), F32)
    a[:, 0], a[:, 70], a[:, 140], a[:, 199] = 4096.0, 1.0, 1.0, -4096.0
    return a
  def tall():
    b = np.zeros((200, 2), F32)
    b[0], b[70], b[140], b[199] = 4096.0, 1.0, 1.0, 4096.0
    return b
  add(('dot_ksplit_partials_meet_in_worker_order',
       lambda sp: sp.dot(sp.from_numpy(wide(), tile_hint=(4, 50)), sp.from_numpy(tall(), tile_hint=(50, 2))), lambda: None, None))
  add(('dot_ksplit_numpy_rhs_in_worker_order',
       lambda sp: sp.dot(sp.from_numpy(wide(), tile_hint=(4, 50)), tall()), lambda: None, None))
  # ties between tiles: the same extreme value in tiles 0, 1 and 3 (tiles 0 and 3 on one worker with three workers),
  # reduced in kernel order -- which occurrence does the pairwise reducer keep?  (NumPy: the first.)
  def ties():
    a = np.zeros((200, 3), F32)
    a[10], a[60], a[160] = 5.0, 5.0, 5.0
    a[20], a[170] = -5.0, -5.0
    return a
  add(('argmax_ties_across_tiles_axis0', lambda sp: sp.argmax(sp.from_numpy(ties(), tile_hint=(50, 3)), 0), lambda: np.argmax(ties(), 0), None))
  add(('argmin_ties_across_tiles_axis0', lambda sp: sp.argmin(sp.from_numpy(ties(), tile_hint=(50, 3)), 0), lambda: np.argmin(ties(), 0), None))
  add(('argmax_ties_across_tiles_flat', lambda sp: sp.argmax(sp.from_numpy(ties(), tile_hint=(50, 3))) * sp.ones((2,), dtype=np.int64),
       lambda: np.argmax(ties()) * np.ones((2,), np.int64), None))
  add(('argmin_ties_across_tiles_flat', lambda sp: sp.argmin(sp.from_numpy(ties(), tile_hint=(50, 3))) * sp.ones((2,), dtype=np.int64),
       lambda: np.argmin(ties()) * np.ones((2,), np.int64), None))
  # NaN and infinities through the reductions and the element-wise extremes (NumPy's rules: max / min / sum propagate
  # NaN, argmax / argmin take the first NaN, np.maximum propagates it, comparisons with it are false) -- per tile and
  # across tiles, the NaN in one tile and the largest finite value in another
  def holes():
    a = (np.arange(200 * 3, dtype=F32).reshape(200, 3) % 13) - 6
    a[30, 1] = np.nan
    a[170, 2] = np.inf
    a[120, 0] = -np.inf
    return a
  hx = lambda sp: sp.from_numpy(holes(), tile_hint=(50, 3))
  for nm, fn in (('max', 'max'), ('min', 'min'), ('sum', 'sum'), ('argmax', 'argmax'), ('argmin', 'argmin')):
    for axis in (None, 0, 1):
      tag = 'nan_%s_axis%s' % (nm, axis)
      if axis is None:
        build = (lambda fn: lambda sp: getattr(sp, fn)(hx(sp)) * sp.ones((2,), dtype=np.int64 if fn.startswith('arg') else F32))(fn)
        want = (lambda fn: lambda: getattr(np, fn)(holes()) * np.ones((2,), np.int64 if fn.startswith('arg') else F32))(fn)
      else:
        build = (lambda fn, axis: lambda sp: getattr(sp, fn)(hx(sp), axis))(fn, axis)
        want = (lambda fn, axis: lambda: getattr(np, fn)(holes(), axis))(fn, axis)
      # (an arg-reduction that meets a NaN returns the reference's SENTINEL, the array's size -- its pairwise reducer
      #  never prefers a NaN and the first tile's own np.argmax index is folded away --, not NumPy's first NaN: the
      #  recorded outputs define these, tests/test_golden.py)
      add((tag, build, (lambda: None) if fn.startswith('arg') else want, None))
  add(('nan_maximum_minimum', lambda sp: sp.maximum(hx(sp), 0.0) + sp.minimum(hx(sp), 1.0), lambda: np.maximum(holes(), F32(0)) + np.minimum(holes(), F32(1)), None))
  add(('nan_comparisons', lambda sp: (hx(sp) > 0) * 1 + (hx(sp) == hx(sp)) * 2 + (hx(sp) < 2) * 4,
       lambda: (holes() > 0) * 1 + (holes() == holes()) * 2 + (holes() < 2) * 4, None))
  add(('nan_abs_sqrt_square', lambda sp: sp.sqrt(sp.abs(hx(sp))) + sp.square(hx(sp)), lambda: np.sqrt(np.abs(holes())) + np.square(holes()), None))
  add(('nan_fused_sum_of_product', lambda sp: sp.sum(hx(sp) * hx(sp) + 1, 0).optimized(), lambda: (holes() * holes() + 1).sum(0), None))
  # a ragged last tile: the worker sorts its tiles by SIZE before popping them (worker.py:250), so the small tile of
  # rows 200..202 runs last on its worker whatever its place in the list -- 2^24 in tile 0, 1 in tiles 1..3, -2^24 in
  # the small one: the answer depends on where the small tile's partial arrives
  def ragged_spikes():
    a = np.zeros((203, 3), F32)
    a[0], a[70], a[140], a[190], a[200] = 16777216.0, 1.0, 1.0, 1.0, -16777216.0
    return a
  add(('sum_axis0_ragged_tile_runs_last', lambda sp: sp.sum(sp.from_numpy(ragged_spikes(), tile_hint=(50, 3)), 0), lambda: None, None))
  add(('sum_all_ragged_tile_runs_last', lambda sp: sp.sum(sp.from_numpy(ragged_spikes(), tile_hint=(50, 3))) * sp.ones((2,)), lambda: None, None))
  # ---- assign / write (tests/test_assign.py:10-83, test_write.py:30-70): the values are NumPy's assignment; what the
  # recordings add is the result's dtype when array and value differ, and its tile table
  add(('assign_row_f64_into_f32', lambda sp: sp.assign(sp.from_numpy(_ar((20, 10))), np.s_[10, ], np.arange(10, dtype=np.float64) / 4),
       lambda: None, None))
  add(('assign_box_from_expr_slice', lambda sp: sp.assign(sp.from_numpy(_ar((200, 100))), np.s_[99:102, 25:75], sp.from_numpy(_ar((6, 100)) * 2)[2:5, 10:60]),
       lambda: None, None))
  add(('assign_int_into_float', lambda sp: sp.assign(sp.from_numpy(_ar((64, 8))), np.s_[5:9, 2:4], sp.from_numpy(np.arange(8, dtype=np.int64).reshape(4, 2))),
       lambda: None, None))
  def quadrants(sp):
    q = [(slice(0, 50), slice(0, 50)), (slice(0, 50), slice(50, 100)), (slice(50, 100), slice(0, 50)), (slice(50, 100), slice(50, 100))]
    t = sp.zeros((100, 100))
    src = (_ar((100, 100)) % 13).astype(np.float64)
    for k, box in enumerate(q):
      t = sp.write(t, box, src + k, box)
    return t
  add(('write_quadrants_f64_into_f32', quadrants, lambda: None, None))
  # ---- integer results keep the operand's width (reduce.py:102 dtype_fn = the input's dtype; tile.pyx:267 casts every
  # partial to it): int32 sums and products WRAP where NumPy's own sum would have promoted to int64
  def big32():
    return (np.arange(40 * 6, dtype=np.int64).reshape(40, 6) % 7 + 2**29).astype(np.int32)
  add(('int32_sum_wraps_axis0', lambda sp: sp.sum(sp.from_numpy(big32()), 0), lambda: None, None))
  add(('int32_sum_wraps_all', lambda sp: sp.sum(sp.from_numpy(big32())) * sp.ones((2,), dtype=np.int32), lambda: None, None))
  add(('int32_prod_wraps_axis1', lambda sp: sp.from_numpy((np.arange(12 * 5, dtype=np.int32).reshape(12, 5) % 5 + 300).astype(np.int32)).prod(1),
       lambda: None, None))
  add(('uint8_like_bool_sum', lambda sp: sp.sum(sp.from_numpy((np.arange(300 * 4).reshape(300, 4) % 3 == 0)), 0), lambda: None, None))
  add(('int64_mean_axis0', lambda sp: sp.mean(sp.from_numpy(np.arange(50 * 4, dtype=np.int64).reshape(50, 4) * 3), 0), lambda: None, None))
  return P


# programs the reference cannot run on an array of more than one tile (it fails an assertion, and so does the
# product): tests/golden/programs_meta.json records them as skipped for those worker counts
ONE_TILE_ONLY = {'bincount_weights': AssertionError}


def run(name, build, sp, workers):
  """build(sp).glom() -- or None after checking that a ONE_TILE_ONLY program fails as the reference does."""
  if name in ONE_TILE_ONLY and workers > 1:
    try:
      build(sp).glom()
    except ONE_TILE_ONLY[name]:
      return None
    raise AssertionError('%s: expected %s on %d workers' % (name, ONE_TILE_ONLY[name].__name__, workers))
  return build(sp).glom()


def check(name, got, want, tol):
  if want is None or got is None:   # a value only the reference's outputs define (tests/test_golden.py checks those)
    return
  got = np.asarray(got)
  want = np.asarray(want)
  assert got.shape == want.shape, '%s: shape %s != %s' % (name, got.shape, want.shape)
  if tol is None:
    assert got.dtype.kind == want.dtype.kind or (got.dtype.kind in 'iu' and want.dtype.kind in 'iu'), \
        '%s: dtype kind %s != %s' % (name, got.dtype, want.dtype)
    np.testing.assert_array_equal(got, want, err_msg=name)
  else:
    rtol, atol = tol
    np.testing.assert_allclose(got.astype(np.float64), want.astype(np.float64), rtol=rtol, atol=atol, err_msg=name)
