"""Host logic (tiling, extents, DAG, fusion, fetch/update plans) with the
NumPy tile backend on CPU, for 1/3/8 logical workers -- the reference's tests
run on 3-worker in-process clusters (tests/test_common.py:128-136)."""
import numpy as np
import pytest

import spartan_amd as sp
from oracle.np_backend import NumpyBackend
from tests import programs

PROGS = programs.programs()


@pytest.fixture(params=[1, 3, 8], ids=lambda n: 'workers%d' % n)
def ctx(request):
  c = sp.initialize(backend=NumpyBackend(), num_workers=request.param)
  yield c
  sp.shutdown()


@pytest.mark.parametrize('prog', PROGS, ids=[p[0] for p in PROGS])
def test_program(ctx, prog):
  name, build, expected, tol = prog
  got = programs.run(name, build, sp, ctx.num_workers)
  programs.check(name, got, expected(), tol)


def test_config1_dtype_and_force(ctx):
  # BASELINE config 1: ones((1000,1000)) + 1 then .force(); fp32 stays fp32
  r = (sp.ones((1000, 1000)) + 1).force()
  assert r.dtype == np.float32
  assert r.shape == (1000, 1000)
  assert len(r.tiles) == max(1, ctx.num_workers if ctx.num_workers != 3 else 4)
  g = r.glom()
  assert g.dtype == np.float32 and np.all(g == 2.0)


def test_eval_cache_and_force_alias(ctx):
  e = sp.ones((10, 10)) + 1
  a = e.evaluate()
  assert e.force() is a        # cached by expr_id (base.py:284-287)
  assert e.optimized().evaluate() is not None


def test_fused_op_strings(ctx):
  # the fused trees the survey observed through the oracle
  e = (sp.ones((4, 4)) + sp.ones((4, 4)) + sp.ones((4, 4)) + sp.ones((4, 4))).optimized()
  s = e.op.pretty_str().replace(' ', '')
  assert s.count('add(') == 3 and s.count('_make_ones(') == 4
  a = sp.ones((4, 4))
  r = sp.sum(a * a + a, axis=0).optimized()
  assert '_sum_local' in r.op.pretty_str() and 'add(' in r.op.pretty_str() and 'multiply(' in r.op.pretty_str()
  assert len(r.children) == 1     # inputs de-duplicated by variable name (optimize.py:119-130)


def test_dot_dispatch(ctx):
  from spartan_amd.expr.map import Map2Expr
  from spartan_amd.expr.outer import OuterProductExpr
  assert isinstance(sp.dot(sp.ones((100, 10)), sp.ones((10, 5))), OuterProductExpr)   # rows > cols
  assert isinstance(sp.dot(sp.ones((10, 100)), sp.ones((100, 5))), Map2Expr)          # rows <= cols
  assert isinstance(sp.dot(sp.ones((10, 10)), np.ones((10, 5), np.float32)), Map2Expr)
  with pytest.raises(ValueError):
    sp.dot(sp.ones((10, 4)), sp.ones((5, 10)))


def test_unknown_reducer_is_loud_on_hip_names():
  from spartan_amd import backend_hip
  with pytest.raises(TypeError):
    backend_hip.HipBackend.reducer_name(None, lambda a, b: a)


def test_pass_through_mapper_does_not_alias_its_input():
  """A user map2 mapper that yields a fetched input tile unchanged, into a one-tile reducer target: the target
  must not adopt the source array's storage (later merges would reduce into the source)."""
  import spartan_amd as sp
  from oracle.np_backend import NumpyBackend
  from spartan_amd.array import extent
  sp.initialize(backend=NumpyBackend(), num_workers=4)
  try:
    x = np.arange(32, dtype=np.float64).reshape(8, 4)
    X = sp.from_numpy(x, tile_hint=(2, 4)).force()

    def passthrough(extents, tiles):
      yield extent.create((0, 0), (2, 4), (2, 4)), tiles[0]
    got = sp.map2(X, 0, fn=passthrough, shape=(2, 4), tile_hint=(2, 4), reducer=np.add).glom()
    np.testing.assert_array_equal(got, x.reshape(4, 2, 4).sum(axis=0))
    np.testing.assert_array_equal(X.glom(), x)          # the input is untouched
  finally:
    sp.shutdown()


def test_rotate_slice_pushes_slices_below_maps():
  """(a + b * 2)[rows, cols] with the slice-rotation pass on: the slice lands on the arrays (the broadcast operand
  through its stretched view), the maps above it fuse into ONE map over the sliced extent, values are NumPy's."""
  import importlib
  import spartan_amd as sp
  from oracle.np_backend import NumpyBackend
  from spartan_amd.expr.map import MapExpr
  from spartan_amd.expr.views import SliceExpr
  opt = importlib.import_module('spartan_amd.expr.optimize')
  a = np.arange(40 * 30, dtype=np.float32).reshape(40, 30) % 11
  b = np.arange(30, dtype=np.float32).reshape(1, 30) - 7
  sp.initialize(backend=NumpyBackend(), num_workers=3)
  opt.FLAGS['opt_rotate_slice'] = True
  try:
    A, B = sp.from_numpy(a).force(), sp.from_numpy(b).force()
    whole = sp.Val(val=A) + sp.Val(val=B) * 2
    for idx in ((slice(5, 25), slice(3, 17)), (slice(0, 40), slice(29, 30)), slice(7, 8)):
      e = whole[idx]
      o = e.optimized()
      assert isinstance(o, MapExpr) and all(isinstance(c, SliceExpr) for c in o.children), o
      np.testing.assert_array_equal(o.glom(), (a + b * 2)[idx])
    # a map consumed whole AND sliced keeps its own value
    np.testing.assert_array_equal(whole.optimized().glom(), a + b * 2)
    np.testing.assert_array_equal((whole[2:4] - whole[3:5]).optimized().glom(), (a + b * 2)[2:4] - (a + b * 2)[3:5])
  finally:
    opt.FLAGS['opt_rotate_slice'] = False
    sp.shutdown()


def test_tile_assignment_strategies(tmp_path, monkeypatch):
  """The reference's five tile-assignment policies (distarray.py:441-476): where the tiles go, and that results
  do not depend on it."""
  import spartan_amd as sp
  from oracle.np_backend import NumpyBackend
  from spartan_amd.array import placement
  a = np.arange(96 * 8, dtype=np.float32).reshape(96, 8)

  def workers_of(strategy, hint=(8, 8)):
    monkeypatch.setattr(placement, 'STRATEGY', strategy)
    sp.initialize(backend=NumpyBackend(), num_workers=4)
    try:
      X = sp.from_numpy(a, tile_hint=hint).force()
      np.testing.assert_array_equal((sp.Val(val=X) * 2).glom(), a * 2)
      np.testing.assert_array_equal(sp.sum(sp.Val(val=X), 0).glom(), a.sum(0))
      return [tid.worker for ex, tid in sorted(X.tiles.items(), key=lambda kv: kv[0].ul)]
    finally:
      sp.shutdown()
  assert workers_of('round_robin') == [0, 1, 2, 3] * 3
  assert workers_of('serpentine') == [0, 1, 2, 3, 3, 2, 1, 0, 0, 1, 2, 3]
  mapping = tmp_path / 'tiles_map'
  mapping.write_text('\n'.join(str(w) for w in [3, 3, 2, 2, 1, 1, 0, 0, 3, 2, 1, 0]) + '\n')
  monkeypatch.setenv('SPARTAN_TILES_MAP', str(mapping))
  assert workers_of('static') == [3, 3, 2, 2, 1, 1, 0, 0, 3, 2, 1, 0]
  placement.seed(5)
  first = workers_of('random')
  placement.seed(5)
  assert workers_of('random') == first and set(first) <= {0, 1, 2, 3} and len(set(first)) > 1
  # 'performance': the least loaded workers first -- worker 0 already holds a big array
  monkeypatch.setattr(placement, 'STRATEGY', 'performance')
  sp.initialize(backend=NumpyBackend(), num_workers=4)
  try:
    big = sp.from_numpy(np.zeros((64, 64), np.float32), tile_hint=(64, 64)).force()      # one tile, on worker 0
    assert [t.worker for t in big.tiles.values()] == [0]
    X = sp.from_numpy(a, tile_hint=(24, 8)).force()                                    # 4 tiles
    got = [tid.worker for ex, tid in sorted(X.tiles.items(), key=lambda kv: kv[0].ul)]
    assert got[-1] == 0 and sorted(got) == [0, 1, 2, 3], got
    np.testing.assert_array_equal(X.glom(), a)
  finally:
    sp.shutdown()
  monkeypatch.setattr(placement, 'STRATEGY', 'no_such_policy')
  sp.initialize(backend=NumpyBackend(), num_workers=2)
  try:
    with pytest.raises(ValueError):
      sp.from_numpy(a).force()
  finally:
    sp.shutdown()


def test_partial_write_across_tiles_reads_masked_and_refuses_kernels():
  """A region that spans a written and a never-written tile (distarray.py:355-365 + tile.pyx:100-113): glom gives
  the reference's MaskedArray -- not zeros --, kernels refuse the masked operand with one explicit error, extents
  take negative axes like the reference's `ul[idx]`."""
  from oracle.np_backend import NumpyBackend
  from spartan_amd.array import extent, tile
  sp.initialize(backend=NumpyBackend(), num_workers=2)
  try:
    be = sp.get_context().backend
    a = sp.ndarray((4, 4), dtype=np.float32).evaluate()
    assert len(a.tiles) == 2
    a.update(extent.create((0, 0), (1, 4), (4, 4)), be.from_numpy(np.ones((1, 4), np.float32)))
    got = a.glom()
    assert isinstance(got, np.ma.MaskedArray)
    np.testing.assert_array_equal(got.mask, np.arange(16).reshape(4, 4) >= 4)
    np.testing.assert_array_equal(got[0].filled(-1), np.ones(4, np.float32))
    for build in (lambda v: v + 1, lambda v: sp.sum(v), lambda v: sp.argmax(v, 1), lambda v: sp.dot(v, v)):
      with pytest.raises(tile.MaskedOperandError):
        build(sp.Val(val=a)).evaluate()
    e = extent.create((1, 2), (3, 5), (10, 10))
    assert e[-1] == e[1] and e[-2] == e[0]
    assert extent.largest_dim_axis(()) == 0
  finally:
    sp.shutdown()


def test_aligned_map_path_is_the_general_path(ctx, monkeypatch):
  """expr/map._evaluate_aligned (same-shape, same-tiling dense operands and scalars) against the general
  tile_mapper / run_kernel / from_table path it short-cuts: same values, dtypes, tile tables and placements, and it
  steps aside (None, nothing done) for what it does not cover -- stretched operands, driver-side arrays, sparse or
  partly written arrays."""
  import importlib
  M = importlib.import_module('spartan_amd.expr.map')
  taken = []
  real = M._evaluate_aligned

  def spy(node, c, values, names):
    out = real(node, c, values, names)
    taken.append(out is not None)
    return out
  a = np.arange(60 * 8, dtype=np.float32).reshape(60, 8) % 7 - 3
  b = np.arange(60 * 8, dtype=np.int64).reshape(60, 8) % 5
  progs = {
      'scalar': lambda A, B: A * 2.5 + 1,
      'two_arrays': lambda A, B: (A + B) * A,
      'compare': lambda A, B: (A > 0) & (B < 3),
      'fused': lambda A, B: ((A * A + A) / (B + 1)).optimized(),
      'row_broadcast': lambda A, B: A + sp.from_numpy(a[:1]),          # stretched operand: general path
      'numpy_operand': lambda A, B: A + a,                             # driver-side array: general path
  }
  results = {}
  for mode in ('aligned', 'general'):
    monkeypatch.setattr(M, '_evaluate_aligned', spy if mode == 'aligned' else (lambda *args: None))
    A, B = sp.from_numpy(a).evaluate(), sp.from_numpy(b).evaluate()
    for name, build in progs.items():
      del taken[:]
      res = build(sp.Val(val=A), sp.Val(val=B)).evaluate()
      if mode == 'aligned':
        assert (True in taken) == (name not in ('row_broadcast', 'numpy_operand')), (name, taken)
      results[(mode, name)] = (res.glom(), sorted((ex.ul, ex.lr, tid.worker) for ex, tid in res.tiles.items()))
  for name in progs:
    got, want = results[('aligned', name)], results[('general', name)]
    assert got[0].dtype == want[0].dtype and got[1] == want[1], name
    np.testing.assert_array_equal(got[0], want[0], err_msg=name)
  # a target that was created empty and only partly written is not "written everywhere": general path
  monkeypatch.setattr(M, '_evaluate_aligned', spy)
  partly = sp.ndarray((60, 8), dtype=np.float32).evaluate()
  del taken[:]
  out = (sp.Val(val=partly) + 1).evaluate()             # (a never-written operand: the general path's business)
  assert True not in taken and out.shape == (60, 8)


def test_aligned_reduce_path_is_the_general_path(ctx, monkeypatch):
  """expr/reduce._evaluate_aligned against the foreach_tile / _reduce_mapper path it short-cuts: same values, dtypes,
  tile tables and placements for every axis, fused prologues with several aligned operands, and a tile hint."""
  import importlib
  R = importlib.import_module('spartan_amd.expr.reduce')
  real = R._evaluate_aligned
  taken = []

  def spy(node, c, values):
    out = real(node, c, values)
    taken.append(out is not None)
    return out
  a = (np.arange(96 * 10, dtype=np.float32).reshape(96, 10) % 11 - 5) / 4
  b = np.arange(96 * 10, dtype=np.int64).reshape(96, 10) % 3
  progs = {}
  for axis in (None, 0, 1):
    progs['sum_%s' % axis] = (lambda axis: lambda A, B: sp.sum(A, axis))(axis)
    progs['max_%s' % axis] = (lambda axis: lambda A, B: sp.max(A * 2 - B, axis).optimized())(axis)
    progs['count_%s' % axis] = (lambda axis: lambda A, B: sp.count_nonzero(B, axis))(axis)
    progs['fused_%s' % axis] = (lambda axis: lambda A, B: sp.sum((A - 0.5) * (A - 0.5) + B, axis).optimized())(axis)
  progs['hinted'] = lambda A, B: sp.sum(A, 0, tile_hint=(5,))
  progs['row_broadcast'] = lambda A, B: sp.sum(A + sp.from_numpy(a[:1]), 0).optimized()      # general path
  results = {}
  for mode in ('aligned', 'general'):
    monkeypatch.setattr(R, '_evaluate_aligned', spy if mode == 'aligned' else (lambda *args: None))
    A, B = sp.from_numpy(a).evaluate(), sp.from_numpy(b).evaluate()
    for name, build in progs.items():
      del taken[:]
      res = build(sp.Val(val=A), sp.Val(val=B)).evaluate()
      if mode == 'aligned':
        assert (True in taken) == (name != 'row_broadcast'), (name, taken)
      results[(mode, name)] = (res.glom(), sorted((ex.ul, ex.lr, tid.worker) for ex, tid in res.tiles.items()))
  for name in progs:
    got, want = results[('aligned', name)], results[('general', name)]
    assert got[0].dtype == want[0].dtype and got[1] == want[1], name
    np.testing.assert_array_equal(got[0], want[0], err_msg=name)


def test_whole_partials_merged_at_once_equal_the_piecewise_merges(ctx, monkeypatch):
  """UpdateBatch._merge_whole_partials (one process, several workers: p whole-array partials into a fresh target cut
  by rows, combined whole and handed to the target tiles as views) against the piece-by-piece merges it replaces:
  bit-identical values, same dtypes and tile tables -- float sums (order matters), max, integer counts, a 2-D
  target."""
  from spartan_amd.array import distarray
  taken = []
  real = distarray.UpdateBatch._merge_whole_partials

  def spy(self, array, items):
    out = real(self, array, items)
    taken.append(out)
    return out
  rng = np.random.RandomState(3)
  a = (rng.rand(640, 24).astype(np.float32) - 0.5) * 1e3
  progs = {
      'sum0': lambda A: sp.sum(A, 0),
      'max0': lambda A: sp.max(A, 0),
      'count0': lambda A: sp.count_nonzero(A > 0, 0),
      'fused0': lambda A: sp.sum(A * A + 1, 0).optimized(),
      'sum1': lambda A: sp.sum(A, 1),                      # partials cover their own rows only: piecewise as before
      'gram': lambda A: sp.dot(sp.transpose(A), A),         # (24, 24) target of a join
  }
  results = {}
  for mode in ('whole', 'piecewise'):
    monkeypatch.setattr(distarray.UpdateBatch, '_merge_whole_partials', spy if mode == 'whole' else (lambda self, array, items: False))
    A = sp.from_numpy(a).evaluate()
    for name, build in progs.items():
      del taken[:]
      res = build(sp.Val(val=A)).evaluate()
      if mode == 'whole' and ctx.num_workers > 1 and name in ('sum0', 'max0', 'count0', 'fused0'):
        assert True in taken, name
      results[(mode, name)] = (res.glom(), sorted((ex.ul, ex.lr, tid.worker) for ex, tid in res.tiles.items()))
  for name in progs:
    got, want = results[('whole', name)], results[('piecewise', name)]
    assert got[0].dtype == want[0].dtype and got[1] == want[1], name
    np.testing.assert_array_equal(got[0], want[0], err_msg=name)
