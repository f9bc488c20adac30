"""The plan table (spartan_amd/expr/plan.py): a DAG with the structure of one optimised before is answered by the
recorded result instantiated over ITS leaves.  What must hold: same values as the long way for every variation a
driver loop produces (other arrays, other driver-side operands, other scalars), no confusion between DAGs that differ
only in which leaves are the same object, values shared between a node and its optimised twin, nothing kept alive,
and everything the walk cannot describe optimised the long way."""
import gc
import importlib
import weakref

import numpy as np
import pytest

import spartan_amd as sp

plan = importlib.import_module('spartan_amd.expr.plan')
optimize = importlib.import_module('spartan_amd.expr.optimize')


@pytest.fixture
def ctx():
  from oracle.np_backend import NumpyBackend
  c = sp.initialize(backend=NumpyBackend(), num_workers=3)
  plan.clear()
  for k in plan.stats:
    plan.stats[k] = 0
  yield c
  sp.shutdown()


def _arr(seed, shape=(30, 8)):
  return (np.random.RandomState(seed).randint(-4, 5, size=shape)).astype(np.float32)


def test_same_structure_other_leaves_hits_and_computes_on_the_new_leaves(ctx):
  a, b = _arr(1), _arr(2)
  A, B = sp.from_numpy(a), sp.from_numpy(b)
  got1 = ((A * A + A) * 0.5 - A).optimized().glom()
  assert plan.stats == {'hits': 0, 'misses': 1, 'unplannable': 0}
  got2 = ((B * B + B) * 0.5 - B).optimized().glom()
  assert plan.stats['hits'] == 1
  np.testing.assert_array_equal(got1, (a * a + a) * np.float32(0.5) - a)
  np.testing.assert_array_equal(got2, (b * b + b) * np.float32(0.5) - b)
  # the instantiated DAG is the fused one: ONE map whose inputs are the leaves themselves
  e = ((B * B + B) * 0.5 - B).optimized()
  assert type(e).__name__ == 'MapExpr' and e._is_optimized
  assert all(type(c).__name__ in ('Val', 'AsArray') for c in e.children)
  assert all(c.val is B.val for c in e.children if type(c).__name__ == 'Val')


def test_which_leaves_are_the_same_object_is_part_of_the_key(ctx):
  a, b = _arr(3), _arr(4)
  A, B = sp.from_numpy(a), sp.from_numpy(b)
  np.testing.assert_array_equal((A * A + 1).optimized().glom(), a * a + 1)
  np.testing.assert_array_equal((A * B + 1).optimized().glom(), a * b + 1)      # not the plan of A * A
  np.testing.assert_array_equal((B * A + 1).optimized().glom(), b * a + 1)
  np.testing.assert_array_equal((B * B + 1).optimized().glom(), b * b + 1)
  assert plan.stats['misses'] == 2 and plan.stats['hits'] == 2
  # the same ARRAY behind two different leaf nodes is told from two arrays as well
  A2 = sp.Val(val=A.val)
  np.testing.assert_array_equal((A * A2 + 1).optimized().glom(), a * a + 1)
  assert plan.stats['misses'] == 3


def test_scalars_are_keyed_by_value_and_driver_arrays_by_type(ctx):
  a = _arr(5)
  A = sp.from_numpy(a)
  for s in (2.0, 3.0, 2.0):
    np.testing.assert_array_equal((A * s + 1).optimized().glom(), a * np.float32(s) + 1)
  assert plan.stats['misses'] == 2 and plan.stats['hits'] == 1
  w = np.arange(8, dtype=np.float32).reshape(8, 1) - 3
  for step in range(3):
    got = sp.dot(A, w).optimized().glom()          # a NEW driver array every step, as a gradient loop hands over
    np.testing.assert_array_equal(got, a.dot(w))
    w = w + 1
  w -= 5                                           # ... and one updated in place
  np.testing.assert_array_equal(sp.dot(A, w).optimized().glom(), a.dot(w))
  assert plan.stats['hits'] >= 3


def test_reductions_and_the_lreg_step_through_plans(ctx):
  from spartan_amd.examples import lreg
  x, y = _arr(6, (40, 8)), _arr(7, (40, 1))
  X, Y = sp.from_numpy(x), sp.from_numpy(y)
  w = np.ones((8, 1), np.float32)
  for _ in range(4):
    g = lreg.gradient(X, Y, w).optimized().glom()
    np.testing.assert_allclose(g, (x * (x.dot(w) - y)).sum(0), rtol=1e-5)     # (three tiles: another summation tree)
    w = w - g.reshape(8, 1) * np.float32(0.001)
  assert plan.stats['misses'] == 1 and plan.stats['hits'] == 3
  for axis in (None, 0, 1, 0):
    np.testing.assert_array_equal(sp.sum(X * 2, axis).optimized().glom(), (x * 2).sum(axis))
  assert plan.stats['misses'] == 4     # axis is part of the structure


def test_a_node_and_its_optimised_twin_share_their_value(ctx):
  a = _arr(8)
  A = sp.from_numpy(a)
  (A * 3 - 1).optimized().force()                  # records the plan
  e = A * 3 - 1
  o = e.optimized()
  assert plan.stats['hits'] == 1 and o.expr_id == e.expr_id
  o.force()
  assert e.cache() is not None                     # found under the id the instantiated root took over
  np.testing.assert_array_equal(e.glom(), a * 3 - 1)


def test_dags_with_values_and_random_builders_are_optimised_the_long_way(ctx):
  a = _arr(9)
  A = sp.from_numpy(a)
  inner = A * 2
  inner.force()                                    # the collapse pass would cut the DAG at this node
  before = dict(plan.stats)
  np.testing.assert_array_equal((inner + 1).optimized().glom(), a * 2 + 1)
  assert plan.stats['unplannable'] == before['unplannable'] + 1 and plan.stats['misses'] == before['misses']
  r1 = (sp.rand(6, 5) + 1).optimized().glom()
  r2 = (sp.rand(6, 5) + 1).optimized().glom()
  assert not np.array_equal(r1, r2)                # never answered from a recording
  assert plan.stats['hits'] == before['hits']


def test_plans_keep_no_array_alive(ctx):
  a = _arr(10)
  A = sp.from_numpy(a)
  ref = weakref.ref(A.val)
  np.testing.assert_array_equal((A * A - 2).optimized().glom(), a * a - 2)
  assert len(plan._plans) == 1
  del A
  gc.collect()
  assert ref() is None, 'the recorded plan holds the array it was recorded on'


def test_flags_and_worker_count_are_part_of_the_key(ctx):
  a = _arr(11)
  A = sp.from_numpy(a)
  (A + 1 + 1).optimized().force()
  optimize.FLAGS['opt_map_fusion'] = False
  try:
    e = (A + 1 + 1).optimized()
    assert plan.stats['misses'] == 2               # not the fused recording
    assert len(e.children) == 2 and type(e.children[0]).__name__ == 'MapExpr'
    np.testing.assert_array_equal(e.glom(), a + 2)
  finally:
    optimize.FLAGS['opt_map_fusion'] = True
  optimize.FLAGS['opt_plan_cache'] = False
  try:
    before = dict(plan.stats)
    (A + 1 + 1).optimized().force()
    assert plan.stats == before
  finally:
    optimize.FLAGS['opt_plan_cache'] = True


def test_kmeans_iterations_through_plans(ctx):
  from spartan_amd.examples.sklearn.cluster import KMeans
  from scipy.spatial.distance import cdist
  x = np.random.RandomState(12).rand(60, 4).astype(np.float32)
  X = sp.from_numpy(x)
  c = np.random.RandomState(13).rand(5, 4)
  for _ in range(4):       # (the start centers are float64, the updated ones float32: two structures)
    lab = np.argmin(cdist(x, c), axis=1)
    cnt = np.bincount(lab, minlength=5)
    want = np.stack([x[lab == i].sum(axis=0) for i in range(5)]) / np.maximum(cnt, 1)[:, None]
    c_new, _ = KMeans(5, 1).fit(X, c, implementation='map2', reducer=np.add)
    if cnt.min() > 0:
      np.testing.assert_allclose(c_new, want, rtol=1e-6)
    c = c_new
  # (rounds 4-5 asserted plan-table hits here: the driver optimised its count / sum joins every iteration.  It now
  # evaluates them as built -- a join has nothing to fuse, and the optimiser's auto-tiling pass cut the one-tile
  # targets by rows again -- so the iterations no longer visit the plan table at all)
  km = KMeans(5, 1)
  lab = sp.map2(X, 0, fn=__import__('spartan_amd.examples.sklearn.cluster.k_means_', fromlist=['x']).kmeans_map2_dist_mapper,
                fn_kw={'centers': c}, shape=(60,))
  counts_arr, sums_arr = km._launch_join(X, lab, np.add)
  assert len(counts_arr.tiles) == 1 and len(sums_arr.tiles) == 1


# ---- kernels lowered by the optimiser (optimize._prelower -> HipBackend.prelower_map) -----------------------------
@pytest.fixture
def hip_ctx():
  c = sp.initialize('hip', num_workers=1)
  plan.clear()
  yield c
  sp.shutdown()


@pytest.mark.gpu
def test_first_evaluation_finds_the_program_the_optimiser_lowered(hip_ctx):
  be = hip_ctx.backend
  a = _arr(21, (64, 48))
  A = sp.from_numpy(a)
  programs, hits = len(be._lowered), be.lowering_hits
  e = (((A * A + A) * 0.375 - A) / (A + 2.5)).optimized()
  assert len(be._lowered) == programs + 1          # lowered, not run: no launch was counted
  launches = be.launches
  got = e.glom()
  assert be.lowering_hits == hits + 1 and be.launches == launches + 1
  np.testing.assert_array_equal(got, ((a * a + a) * np.float32(0.375) - a) / (a + np.float32(2.5)))
  # ragged tilings: one program per distinct tile shape
  sp.shutdown()
  c = sp.initialize('hip', num_workers=4)
  plan.clear()
  B = sp.from_numpy(_arr(22, (50, 7)))
  programs = len(c.backend._lowered)
  f = (B * 3.0 - B * B).optimized()
  assert len(c.backend._lowered) > programs
  hits = c.backend.lowering_hits
  np.testing.assert_array_equal(f.glom(), _arr(22, (50, 7)) * np.float32(3) - _arr(22, (50, 7)) ** 2)
  assert c.backend.lowering_hits > hits


@pytest.mark.gpu
def test_prelowering_can_be_switched_off_and_skips_what_cannot_be_replayed(hip_ctx):
  be = hip_ctx.backend
  A = sp.from_numpy(_arr(23, (64, 48)))
  optimize.FLAGS['opt_prelower'] = False
  try:
    programs = len(be._lowered)
    e = (A * 1.5 + A * A * A).optimized()
    assert len(be._lowered) == programs
    np.testing.assert_array_equal(e.glom(), _arr(23, (64, 48)) * np.float32(1.5) + _arr(23, (64, 48)) ** 3)
  finally:
    optimize.FLAGS['opt_prelower'] = True
  # a driver-side NumPy operand is uploaded at evaluation time: left to evaluate_map, same values
  w = np.arange(48, dtype=np.float32)
  programs = len(be._lowered)
  g = (A * w - 2.0).optimized()
  assert len(be._lowered) == programs
  np.testing.assert_array_equal(g.glom(), _arr(23, (64, 48)) * w - np.float32(2))
  # inputs that are expressions themselves: the inner map is prepared, the outer one when its input exists
  h = sp.sum(A * A + 1.0, axis=0).optimized()
  np.testing.assert_allclose(h.glom(), (_arr(23, (64, 48)) ** 2 + 1).sum(axis=0), rtol=1e-6)


def test_an_array_inside_fn_kw_is_never_answered_from_a_plan(ctx):
  """A map whose keywords hold an array (fn_kw={'w': array}): the operator tree of a plan is shared between its
  instances, so the array cannot be swapped per instance -- every such DAG is optimised the long way and computes
  with ITS array, plain and fused into a larger tree."""
  a = _arr(7, (6, 8))
  A = sp.from_numpy(a)

  def f(x, w):
    return x + w
  for i in range(3):
    w = np.full((1, 8), i, np.float32)
    np.testing.assert_array_equal(sp.map(A, fn=f, fn_kw={'w': w}).optimized().glom(), a + i)
    np.testing.assert_array_equal(((sp.map(A, fn=f, fn_kw={'w': w}) * 2) + A).optimized().glom(), (a + i) * 2 + a)
  assert plan.stats['hits'] == 0 and plan.stats['unplannable'] == 6, plan.stats


def test_signed_zero_and_nan_scalars_are_keyed_by_their_bits(ctx):
  a = np.array([[1.0, -2.0, 0.0]], np.float32)
  A = sp.from_numpy(a)
  pos = (A * 0.0).optimized().glom()
  neg = (A * -0.0).optimized().glom()
  np.testing.assert_array_equal(np.signbit(pos), np.signbit(a * np.float32(0.0)))
  np.testing.assert_array_equal(np.signbit(neg), np.signbit(a * np.float32(-0.0)))
  assert plan.stats['misses'] == 2 and plan.stats['hits'] == 0
  for _ in range(2):
    assert np.isnan((A + float('nan')).optimized().glom()).all()
  assert plan.stats['misses'] == 3 and plan.stats['hits'] == 1          # a NaN equals itself in the key


def test_results_die_with_their_last_reference_without_the_cyclic_collector(ctx):
  """A result is a multi-GiB set of tiles: it must go back to the tile store when the driver drops it, not when the
  cyclic collector next runs (a loop that rebuilds its expression would meanwhile allocate fresh tiles every step --
  160 ms per 2 GiB hipMalloc).  Checked with the collector OFF, first evaluation (plan miss) and later ones (hits)."""
  X = sp.from_numpy(_arr(11, (8, 8))).force()
  Xv = sp.Val(val=X)
  programs = {
      'fused map': lambda: (Xv * Xv + Xv).optimized().force(),
      'plain map': lambda: (Xv + 1).force(),
      'chain': lambda: (((Xv * Xv + Xv) * 0.5 - Xv) / (Xv + 2.0)).optimized().force(),
      'sum': lambda: sp.sum(Xv, 0).force(),
      'map -> sum': lambda: sp.sum((Xv - 0.5) * (Xv - 0.5), axis=0).optimized().force(),
      'argmax': lambda: sp.argmax(Xv, 1).force(),
      'dot': lambda: sp.dot(Xv, Xv).optimized().force(),
      'dot + driver array': lambda: sp.dot(Xv, np.ones((8, 1), np.float32)).optimized().force(),
  }
  base = importlib.import_module('spartan_amd.expr.base')
  gc.collect()
  gc.disable()
  try:
    for name, build in programs.items():
      for i in range(3):
        r = build()
        w = weakref.ref(r)
        del r
        assert w() is None, '%s, evaluation %d: the result is still alive' % (name, i)
    assert not base.eval_cache._values, sorted(base.eval_cache._values)
  finally:
    gc.enable()


def test_the_one_pass_description_of_operator_maps_agrees_with_the_general_walk(ctx):
  """_Walk.map_node describes the maps operators build in one pass: it must enter the same leaves and node ids in the
  same order as the general walk (a plan's slots and ids are positions in those lists), and tell two DAGs apart
  exactly when the general walk does."""
  a, b = _arr(11), _arr(12)
  A, B = sp.from_numpy(a), sp.from_numpy(b)
  w = np.arange(8, dtype=np.float32)
  builders = [lambda: (A * A + A) * 0.5 - A, lambda: (B * B + B) * 0.5 - B, lambda: (A * B + A) * 0.5 - A,
              lambda: (A * A + A) * 0.25 - A, lambda: A * A + 1, lambda: A * B + 1, lambda: B * A + 1,
              lambda: -A + w, lambda: -B + (w + 1), lambda: sp.sum(A * A + B, axis=0), lambda: sp.sum(B * B + A, axis=0),
              lambda: sp.sqrt(sp.abs(A)) * B, lambda: sp.maximum(A, 0) + B, lambda: sp.maximum(B, 0) + A, lambda: sp.minimum(A, B)]
  flags = tuple(optimize.FLAGS.values())

  def both(build):
    dag = build()
    plan.signature(dag, flags)                   # (binds the node type on first use)
    fast = plan.signature(dag, flags)
    kept = plan._MapExpr[0]
    plan._MapExpr[0] = None.__class__            # no node has this type: every MapExpr goes the general way
    try:
      general = plan.signature(dag, flags)
    finally:
      plan._MapExpr[0] = kept
    assert fast is not None and general is not None
    assert [id(v) for v in fast[1]] == [id(v) for v in general[1]] and fast[2] == general[2]
    assert 'M!' in repr(fast[0]) and 'M!' not in repr(general[0])
    return dag, fast[0], general[0]
  described = [both(bd) for bd in builders]      # (the DAGs stay alive: their ids are parts of nothing, but fns' are)
  for i, (_, fi, gi) in enumerate(described):
    for j, (_, fj, gj) in enumerate(described):
      assert (fi == fj) == (gi == gj), (i, j)
  assert described[0][1] == described[1][1] and described[0][1] != described[2][1] != described[3][1]
