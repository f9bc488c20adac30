"""Auto-tiling: the C++ solver behind sp_tiling_solve against brute force, and the pass
(spartan_amd/expr/tiling.py, reference optimize.py:459-1054) on expression DAGs: values unchanged, tilings chosen so
that reductions / joins need no redistribution, modelled xGMI bytes never above those of the default tiling."""
import itertools

import numpy as np
import pytest

import spartan_amd as sp
import importlib

opt = importlib.import_module('spartan_amd.expr.optimize')   # (`spartan_amd.expr.optimize` the ATTRIBUTE is the function)
tiling = importlib.import_module('spartan_amd.expr.tiling')


def _brute(n_nodes, edges, groups):
  best = None
  grouped = {n for g in groups for n in g}
  for pick in itertools.product(*[range(len(g)) for g in groups]):
    chosen = set(range(n_nodes)) - grouped
    chosen |= {g[s] for g, s in zip(groups, pick)}
    cost = sum(c for u, v, c in edges if u in chosen and v in chosen)
    if best is None or cost < best[0] - 1e-9:
      best = (cost, pick)
  return best


def test_solver_matches_brute_force():
  rng = np.random.RandomState(0)
  for case in range(60):
    n_groups = rng.randint(0, 6)
    groups, n = [], 0
    for _ in range(n_groups):
      k = rng.randint(1, 5)
      groups.append(list(range(n, n + k)))
      n += k
    n += rng.randint(0, 4)                       # ungrouped nodes
    if n == 0:
      continue
    edges = []
    for _ in range(rng.randint(0, 40)):
      u, v = rng.randint(0, n, size=2)
      edges.append((int(u), int(v), float(rng.randint(0, 100))))
    choice, total = tiling.solve(n, edges, groups)
    want = _brute(n, edges, groups)
    assert abs(total - want[0]) < 1e-6, (case, total, want)
    chosen = (set(range(n)) - {m for g in groups for m in g}) | {g[s] for g, s in zip(groups, choice)}
    assert abs(sum(c for u, v, c in edges if u in chosen and v in chosen) - total) < 1e-6


def test_solver_large_problem_is_a_local_optimum():
  rng = np.random.RandomState(1)
  groups = [list(range(4 * i, 4 * i + 4)) for i in range(60)]
  edges = [(int(u), int(v), float(c)) for u, v, c in
           zip(rng.randint(0, 240, 2000), rng.randint(0, 240, 2000), rng.randint(1, 1000, 2000))]
  choice, total = tiling.solve(240, edges, groups)

  def cost(ch):
    chosen = {g[s] for g, s in zip(groups, ch)}
    return sum(c for u, v, c in edges if u in chosen and v in chosen)
  assert abs(cost(choice) - total) < 1e-6
  for g in range(len(groups)):                   # no single-group move improves it
    for s in range(4):
      alt = list(choice)
      alt[g] = s
      assert cost(alt) >= total - 1e-6


@pytest.fixture
def ctx4():
  from oracle.np_backend import NumpyBackend
  c = sp.initialize(backend=NumpyBackend(), num_workers=4)
  opt.FLAGS['opt_auto_tiling'] = True
  yield c
  opt.FLAGS['opt_auto_tiling'] = True   # (the default)
  sp.shutdown()


def _tiles(arr):
  return sorted((ex.ul, ex.lr) for ex in arr.tiles)


def test_reduction_axis_decides_the_tiling_of_a_new_array(ctx4):
  # sum over axis 0 is free when the array is split by COLUMNS, sum over axis 1 when split by ROWS
  # (optimize.py:620-629); the creation node is free in every tiling, so the reduce decides.
  at = tiling.AutomaticTiling()
  e = sp.sum(sp.ones((64, 48)) * 2, axis=0)
  at.visit(e)
  nd = [x for x in _walk(e) if x.typename() == 'NdArrayExpr'][0]
  assert nd.tile_hint == (64, 12) and at.report['link_bytes'] == 0
  np.testing.assert_array_equal(e.optimized().glom(), np.full(48, 128, np.float32))
  at = tiling.AutomaticTiling()
  e = sp.sum(sp.ones((64, 48)) * 2, axis=1)
  at.visit(e)
  nd = [x for x in _walk(e) if x.typename() == 'NdArrayExpr'][0]
  assert nd.tile_hint == (16, 48) and at.report['link_bytes'] == 0
  np.testing.assert_array_equal(e.optimized().glom(), np.full(64, 96, np.float32))


def _walk(e, seen=None):
  seen = seen if seen is not None else set()
  if not isinstance(e, sp.Expr) or id(e) in seen:
    return
  seen.add(id(e))
  yield e
  for v in e.dependencies().values():
    if isinstance(v, sp.Expr):
      for x in _walk(v, seen):
        yield x
    elif isinstance(v, (list, tuple)):
      for w in v:
        for x in _walk(w, seen):
          yield x
  vals = getattr(e, 'vals', None)
  if isinstance(vals, (list, tuple)):
    for w in vals:
      for x in _walk(w, seen):
        yield x


def test_existing_arrays_keep_their_tiling_and_values_do_not_change(ctx4):
  rng = np.random.RandomState(2)
  a = rng.randint(-3, 4, size=(40, 24)).astype(np.float32)
  b = rng.randint(-3, 4, size=(24, 32)).astype(np.float32)
  A, B = sp.from_numpy(a), sp.from_numpy(b)
  progs = {
      'dot': lambda: sp.dot(A, B),
      'dot_then_sum': lambda: sp.sum(sp.dot(A, B) + 1, axis=0),
      'map_two_inputs': lambda: A * 2 + sp.ones((40, 24)),
      'transpose_dot': lambda: sp.dot(sp.transpose(A), sp.from_numpy(a)),
      'tall_dot': lambda: sp.dot(sp.from_numpy(np.tile(a, (4, 1))), B),
  }
  want = {'dot': a @ b, 'dot_then_sum': (a @ b + 1).sum(0), 'map_two_inputs': a * 2 + 1,
          'transpose_dot': a.T @ a, 'tall_dot': np.tile(a, (4, 1)) @ b}
  for name, build in progs.items():
    got = build().optimized().glom()
    np.testing.assert_array_equal(got, want[name], err_msg=name)
  assert _tiles(A.val) == _tiles(sp.from_numpy(a).val)          # inputs are not re-tiled


def test_modelled_bytes_never_exceed_the_default_tiling(ctx4):
  """The chosen assignment costs at most what "every new array by rows" costs in the same model."""
  a = sp.from_numpy(np.ones((64, 64), np.float32), tile_hint=(64, 16))        # a COLUMN-tiled input
  e = sp.sum(sp.ones((64, 64)) + a, axis=0)
  at = tiling.AutomaticTiling()
  at.visit(e)
  edges = [(u, v, c) for (u, v), c in at.edges.items()]
  rows_only = set(range(len(at.nodes))) - {n for g in at.groups for n in g} | {g[0] for g in at.groups}
  default_cost = sum(c for u, v, c in edges if u in rows_only and v in rows_only)
  assert at.report['link_bytes'] <= default_cost
  assert at.report['link_bytes'] == 0 and default_cost > 0      # columns everywhere: nothing moves
  nd = [x for x in _walk(e) if x.typename() == 'NdArrayExpr'][0]
  assert nd.tile_hint == (64, 16)
  np.testing.assert_array_equal(e.optimized().glom(), np.full(64, 128, np.float32))


def test_pass_is_on_by_default_and_a_no_op_on_one_worker():
  from oracle.np_backend import NumpyBackend
  assert opt.FLAGS['opt_auto_tiling'] is True          # as in the reference (optimize.py:1094)
  sp.initialize(backend=NumpyBackend(), num_workers=1)
  try:
    e = sp.sum(sp.ones((8, 8)), axis=0)
    at = tiling.AutomaticTiling()
    assert at.visit(e) is e and at.report == {}
  finally:
    sp.shutdown()


@pytest.mark.gpu
def test_auto_tiled_programs_on_the_hip_backend():
  """The pass only changes tile hints; the HIP tile path must give the same values under the tilings it picks
  (column tiles, blocks) as under the default row tiles."""
  sp.initialize('hip', num_workers=4)
  opt.FLAGS['opt_auto_tiling'] = True
  try:
    rng = np.random.RandomState(3)
    a = rng.randint(-3, 4, size=(96, 64)).astype(np.float32)
    b = rng.randint(-3, 4, size=(64, 80)).astype(np.float32)
    A, B = sp.from_numpy(a), sp.from_numpy(b)
    Ac = sp.from_numpy(a, tile_hint=(96, 16))
    np.testing.assert_array_equal(sp.dot(A, B).optimized().glom(), a @ b)
    np.testing.assert_array_equal(sp.sum(sp.dot(A, B) + 1, axis=0).optimized().glom(), (a @ b + 1).sum(0))
    np.testing.assert_array_equal(sp.sum(sp.ones((96, 64)) * 2 + Ac, axis=0).optimized().glom(), (a + 2).sum(0))
    np.testing.assert_array_equal(sp.sum(sp.ones((96, 64)) * 2 + Ac, axis=1).optimized().glom(), (a + 2).sum(1))
    np.testing.assert_array_equal(sp.dot(sp.transpose(A), sp.from_numpy(a)).optimized().glom(), a.T @ a)
    e = sp.sum(sp.ones((96, 64)) + Ac, axis=0)
    at = tiling.AutomaticTiling()
    at.visit(e)
    assert at.report['link_bytes'] == 0
  finally:
    opt.FLAGS['opt_auto_tiling'] = True   # (the default)
    sp.shutdown()


def _random_programs(backend_factory, seeds, n_dots):
  """tests/test_random_autotiling.py in spirit: the random expression DAGs of tests/test_fuzz_gpu.py (maps, views,
  reductions, argmax, means) and of its dot generator evaluate to the same values with the pass on and off
  (reductions to a summation-order tolerance: another tiling is another summation tree)."""
  from tests import test_fuzz_gpu as F

  def run(flag):
    out = {}
    sp.initialize(backend=backend_factory(), num_workers=4)
    opt.FLAGS['opt_auto_tiling'] = flag
    try:
      for s in seeds:
        try:
          with np.errstate(all='ignore'):
            out[s] = F._program(s, sp)
        except Exception as e:   # noqa: BLE001
          out[s] = type(e).__name__
      for s in range(n_dots):
        try:
          out['dot%d' % s] = F._dot_case(s, sp)
        except Exception as e:   # noqa: BLE001
          out['dot%d' % s] = type(e).__name__
    finally:
      opt.FLAGS['opt_auto_tiling'] = True   # (the default)
      sp.shutdown()
    return out
  off, on = run(False), run(True)
  bad = []
  for k in off:
    a, b = off[k], on[k]
    if isinstance(a, str) or isinstance(b, str):
      if a != b and not isinstance(a, str):
        bad.append((k, 'raised %s only with the pass on' % b))
      continue
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape or a.dtype != b.dtype:
      bad.append((k, 'shape/dtype'))
    elif a.dtype.kind in 'iub':
      if not np.array_equal(a, b):
        bad.append((k, 'integer result differs'))
    elif not np.allclose(a, b, rtol=1e-5, atol=1e-6, equal_nan=True):
      bad.append((k, 'float result differs'))
  assert not bad, bad[:10]


def test_random_programs_do_not_change_under_auto_tiling():
  from oracle.np_backend import NumpyBackend
  _random_programs(NumpyBackend, range(4000, 4120), 20)


@pytest.mark.gpu
def test_random_programs_do_not_change_under_auto_tiling_gpu():
  """The same on the HIP backend: column- and block-tiled arrays chosen by the pass go through the strided kernels."""
  from spartan_amd.backend_hip import HipBackend
  _random_programs(HipBackend, range(4000, 4200), 40)


def test_shuffle_cost_hint_steers_the_target_tiling():
  """A shuffle's user cost hint (reference shuffle.py:99-135 / optimize.py:701-707): elements of its target that
  move for each (target tiling, shuffle tiling) pair.  With "free when the target is column-tiled" the pass creates
  the target column-tiled; with "free when row-tiled" row-tiled; without a hint every choice costs the same."""
  import importlib
  import spartan_amd as sp
  from oracle.np_backend import NumpyBackend
  from spartan_amd.expr import tiling
  sp.initialize(backend=NumpyBackend(), num_workers=4)
  try:
    src = sp.from_numpy(np.ones((64, 64), np.float32)).force()          # row-tiled source

    def fn(array, ex, out):
      return []
    for free_key, want in (('10', tiling.COL), ('00', tiling.ROW)):
      target = sp.ndarray((64, 64), dtype=np.float32)
      hint = {k: 64 * 64 for k in ('00', '01', '10', '11')}
      hint[free_key] = 0
      e = sp.shuffle(sp.Val(val=src), fn, kw={'out': target}, target=None, cost_hint={hash(target): hint})
      pas = tiling.AutomaticTiling()
      pas.visit(e)
      assert pas.report['tilings'][target.expr_id] == want, (free_key, pas.report)
      assert pas.report['link_bytes'] == 0
  finally:
    sp.shutdown()


# ---- the solver against the REFERENCE's solver (tests/golden/tiling_golden.json: make_golden.py --tiling) ----------
def _reference_cost(t, edges, groups, chosen):
  """tiling.cc:172-206 calc_cost: the objective of the reference's best_tiling -- edge costs along everything
  reachable from node 0 through chosen nodes; negative (infeasible) when a chosen node has an edge into a group
  but none to the group's chosen member, or into an ungrouped node that is not chosen."""
  out, member = {}, {}
  for u, v, c in edges:
    out.setdefault(u, []).append((v, c))
  for g in groups:
    for n in g:
      member[n] = g
  chosen = set(chosen)
  visited = set()

  def walk(s):
    if s == t or s in visited:
      return 0
    cost = 0
    for v, c in out.get(s, ()):
      if v in chosen:
        below = walk(v)
        if below < 0:
          cost = below
          break
        cost += below + c
      elif v in member:
        pick = [m for m in member[v] if m in chosen][0]
        if not any(w == pick for w, _ in out.get(s, ())):
          cost = -1
          break
      else:
        cost = -1
        break
    visited.add(s)
    return cost
  return walk(0)


def _for_our_solver(t, edges, groups):
  """The reference's graph in sp_tiling_solve's terms: a node with an edge into a group but none to one of its
  members cannot be chosen together with that member (the reference prices the gap as infinite, tiling.cc:103)."""
  have = {(u, v) for u, v, _ in edges}
  member = {n: g for g in groups for n in g}
  extra = []
  for u in sorted({u for u, _, _ in edges}):
    for g in groups:
      if any((u, m) in have for m in g):
        extra += [(u, m, 1e30) for m in g if (u, m) not in have]
  return t + 1, [(u, v, float(c)) for u, v, c in edges] + extra, [list(g) for g in groups]


@pytest.mark.host_logic
def test_solver_matches_the_reference_solver_on_its_own_graphs():
  """Every cost graph the reference's AutomaticTiling pass built for the shared test programs (4 and 8 workers),
  as recorded on its way into the reference's tiling.cc: (i) the recorded choice of its exhaustive `best_tiling`
  is an optimum of its own objective (pins the restatement of that objective used here); (ii) sp_tiling_solve finds
  a choice of the SAME cost on every graph; (iii) it is never worse than the reference's default heuristic
  (`mincost_tiling`), whose recorded choices are feasible but not always optimal."""
  import json
  import os
  gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'tiling_golden.json')))
  graphs = 0
  heuristic_worse = 0
  for rec in gold:
    by_alg = {}
    for alg in ('best', 'mincost'):
      for call in rec.get(alg, {}).get('calls', []):
        by_alg.setdefault(json.dumps([call['t'], call['edges'], call['groups']]), {})[alg] = call
    for key, calls in by_alg.items():
      t, edges, groups = json.loads(key)
      grouped = {n for g in groups for n in g}
      fixed = set(range(t + 1)) - grouped

      def cost_of(chosen_members):
        return _reference_cost(t, edges, groups, fixed | set(chosen_members))
      optimum = None
      if len(groups) <= 7:
        for pick in itertools.product(*groups):
          c = cost_of(pick)
          if c >= 0 and (optimum is None or c < optimum):
            optimum = c
      if 'best' in calls:
        assert optimum is not None
        assert cost_of([n for n in calls['best']['chosen'] if n in grouped]) == optimum, rec['program']
      n_nodes, our_edges, our_groups = _for_our_solver(t, edges, groups)
      choice, total = tiling.solve(n_nodes, our_edges, our_groups)
      ours = cost_of([g[s] for g, s in zip(our_groups, choice)])
      assert ours >= 0, (rec['program'], 'infeasible choice')
      if optimum is not None:
        assert ours == optimum, (rec['program'], rec['workers'], ours, optimum)
      if 'mincost' in calls:
        theirs = cost_of([n for n in calls['mincost']['chosen'] if n in grouped])
        assert theirs < 0 or ours <= theirs, (rec['program'], ours, theirs)
        heuristic_worse += (theirs < 0 or theirs > ours)
      graphs += 1
  assert graphs >= 100, graphs
