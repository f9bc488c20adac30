"""Body of one rank of the world_size-2 gloo tests (launched by test_multiprocess.py).
Runs the shared Spartan programs with one/two workers per rank and checks every
result on every rank; prints 'RANK r OK n' on success."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import spartan_amd as sp  # noqa: E402
from oracle.np_backend import NumpyBackend  # noqa: E402
from tests import programs  # noqa: E402


def run_examples(total_workers):
  from tests import test_examples as T
  hip = sp.get_context().backend.name == 'hip'
  count = 0
  if total_workers in T.GOLD:
    for impl in ('map2', 'outer', 'shuffle'):
      for tag in ('', '_empty'):
        T._check_kmeans(total_workers, impl, tag, exact_centers=True)
        count += 1
    T._check_regressions(total_workers, rtol=2e-6 if hip else 0)
    count += 1
  # true k-means update (partials combined across ranks by the reducer)
  x, init = T.INPUTS['km_x'], T.INPUTS['km_init'].copy()
  centers, labels = T.KMeans(5, 1).fit(sp.from_numpy(x), init.copy(), implementation='map2', reducer=np.add)
  from scipy.spatial.distance import cdist
  lab = np.argmin(cdist(x, init), axis=1)
  want = np.stack([x[lab == i].sum(axis=0) for i in range(5)]) / np.bincount(lab, minlength=5)[:, None]
  np.testing.assert_array_equal(T._val(labels), lab.astype(np.float32))
  np.testing.assert_allclose(centers, want, rtol=1e-6)
  # What the driver draws at random is replicated state: with DIFFERENT np.random streams on the ranks the
  # start centers / start weights must still be the same everywhere (rank 0 draws, the others receive).
  from spartan_amd.examples import lreg
  world = sp.get_context().world
  np.random.seed(100 + world.rank)
  c2, _ = T.KMeans(4, 2).fit(sp.from_numpy(x), None, implementation='map2', reducer=np.add)
  w2 = lreg.fit(sp.from_numpy(T.INPUTS['reg_x']), sp.from_numpy(T.INPUTS['reg_y']), 2)
  for mine, name in ((c2, 'centers'), (w2, 'weights')):
    copies = world.all_gather_object(np.asarray(mine))
    for other in copies[1:]:
      np.testing.assert_array_equal(copies[0], other, err_msg='ranks diverged: ' + name)
  sp.set_random_seed(None)                 # clock seed: taken on rank 0
  draws = world.all_gather_object(np.random.rand(3))
  np.testing.assert_array_equal(draws[0], draws[-1])
  return count + 2


def run_sparse(total_workers):
  """Sparse arrays across two ranks: tiles are built and used by the rank that owns them; blocks that cross ranks
  (sparse x sparse joins, row-tiled shuffles, fetches of remote regions) travel as their three arrays in grouped
  exchanges (distarray._ship_sparse) -- only glom() of a sparse result gathers host objects."""
  import json
  from tests import sparse_programs as SP
  here = os.path.join(ROOT, 'tests', 'golden')
  meta = json.load(open(os.path.join(here, 'sparse_meta.json')))
  gold1 = np.load(os.path.join(here, 'sparse_w1.npz'))
  count = 0
  from spartan_amd import context
  shipped_before = context.get().world.stats['sparse_blocks']
  for name, build, tol in SP.programs():
    res = build(sp)
    res = res.force() if hasattr(res, 'force') else res
    got, was_sparse = SP.to_dense(res.glom())
    want, m = gold1[name], meta['1'][name]           # the programs' values do not depend on the tiling
    assert was_sparse == m['sparse'] and got.dtype.str == m['dtype'], (name, was_sparse, got.dtype)
    if tol is None:
      np.testing.assert_array_equal(got, want, err_msg=name)
    else:
      np.testing.assert_allclose(got, want, rtol=tol[0], atol=tol[1], err_msg=name)
    count += 1
  # blocks did cross ranks as device arrays (every rank owns tiles other ranks read)
  assert context.get().world.stats['sparse_blocks'] > shipped_before
  return count


def run_auto_tiling(world):
  """Auto-tiling across ranks: same values, and the bytes that actually cross ranks do not go up (here they go to
  what the model predicts: a column-tiled operand summed over axis 0 needs no exchange when the new array is created
  column-tiled too)."""
  import importlib
  opt = importlib.import_module('spartan_amd.expr.optimize')
  a = (np.arange(64 * 64, dtype=np.float32).reshape(64, 64) % 7)

  def moved(flag):
    opt.FLAGS['opt_auto_tiling'] = flag
    try:
      A = sp.from_numpy(a, tile_hint=(64, 64 // sp.get_context().num_workers))     # column tiles
      before = dict(world.stats)
      got = sp.sum(sp.ones((64, 64)) * 2 + A, axis=1).optimized().glom()
      np.testing.assert_array_equal(got, (a + 2).sum(1))
      got0 = sp.sum(sp.ones((64, 64)) * 2 + A, axis=0).optimized().force()
      after = dict(world.stats)
      np.testing.assert_array_equal(got0.glom(), (a + 2).sum(0))
      return (after['collective_bytes'] + after['p2p_bytes']) - (before['collective_bytes'] + before['p2p_bytes'])
    finally:
      opt.FLAGS['opt_auto_tiling'] = True   # (the default)
  plain, tiled = moved(False), moved(True)
  assert tiled <= plain, (plain, tiled)
  return 1


def run_masked_fetch(world):
  """A region with never-written cells read ACROSS ranks (distarray.py:355-365 + tile.pyx:100-113): glom gives the
  reference's MaskedArray on every rank; once every cell is written the same read is plain again."""
  from spartan_amd.array import extent
  ctx = sp.get_context()
  be = ctx.backend
  a = sp.ndarray((8, 6), dtype=np.float32, tile_hint=(2, 6)).evaluate()        # 4 row tiles over the ranks
  assert a.written == set()
  row = np.arange(6, dtype=np.float32).reshape(1, 6)
  a.update(extent.create((3, 0), (4, 6), (8, 6)), be.from_numpy(row))            # half of the second tile
  a.update(extent.create((4, 0), (6, 6), (8, 6)), be.from_numpy(np.ones((2, 6), np.float32)))   # all of the third
  assert a.written == {extent.create((4, 0), (6, 6), (8, 6))}
  got = a.glom()
  assert isinstance(got, np.ma.MaskedArray)
  want_mask = np.ones((8, 6), bool)
  want_mask[3:6] = False
  np.testing.assert_array_equal(np.ma.getmaskarray(got), want_mask)
  np.testing.assert_array_equal(got[3].filled(-1), row[0])
  np.testing.assert_array_equal(got[4:6].filled(-1), np.ones((2, 6), np.float32))
  part = a.fetch(extent.create((4, 1), (6, 3), (8, 6)))                          # inside the written tile: plain
  np.testing.assert_array_equal(be.to_numpy(part), np.ones((2, 2), np.float32))
  a.update(extent.create((0, 0), (8, 6), (8, 6)), be.from_numpy(np.full((8, 6), 2, np.float32)))
  assert a.written is None
  np.testing.assert_array_equal(a.glom(), np.full((8, 6), 2, np.float32))
  return 1


def run_heartbeat(world):
  """Two ranks: rank 1 stops reporting; BOTH ranks mark its workers failed at the same safe point (the watchers
  agree through the control plane), and a cached value is recomputed on the survivors' tiles and rank 1's."""
  import time
  ctx = sp.get_context()
  hb = ctx.start_heartbeat(interval=0.05, threshold=4, probe=lambda: True)
  a = np.ones((16, 4), np.float32)
  e = sp.ones((16, 4)) + 3             # (a value with lineage: it can be computed again)
  first = e.evaluate()
  world.barrier()
  if world.rank == 1:
    hb.pause()
  deadline = time.time() + 10
  while 1 not in hb.failed_ranks and world.rank == 0 and time.time() < deadline:
    time.sleep(0.02)
  world.barrier()                      # rank 0 has seen it; rank 1 may or may not have: the union decides
  second = (e * 1).evaluate()
  mine = sorted(w for w in range(ctx.num_workers) if ctx.rank_of(w) == 1)
  assert sorted(ctx.failed_workers) == mine, (ctx.failed_workers, mine)
  assert len(first.bad_tiles) == sum(1 for t in first.tiles.values() if ctx.rank_of(t.worker) == 1)
  np.testing.assert_array_equal(second.glom(), a + 3)
  hb.stop()
  ctx.heartbeat = None
  return 1


def run_sort(workers):
  """sort / argsort along an axis and the sample sort (axis=None) with tiles on both ranks."""
  rng = np.random.RandomState(5)
  x = (rng.randn(64, 40) * 30).astype(np.float32)
  x.flat[::9] = x.flat[0]
  a = sp.from_numpy(x)
  for axis in (0, 1):
    np.testing.assert_array_equal(sp.sort(a, axis).glom(), np.sort(x, axis, kind='stable'))
    np.testing.assert_array_equal(sp.argsort(a, axis).glom(), np.argsort(x, axis, kind='stable').astype(np.float32))
  flat = sp.sort(a, axis=None).force()
  np.testing.assert_array_equal(flat.glom(), np.sort(x, axis=None))
  # stencil: every rank convolves its own image tiles
  img = rng.randint(-3, 4, size=(4, 2, 8, 8)).astype(np.float32)
  flt = rng.randint(-2, 3, size=(3, 2, 3, 3)).astype(np.float32)
  res = sp.stencil(sp.from_numpy(img, tile_hint=(max(1, 4 // workers), 2, 8, 8)), sp.from_numpy(flt, tile_hint=(3, 2, 3, 3)), 1).glom()
  want = np.zeros((4, 3, 8, 8), np.float32)
  for xx in range(8):
    for yy in range(8):
      for i in range(3):
        for j in range(3):
          if xx + i < 8 and yy + j < 8:
            want[:, :, xx, yy] += np.einsum('nc,fc->nf', img[:, :, xx + i, yy + j], flt[:, :, i, j])
  np.testing.assert_array_equal(res, want)
  return 2


def run_rowdot(ctx):
  """The one-pass gradient node (expr/rowdot.py) across ranks: every rank's row tiles contribute a (d,) partial that
  joins the target like a reduction's.  On the oracle backend the kernel is a NumPy stand-in attached for this check
  only (the rewrite never fires without one)."""
  from spartan_amd.examples import lreg
  from spartan_amd.expr.rowdot import RowDotColSumExpr
  rng = np.random.RandomState(11)
  xh, yh = rng.rand(101, 32).astype(np.float32), rng.rand(101, 1).astype(np.float32)
  w = rng.rand(32, 1).astype(np.float32)
  attached = not hasattr(ctx.backend, 'rowdot_colsum')
  if attached:
    def stand_in(x, wv, y):
      t = x.dot(np.asarray(wv, np.float32).reshape(-1, 1))
      return (x * (t if y is None else t - np.asarray(y).reshape(t.shape))).sum(0).astype(np.float32)
    ctx.backend.rowdot_colsum = stand_in
  try:
    x, y = sp.Val(val=sp.from_numpy(xh).force()), sp.Val(val=sp.from_numpy(yh).force())
    g = lreg.gradient(x, y, w).optimized()
    assert isinstance(g, RowDotColSumExpr)
    np.testing.assert_allclose(g.glom(), (xh * (xh.dot(w) - yh)).sum(0), rtol=2e-5)
  finally:
    if attached:
      del ctx.backend.rowdot_colsum
  return 1


def main():
  workers = int(sys.argv[1])
  use_hip = len(sys.argv) > 2 and sys.argv[2] == 'hip'
  which = os.environ.get('SPARTAN_TEST_BACKEND', 'socket')
  if which == 'own-torch':
    # a caller that lives in a torch job: it brought its own initialised group and names no backend; on a box
    # without GPUs the world keeps that group as control plane AND data plane (gloo), not the hub's sockets
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=int(os.environ['RANK']), world_size=int(os.environ['WORLD_SIZE']))
    os.environ.pop('SPARTAN_DIST_BACKEND', None)
    world = sp.World.from_env()
    assert world.note == 'gloo', world.note
  else:
    world = sp.World.from_env(backend=which)
  assert world.size == 2
  if use_hip:
    # two ranks sharing GPU 0, HBM blobs staged through the host by the debug transport
    world.staged = True
    ctx = sp.initialize('hip', num_workers=workers, world=world)
  else:
    ctx = sp.initialize(backend=NumpyBackend(), num_workers=workers, world=world)
  n = 0
  # ... and against what the REFERENCE returned for this many workers, where it was recorded (4, 8): that is what
  # pins the programs whose answer depends on the order the kernels run in -- across ranks as inside one process
  gold_path = os.path.join(ROOT, 'tests', 'golden', 'programs_w%d.npz' % ctx.num_workers)
  gold = np.load(gold_path) if os.path.exists(gold_path) else None
  for name, build, expected, tol in programs.programs():
    got = programs.run(name, build, sp, ctx.num_workers)
    programs.check(name, got, expected(), tol)
    if gold is not None and got is not None and name in gold.files:
      programs.check(name, got, gold[name], tol if tol is not None or not use_hip else None)
    n += 1
  joins_path = os.path.join(ROOT, 'tests', 'golden', 'joins_w%d.npz' % ctx.num_workers)
  if os.path.exists(joins_path):
    from tests import join_programs
    jg = np.load(joins_path)
    for name, build in join_programs.programs():
      got = np.asarray(build(sp).force().glom())
      assert got.dtype == jg[name].dtype, (name, got.dtype)
      np.testing.assert_array_equal(got, jg[name], err_msg=name)
      n += 1
  # the regular patterns must have been carried by collectives, not per-tile messages
  if workers == 2:
    a = sp.from_numpy(np.arange(64 * 32, dtype=np.float32).reshape(64, 32) % 7)
    before = dict(world.stats)
    s = sp.sum(a, axis=0).glom()                 # partial (32,) per rank -> reduce_scatter + all_gather
    np.testing.assert_array_equal(s, (np.arange(64 * 32, dtype=np.float32).reshape(64, 32) % 7).sum(0))
    assert world.stats['collectives'] >= before['collectives'] + 2, world.stats
    assert world.stats['p2p_msgs'] == before['p2p_msgs'], world.stats
    before = dict(world.stats)
    b = sp.from_numpy(np.arange(32 * 64, dtype=np.float32).reshape(32, 64) % 5)
    d = sp.dot(b, a).glom()                      # K-split map2: A slabs by p2p, partials by reduce
    np.testing.assert_array_equal(d, (np.arange(32 * 64, dtype=np.float32).reshape(32, 64) % 5).dot(
        np.arange(64 * 32, dtype=np.float32).reshape(64, 32) % 7))
    assert world.stats['collectives'] > before['collectives'], world.stats
    # the north-star shape of the K-split: both operands AND the target row-tiled one tile per rank ->
    # dot.ksplit_plan: one grouped exchange of A blocks into one slab, per column chunk one GEMM + one reduce-scatter
    os.environ['SPARTAN_DOT_CHUNK_COLS'] = '8'
    an = (np.arange(48 * 64, dtype=np.float32).reshape(48, 64) % 9) - 4
    bn = (np.arange(64 * 32, dtype=np.float32).reshape(64, 32) % 5) - 2
    for dtype in (np.float32, np.float64):
      before = dict(world.stats)
      gemms = getattr(ctx.backend, 'gemms', 0)
      got = sp.dot(sp.from_numpy(an.astype(dtype)), sp.from_numpy(bn.astype(dtype)), tile_hint=(24, 32)).force()
      np.testing.assert_array_equal(got.glom(), an.dot(bn).astype(dtype))
      assert got.dtype == dtype and sorted(ex.shape for ex in got.tiles) == [(24, 32), (24, 32)]
      assert world.stats['collectives'] - before['collectives'] == 4 + 1, world.stats     # 4 chunks (+ the glom)
      assert world.stats['p2p_msgs'] - before['p2p_msgs'] == 1, world.stats                # my block for the peer
      assert world.stats['p2p_bytes'] - before['p2p_bytes'] == 24 * 32 * an.astype(dtype).itemsize
      assert ctx.backend.gemms - gemms == 2 + 3, ctx.backend.gemms - gemms      # chunk 0: my block + the other; then one per chunk
    del os.environ['SPARTAN_DOT_CHUNK_COLS']
    n += 2
  # tall dot (outer path): B is gathered as asynchronous column chunks, one GEMM per chunk
  os.environ['SPARTAN_RHS_CHUNK_COLS'] = '8'
  ta = np.arange(128 * 32, dtype=np.float32).reshape(128, 32) % 11
  tb = np.arange(32 * 16, dtype=np.float32).reshape(32, 16) % 3
  before = dict(world.stats)
  got = sp.dot(sp.from_numpy(ta), sp.from_numpy(tb), tile_hint=(128 // workers, 16)).glom()
  np.testing.assert_array_equal(got, ta.dot(tb))
  if workers == world.size:
    assert world.stats['collectives'] >= before['collectives'] + 2, (before, world.stats)   # 2 column chunks
  os.environ['SPARTAN_RHS_CHUNK_COLS'] = '0'
  got = sp.dot(sp.from_numpy(ta), sp.from_numpy(tb), tile_hint=(128 // workers, 16)).glom()
  np.testing.assert_array_equal(got, ta.dot(tb))
  del os.environ['SPARTAN_RHS_CHUNK_COLS']
  n += 2
  # the example drivers across ranks: identical to the single-process goldens of the reference
  # (tile->worker round robin puts tiles on both ranks; every rank must see the same result)
  n += run_examples(workers)
  n += run_sparse(workers)
  n += run_auto_tiling(world)
  n += run_sort(workers)
  n += run_masked_fetch(world)
  n += run_rowdot(ctx)
  if not use_hip:
    n += run_heartbeat(world)
  if world.control.name == 'socket':
    assert 'torch' not in sys.modules, 'the socket control plane must not bring torch in'
  world.barrier()
  print('RANK %d OK %d' % (world.rank, n))
  sys.stdout.flush()


if __name__ == '__main__':
  main()
