"""save / load / pickle / unpickle in the reference's on-disk layout (spartan/expr/fio.py).
The reference's writer cannot run here (Python-2 str/bytes file handling), so the layout is pinned by
(i) a file assembled byte by byte the way fio.py:70-112 assembles it, read back through `load`, and
(ii) round trips through both backends, zipped and unzipped, for several tilings."""
import os

import numpy as np
import pytest

import spartan_amd as sp


def _reference_style_tile_file(path, prefix, ul, lr, data):
  """fio.py:70-112 restated: magic, 2-byte LE dict length, str(dict) space-padded to 16 B, raw bytes."""
  tile_dict = {'dtype': str(data.dtype), 'shape': data.shape, 'type': "DENSITY", 'lr': lr, 'ul': ul}
  cnt = b"\x93NUMPY\x01\x00"
  dict_cnt = str(tile_dict)
  if (len(cnt) + 2 + len(dict_cnt)) % 16 != 0:
    dict_cnt += (16 - (len(cnt) + 2 + len(dict_cnt)) % 16) * ' '
  cnt += bytes([len(dict_cnt) % 256, len(dict_cnt) // 256]) + dict_cnt.encode('latin-1')
  os.makedirs(os.path.join(path, prefix), exist_ok=True)
  with open(os.path.join(path, prefix, '%s_%s_%s_spf' % (prefix, str(ul), str(lr))), 'wb') as fp:
    fp.write(cnt)
    fp.write(data.tobytes())


def _check(tmp):
  a = (np.arange(48 * 10, dtype=np.float32).reshape(48, 10) % 23) - 7
  # (i) files laid out by hand exactly as the reference lays them out, two row tiles of 24
  _reference_style_tile_file(tmp, 'ref', (0, 0), (24, 10), a[:24])
  _reference_style_tile_file(tmp, 'ref', (24, 0), (48, 10), a[24:])
  with open(os.path.join(tmp, 'ref', 'ref_dist.spf'), 'w') as fp:
    fp.write('48 10 \n24 10 \nfloat32\nDENSITY\n')            # fio.py:115-131
  L = sp.load('ref', path=tmp).evaluate()
  assert L.dtype == np.float32 and L.shape == (48, 10)
  assert sorted(ex.ul for ex in L.tiles) == [(0, 0), (24, 0)]
  np.testing.assert_array_equal(L.glom(), a)
  # (ii) round trips
  for dtype in (np.float32, np.float64, np.int64):
    x = a.astype(dtype)
    for hint in (None, (12, 10), (48, 5)):
      X = sp.from_numpy(x, tile_hint=hint).evaluate() if hint else sp.from_numpy(x).evaluate()
      for z in (False, True):
        name = 'rt_%s_%s_%d' % (np.dtype(dtype).name, 'x'.join(map(str, hint)) if hint else 'd', z)
        assert sp.save(X, name, path=tmp, iszip=z) is True
        Y = sp.load(name, path=tmp, iszip=z).evaluate()
        assert Y.dtype == np.dtype(dtype) and sorted(Y.tiles) == sorted(X.tiles)
        np.testing.assert_array_equal(Y.glom(), x)
        assert sp.pickle(X, name + 'p', path=tmp, iszip=z) is True
        np.testing.assert_array_equal(sp.unpickle(name + 'p', path=tmp, iszip=z).glom(), x)
  # an expression can be saved directly; a missing prefix is an IOError as in the reference (fio.py:196-197)
  assert sp.save(sp.from_numpy(a) + 1, 'expr', path=tmp)
  np.testing.assert_array_equal(sp.load('expr', path=tmp).glom(), a + 1)
  with pytest.raises(IOError):
    sp.load('nothing-here', path=tmp)
  # from_file: dense .npy / .npz / Matrix Market read on the driver (write_array.py:380-421)
  import scipy.io
  np.save(os.path.join(tmp, 'dense.npy'), a)
  np.savez(os.path.join(tmp, 'dense.npz'), only=a)
  scipy.io.mmwrite(os.path.join(tmp, 'dense.mtx'), a.astype(np.float64))
  np.testing.assert_array_equal(sp.from_file(os.path.join(tmp, 'dense.npy'), sparse=False).glom(), a)
  np.testing.assert_array_equal(sp.from_file(os.path.join(tmp, 'dense.npz'), sparse=False, tile_hint=(12, 10)).glom(), a)
  np.testing.assert_array_equal(sp.from_file(os.path.join(tmp, 'dense.mtx'), file_type='mm').glom(), a.astype(np.float64))
  # ---- sparse arrays (tests/test_fio.py:test_fio_sparse / test_fio_partial_sparse in the reference)
  import scipy.sparse as sps
  rng = np.random.RandomState(3)
  m = sps.random(60, 44, density=0.1, format='coo', dtype=np.float32, random_state=rng)
  S = sp.from_numpy(m, tile_hint=(20, 44)).evaluate()
  assert S.sparse and sps.issparse(S.glom())
  np.testing.assert_array_equal(S.glom().toarray(), m.toarray())
  for z in (False, True):
    assert sp.save(S, 'sps%d' % z, path=tmp, iszip=z) is True
    T = sp.load('sps%d' % z, path=tmp, iszip=z).evaluate()
    assert T.sparse and sorted(T.tiles) == sorted(S.tiles) and T.dtype == np.float32
    np.testing.assert_array_equal(T.glom().toarray(), m.toarray())
    assert sp.pickle(S, 'spp%d' % z, path=tmp, iszip=z) is True
    np.testing.assert_array_equal(sp.unpickle('spp%d' % z, path=tmp, iszip=z).glom().toarray(), m.toarray())
  # the file layout of a sparse tile (fio.py:98-104): header file + <name>.npz with row / col / data / shape
  first = sorted(S.tiles)[0]
  z = np.load(os.path.join(tmp, 'sps0', 'sps0_%s_%s.npz' % (str(first.ul), str(first.lr))))
  assert sorted(z.files) == ['col', 'data', 'row', 'shape'] and tuple(z['shape']) == first.shape
  assert open(os.path.join(tmp, 'sps0', 'sps0_dist.spf')).read().split('\n')[3] == 'SPARSE'
  # partial_load / partial_unpickle: some tiles onto chosen workers
  some = {ex: 0 for ex in sorted(S.tiles)[:2]}
  for loader, name in ((sp.partial_load, 'sps0'), (sp.partial_unpickle, 'spp0')):
    got = loader(some, name, path=tmp)
    assert sorted(got) == sorted(some)
    ctx = sp.get_context()
    for ex, tid in got.items():
      assert tid.worker == 0
      blob = ctx.tile(tid).data
      np.testing.assert_array_equal(ctx.backend.sparse_to_host(blob).toarray(), m.toarray()[ex.to_slice()])
  dense_some = {ex: 0 for ex in sorted(sp.load('rt_float32_12x10_0', path=tmp).evaluate().tiles)[:2]}
  for ex, tid in sp.partial_load(dense_some, 'rt_float32_12x10_0', path=tmp).items():
    np.testing.assert_array_equal(sp.get_context().backend.to_numpy(sp.get_context().tile(tid).data), a[ex.to_slice()])
  # from_file, sparse: the four .npy files of a COO matrix, and a sparse Matrix Market file
  base = os.path.join(tmp, 'coo')
  np.save(base + '_shape.npy', np.array(m.shape))
  np.save(base + '_row.npy', m.row)
  np.save(base + '_col.npy', m.col)
  np.save(base + '_data.npy', m.data)
  F = sp.from_file(base).evaluate()
  assert F.sparse
  np.testing.assert_array_equal(F.glom().toarray(), m.toarray())
  scipy.io.mmwrite(os.path.join(tmp, 'sparse.mtx'), m.astype(np.float64))
  G = sp.from_file(os.path.join(tmp, 'sparse.mtx'), file_type='mm').evaluate()
  assert G.sparse and G.dtype == np.float32                      # narrowed like write_array.py:416-417
  H = sp.from_file_parallel(os.path.join(tmp, 'sparse.mtx'), file_format='mm', tile_hint=(20, 44)).evaluate()
  np.testing.assert_allclose(H.glom().toarray(), m.toarray(), rtol=1e-6)
  np.testing.assert_allclose(G.glom().toarray(), m.toarray(), rtol=1e-6)
  # tocoo / tile_operation / checkpoint
  np.testing.assert_array_equal(sp.tocoo(sp.Val(val=S)).glom().toarray(), m.toarray())
  counts = sp.tile_operation(sp.Val(val=S), lambda arr, ex: [(ex, ex.shape[0])]).evaluate()
  assert sorted(v[0] for v in counts.values()) == [20, 20, 20]
  ck = sp.checkpoint(sp.from_numpy(a) * 2)
  np.testing.assert_array_equal(ck.evaluate().glom(), a * 2)
  assert ck.ready
  np.testing.assert_array_equal(ck.load_data(None).glom(), a * 2)
  res = ck.evaluate()
  lost = sorted(res.tiles)[0]
  res.bad_tiles.append(lost)
  res = ck.load_data(res)
  assert res.bad_tiles == []
  np.testing.assert_array_equal(res.glom(), a * 2)


@pytest.mark.parametrize('workers', [1, 4])
def test_fio_host_framework(workers, tmp_path):
  from oracle.np_backend import NumpyBackend
  sp.initialize(backend=NumpyBackend(), num_workers=workers)
  try:
    _check(str(tmp_path))
  finally:
    sp.shutdown()


@pytest.mark.gpu
@pytest.mark.parametrize('workers', [1, 3])
def test_fio_hip(workers, tmp_path):
  sp.initialize('hip', num_workers=workers)
  try:
    _check(str(tmp_path))
  finally:
    sp.shutdown()


def _failed_worker(backend_factory):
  """tests/test_checkpoint.py:test1: a checkpointed value survives the loss of a worker (its bad tiles are reloaded
  from disk when the value is used again); a value without a checkpoint is recomputed from its dependencies."""
  ctx = sp.initialize(backend=backend_factory(), num_workers=4)
  try:
    a, b, c = sp.ones((10, 10)), sp.ones((10, 10)), sp.ones((10, 10))
    x = a + b + c
    y = x + x
    z = sp.checkpoint(y + y, mode='disk')
    zv = z.evaluate()
    plain = (x * 2)
    pv = plain.evaluate()
    ctx.mark_failed_worker(0)
    assert len(zv.bad_tiles) > 0 and len(pv.bad_tiles) > 0
    res = z + z
    np.testing.assert_array_equal(res.glom(), np.ones((10, 10)) * 24)
    assert zv.bad_tiles == []                                   # reloaded in place (checkpoint.py:27-37)
    np.testing.assert_array_equal((plain + 1).glom(), np.ones((10, 10)) * 7)     # recomputed
  finally:
    sp.shutdown()


def test_failed_worker_reload_and_recompute_cpu():
  from oracle.np_backend import NumpyBackend
  _failed_worker(NumpyBackend)


@pytest.mark.gpu
def test_failed_worker_reload_and_recompute_gpu():
  from spartan_amd.backend_hip import HipBackend
  _failed_worker(HipBackend)


def test_heartbeat_marks_a_silent_worker_failed_and_the_value_is_recomputed():
  """Failure DETECTION (master.py:142-146, worker.py:347-368): the device stops answering the heartbeat's probe,
  the watcher declares the rank failed after `threshold` missed beats, the driver's next evaluation applies it --
  the workers' tiles become bad tiles -- and the cached value is recomputed from its dependencies."""
  import time
  from oracle.np_backend import NumpyBackend
  sp.initialize(backend=NumpyBackend(), num_workers=3)
  try:
    ctx = sp.get_context()
    alive = {'ok': True}
    hb = ctx.start_heartbeat(interval=0.05, threshold=12, probe=lambda: alive['ok'])   # (0.6 s of silence: a loaded test box may starve the beat thread for a while)
    a = np.ones((12, 5), np.float32)
    e = sp.ones((12, 5)) * 2 + 1          # (a value with lineage: builders can be re-run, loaded data cannot)
    first = e.evaluate()
    time.sleep(0.4)
    assert ctx.apply_failures() == [] and not first.bad_tiles          # beating: nobody is marked
    alive['ok'] = False                                                 # the "GPU" hangs
    deadline = time.time() + 5
    while not hb.failed_ranks and time.time() < deadline:
      time.sleep(0.02)
    assert hb.failed_ranks == {0}
    assert not first.bad_tiles                                          # nothing changes behind the driver's back
    second = (e + 0).evaluate()                                         # safe point: failures are applied here
    assert ctx.failed_workers == {0, 1, 2} and len(first.bad_tiles) == len(first.tiles)
    np.testing.assert_array_equal(second.glom(), a * 2 + 1)            # e was recomputed, not read from dead tiles
  finally:
    sp.shutdown()


def test_heartbeat_second_failure_of_a_recovered_rank_and_agreement_without_the_dead_rank():
  """(i) A rank whose beats resume can fail AGAIN and is marked again (failed_ranks is cleared when it beats);
  (ii) the ranks agree on verdicts through the key-value store, so a rank that never reaches the safe point does
  not block the others: it is added to the verdict after the deadline."""
  import time
  from oracle.np_backend import NumpyBackend
  from spartan_amd import heartbeat
  sp.initialize(backend=NumpyBackend(), num_workers=2)
  try:
    ctx = sp.get_context()
    alive = {'ok': True}
    hb = ctx.start_heartbeat(interval=0.03, threshold=10, probe=lambda: alive['ok'])
    e = sp.ones((8, 3)) + 1
    for episode in range(2):
      first = e.evaluate()
      alive['ok'] = False
      deadline = time.time() + 5
      while not hb.failed_ranks and time.time() < deadline:
        time.sleep(0.01)
      assert hb.failed_ranks == {0}, episode
      assert sorted(ctx.apply_failures()) == [0, 1], episode            # marked in BOTH episodes
      assert len(first.bad_tiles) == len(first.tiles)
      alive['ok'] = True
      deadline = time.time() + 5
      while hb.failed_ranks and time.time() < deadline:
        time.sleep(0.01)
      assert not hb.failed_ranks                                        # beating again
      np.testing.assert_array_equal((e + 0).evaluate().glom(), np.full((8, 3), 2, np.float32))
    hb.stop()
    ctx.heartbeat = None
    # (ii) three ranks share one store; rank 2 never posts
    store = heartbeat._LocalStore()

    class W(object):
      def __init__(self, rank):
        self.rank, self.size, self.distributed = rank, 3, True

    class C(object):
      backend = None

      def __init__(self, rank):
        self.world = W(rank)
    hbs = [heartbeat.Heartbeat(C(r), interval=0.02, threshold=5, probe=lambda: True, store=store) for r in range(2)]
    for h in hbs:
      h.agree_floor_s = 0.5
    import threading
    out = [None, None]

    def run(i):
      out[i] = hbs[i].agree([] if i else [1])
    ts = [threading.Thread(target=run, args=(i,)) for i in range(2)]
    t0 = time.time()
    [t.start() for t in ts]
    [t.join(10) for t in ts]
    assert out[0] == out[1] == [1, 2], out                               # union of the posts + the rank that never came
    assert time.time() - t0 < 5
    # the dead rank costs that wait ONCE: the next safe point does not wait for it and reports nothing new
    ts = [threading.Thread(target=lambda i=i: out.__setitem__(i, hbs[i].agree([]))) for i in range(2)]
    t0 = time.time()
    [t.start() for t in ts]
    [t.join(10) for t in ts]
    assert out[0] == out[1] == [], out
    assert time.time() - t0 < 0.3, time.time() - t0
    # (iii) a live rank that reaches the safe point AFTER the others gave up on it finds their posts, reads that it
    # was given up on, and arrives at the same verdict (it marks its own workers' tiles bad like everyone else)
    store = heartbeat._LocalStore()
    hbs = [heartbeat.Heartbeat(C(r), interval=0.02, threshold=5, probe=lambda: True, store=store) for r in range(3)]
    for h in hbs:
      h.agree_floor_s = 0.4
    hbs[0].failed_ranks.add(1)          # the local watchers disagree about rank 1: must not change who is read
    out = [None, None, None]

    def run3(i):
      if i == 2:
        time.sleep(1.2)
      out[i] = hbs[i].agree([])
    ts = [threading.Thread(target=run3, args=(i,)) for i in range(3)]
    [t.start() for t in ts]
    [t.join(10) for t in ts]
    assert out[0] == out[1] == out[2] == [2], out
    # (iv) the posts of old safe points are deleted: after many rounds the store holds two rounds' worth at most
    for rnd in range(6):
      ts = [threading.Thread(target=lambda i=i: hbs[i].agree([])) for i in range(3)]
      [t.start() for t in ts]
      [t.join(10) for t in ts]
    assert len([k for k in store._d if k.startswith('spartan_hb_agree/')]) <= 2 * 2 * 3, sorted(store._d)
    # (v) a rank the others once gave up on (it was late, case iii) is counted on again once its posts are read, and
    # until then it FOLLOWS: a failure only ITS watcher flags must not make its verdict differ from the others'
    store = heartbeat._LocalStore()
    hbs = [heartbeat.Heartbeat(C(r), interval=0.02, threshold=5, probe=lambda: True, store=store) for r in range(3)]
    for h in hbs:
      h.agree_floor_s = 0.4
    out = [None, None, None]

    def late2(i):
      if i == 2:
        time.sleep(1.2)
      out[i] = hbs[i].agree([])
    ts = [threading.Thread(target=late2, args=(i,)) for i in range(3)]
    [t.start() for t in ts]
    [t.join(10) for t in ts]
    assert out[0] == out[1] == out[2] == [2] and all(h._given_up == {2} for h in hbs)
    # rank 2 posts AFTER the others have left the safe point (nobody waits for it): its finding about rank 1 is in
    # nobody's verdict, its own included, and it stays given up on every rank
    def unheard(i):
      if i == 2:
        time.sleep(0.3)
      out[i] = hbs[i].agree([1] if i == 2 else [])
    ts = [threading.Thread(target=unheard, args=(i,)) for i in range(3)]
    [t.start() for t in ts]
    [t.join(10) for t in ts]
    assert out[0] == out[1] == out[2] == [], out
    assert all(h._given_up == {2} for h in hbs), [h._given_up for h in hbs]
    assert hbs[2]._carry == {1}
    # next safe point: rank 2 posts first, the others read it: the carried finding reaches every verdict through
    # the counted-on ranks' unions, and rank 2 is counted on again everywhere
    def heard(i):
      if i != 2:
        time.sleep(0.2)
      out[i] = hbs[i].agree([])
    ts = [threading.Thread(target=heard, args=(i,)) for i in range(3)]
    [t.start() for t in ts]
    [t.join(10) for t in ts]
    assert out[0] == out[1] == out[2] == [1], out
    assert all(h._given_up == {1} for h in hbs), [h._given_up for h in hbs]
    assert hbs[2]._carry == set()
  finally:
    sp.shutdown()
