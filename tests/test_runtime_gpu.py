"""The tile store and the collectives of the C-ABI (spartan_amd/csrc/runtime.hip) on the GPU box:
library-owned blobs driven by a host without torch, and the direct RCCL transport of spartan_amd/comm.py on a
one-rank communicator (RCCL refuses two ranks on one device, so N > 1 runs only on the multi-GPU node; every
primitive, the rendezvous, the stream ordering and the start-up self-test are exercised here)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def test_torch_free_host_runs_the_tile_path():
  out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'torch_free_host.py')], cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
  text = out.stdout.decode('utf-8', 'replace')
  assert out.returncode == 0, text[-3000:]
  assert 'torch-free host OK' in text and 'collectives OK' in text, text[-2000:]


def test_product_runs_without_torch():
  """The whole host framework on the HIP backend -- builders, fusion, map / reduce / argmax / dot, glom -- in a fresh
  interpreter in which torch is never imported: tiles are blobs of the library's own store (sp_blob_*), streams
  and events the C-ABI's.  The store's counters are asserted as well: blobs are live while arrays are, and
  destroyed ones are pooled for reuse."""
  prog = (
      "import sys, numpy as np\n"
      "import spartan_amd as sp\n"
      "from spartan_amd import devarray as D\n"
      "ctx = sp.initialize('hip', num_workers=3)\n"
      "live0 = D.blob_stats()[0]\n"
      "a = np.arange(96 * 64, dtype=np.float32).reshape(96, 64) % 7 - 3\n"
      "A = sp.from_numpy(a)\n"
      "r = (A * A + A - 1).optimized().force()\n"
      "assert D.blob_stats()[0] > live0\n"
      "t = ctx.tile(list(r.tiles.values())[0]).data\n"
      "assert isinstance(t, D.DevArray) and t.storage.on_device\n"
      "np.testing.assert_array_equal(r.glom(), a * a + a - 1)\n"
      "np.testing.assert_array_equal(sp.sum(A, 0).glom(), a.sum(0))\n"
      "np.testing.assert_array_equal(sp.argmax(A, 1).glom(), a.argmax(1))\n"
      "np.testing.assert_array_equal(sp.dot(A, sp.from_numpy(a.T.copy())).glom(), a.dot(a.T))\n"
      "del r, t, A\n"
      "sp.zeros((4, 4)).force()      # a safe point: tiles of dead arrays are destroyed here\n"
      "live1, pooled = D.blob_stats()\n"
      "assert pooled > 0, (live1, pooled)\n"
      "assert 'torch' not in sys.modules, 'torch was imported'\n"
      "print('NO TORCH OK', live0, live1, pooled)\n")
  out = subprocess.run([sys.executable, '-c', prog], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
  text = out.stdout.decode('utf-8', 'replace')
  assert out.returncode == 0 and 'NO TORCH OK' in text, text[-3000:]


def test_rccl_transport_one_rank():
  from spartan_amd import comm
  from spartan_amd import devarray as D
  t = comm.RcclTransport(1, 0, comm.RcclTransport.unique_id())
  try:
    ok, msg = t.self_test(60.0)
    assert ok, msg
    x = D.from_numpy(np.arange(1 << 20, dtype=np.float32))
    # asynchronous transfers: issued on the side stream behind the producer, consumed after wait()
    y = D.empty(x.shape, np.float32)
    x = x * np.float32(2.0)
    h = t.reduce_scatter(y, x, 'ADD', async_=True)
    z = D.empty(x.shape, np.float32)
    h2 = t.all_gather_into(z, x, async_=True)
    h.wait()
    h2.wait()
    y = y + np.float32(1.0)
    D.synchronize()
    want = np.arange(1 << 20, dtype=np.float32) * 2
    np.testing.assert_array_equal(y.numpy(), want + 1)
    np.testing.assert_array_equal(z.numpy(), want)
    b = D.from_numpy(np.ones(1000, np.bool_))
    t.all_reduce(b, 'ADD')                     # bool travels as bytes
    i = D.from_numpy(np.arange(777, dtype=np.int64))
    t.reduce(i, 0, 'MAX')
    t.broadcast(i, 0)
    h3 = t.exchange([(0, i)], [(0, D.empty(i.shape, np.int64))], async_=True)
    del h3                                     # dropped without wait(): the buffers are held until the transfer is over
    D.synchronize()
    np.testing.assert_array_equal(i.numpy(), np.arange(777))
  finally:
    t.close()


def test_world_counts_and_one_rank_world():
  from spartan_amd import comm
  w = comm.World()
  assert not w.distributed and w.transport is None
  assert w.broadcast_object({'a': 1}, 0) == {'a': 1} and w.all_gather_object(3) == [3]


def test_bench_launches_its_own_ranks():
  """`python3 bench.py --gpus 2` as a PLAIN process (no torch.distributed.run): the script starts its two ranks
  itself; on this one-GPU box they share the device over the staged debug transport, and the one JSON line says
  so (`rccl.ranks` == 0, `valid_scaling_measurement` false) -- on a multi-GPU node the same command yields the
  RCCL run."""
  import json
  env = dict(os.environ, SPARTAN_BENCH_DIST_ROWS='256', SPARTAN_BENCH_LREG_ROWS='4096', SPARTAN_BENCH_LREG_COLS='256',
             SPARTAN_BENCH_LREG_STEPS='3', SPARTAN_BENCH_KMEANS_POINTS='8192', SPARTAN_BENCH_KMEANS_K='64')
  out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--size', '2048', '--steps', '2',
                        '--warmup', '1'], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
  text = out.stdout.decode('utf-8', 'replace').strip()
  assert out.returncode == 0, (text[-2000:], out.stderr.decode('utf-8', 'replace')[-3000:])
  lines = text.splitlines()
  assert len(lines) == 1, 'stdout must carry exactly one line, got %d' % len(lines)
  line = json.loads(lines[0])
  assert line['n_gpus'] == 2 and line['steps'] == 2 and line['value'] > 0
  assert line['launcher'].startswith('bench.py self-launch')
  assert line['rccl']['ranks'] in (0, 2)
  if line['rccl']['ranks'] == 0:
    assert line['valid_scaling_measurement'] is False
  for key in ('dot_breakdown', 'hbm_dist', 'lreg_dist', 'kmeans_dist'):
    assert key in line and 'error' not in line[key], (key, line.get(key))


def test_bench_launcher_reports_a_failed_rank():
  """A rank that dies gives rc != 0 and the reason in the line (here: an order the ranks cannot split)."""
  import json
  out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--size', '2049', '--steps', '1',
                        '--no-extras'], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
  assert out.returncode != 0
  line = json.loads(out.stdout.decode().strip().splitlines()[-1])
  assert line['value'] is None and 'FAILED' in line['error']
