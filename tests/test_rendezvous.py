"""The control plane of a multi-rank job (spartan_amd/rendezvous.py) and what it is for: a process that joins a job
holds ONE HIP runtime, and the RCCL that libspartan_hip.so binds is the copy installed beside that runtime.

Replaces the reference's ZeroMQ registration of workers with the master (spartan/rpc/zeromq.py:242-253,
spartan/worker.py:98-102).  No GPU needed: the maps test runs World._join's own order of imports / library loads in
fresh interpreters up to the point where RCCL wants a device."""
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  port = s.getsockname()[1]
  s.close()
  return port


@pytest.fixture
def job_env(monkeypatch):
  port = _free_port()
  monkeypatch.setenv('MASTER_ADDR', '127.0.0.1')
  monkeypatch.setenv('MASTER_PORT', str(port))
  monkeypatch.delenv('SPARTAN_RDZV_PORT', raising=False)
  monkeypatch.delenv('TORCHELASTIC_USE_AGENT_STORE', raising=False)
  return port


def _threads(n, body):
  """n ranks as threads of this process (every Client keeps per-thread connections): returns [result or exception]."""
  from spartan_amd import rendezvous
  out = [None] * n
  hub_box = []
  ready = threading.Event()

  def run(r):
    try:
      if r == 0:
        client, hub = rendezvous.join(0, n, timeout_s=20)
        hub_box.append(hub)
        ready.set()
      else:
        ready.wait(10)
        client, _ = rendezvous.join(r, n, timeout_s=20)
      try:
        out[r] = body(client, r)
      finally:
        client.close()
    except Exception as e:          # noqa: BLE001 -- handed to the asserting thread
      out[r] = e
  ts = [threading.Thread(target=run, args=(r,)) for r in range(n)]
  [t.start() for t in ts]
  [t.join(60) for t in ts]
  if hub_box:
    hub_box[0].close(5.0)
  return out


def test_rounds_mailbox_and_store(job_env):
  def body(c, r):
    c.barrier()
    got = c.all_gather_object({'rank': r})
    assert [g['rank'] for g in got] == [0, 1, 2]
    assert c.broadcast_object('from 1' if r == 1 else None, 1) == 'from 1'
    x = np.arange(6, dtype=np.float32) + r
    np.testing.assert_array_equal(c.all_reduce(x, 'ADD'), 3 * np.arange(6, dtype=np.float32) + 3)
    np.testing.assert_array_equal(c.reduce_scatter(x, 'MAX'), (np.arange(6, dtype=np.float32) + 2)[2 * r:2 * r + 2])
    red = c.reduce(x, 2, 'MIN')
    assert (red is None) == (r != 2)
    # ordered blocks between two ranks, and a rank that takes no part in an exchange
    if r == 0:
      c.send(1, 'a')
      c.send(1, 'b')
    if r == 1:
      assert [c.recv(0), c.recv(0)] == ['a', 'b']
    # store: reads never block
    assert c.get('never/set') is None
    c.set('k/%d' % r, str(r))
    c.barrier()
    assert [c.get('k/%d' % q) for q in range(3)] == ['0', '1', '2']
    c.barrier()
    c.delete('k/%d' % r)
    c.barrier()
    assert c.get('k/0') is None
    return 'ok'
  assert _threads(3, body) == ['ok'] * 3


def test_a_round_whose_reduction_fails_tells_every_rank_and_the_hub_lives_on(job_env):
  """An unknown reducer (a caller passing np.add for 'ADD') used to raise inside the hub's connection thread: rank 0
  lost its hub and every other rank its connection.  Now every rank of the round gets the error, the round is
  dropped, and the next round works."""
  from spartan_amd import rendezvous

  def body(c, r):
    x = np.arange(4, dtype=np.float32)
    with pytest.raises(rendezvous.RendezvousError, match='unknown reducer'):
      c.reduce_scatter(x, np.add)
    np.testing.assert_array_equal(c.all_reduce(x, 'ADD'), 2 * x)
    return 'ok'
  assert _threads(2, body) == ['ok'] * 2


def test_store_from_a_second_thread_while_the_driver_waits_in_a_round(job_env):
  """The heartbeat's watcher talks to the store while the driver thread is inside a collective round."""
  def body(c, r):
    if r == 1:
      seen = []
      t = threading.Thread(target=lambda: (c.set('beat', '7'), seen.append(c.get('beat'))))
      t.start()
      t.join(10)
      assert seen == ['7']
      time.sleep(0.2)
    c.barrier()                 # rank 0 waits here while rank 1's second thread uses the store
    return 'ok'
  assert _threads(2, body) == ['ok'] * 2


def test_a_rank_that_leaves_fails_the_round_it_never_joined(job_env):
  from spartan_amd import rendezvous

  def body(c, r):
    c.barrier()
    if r == 1:
      for s in c._all:          # the process dies: its sockets close without a goodbye
        s.close()
      c._all = []
      return 'left'
    with pytest.raises(rendezvous.RendezvousError, match='rank 1 left the job'):
      c.barrier()
    return 'saw it'
  assert _threads(2, body) == ['saw it', 'left']


def test_a_peer_without_the_jobs_secret_is_not_admitted(job_env, monkeypatch):
  """The handshake is a challenge: the hub sends a nonce, the peer answers HMAC(secret, nonce + job name).  A peer
  that knows the job's name (it is sent in clear) but not $SPARTAN_JOB_SECRET never reaches the pickled protocol;
  a hub that other hosts can reach refuses to start without a secret."""
  from spartan_amd import rendezvous
  monkeypatch.setenv('SPARTAN_JOB_SECRET', 'token-of-this-job')
  client, hub = rendezvous.join(0, 2, timeout_s=5)
  try:
    monkeypatch.setenv('SPARTAN_JOB_SECRET', 'something-else')
    with pytest.raises(rendezvous.RendezvousError, match='refused the proof'):
      rendezvous.Client(1, 2, 0.5, port=hub.port)
    monkeypatch.setenv('SPARTAN_JOB_SECRET', 'token-of-this-job')
    other = rendezvous.Client(1, 2, 5.0, port=hub.port)
    other.set('k', 'v')
    assert client.get('k') == 'v'
    other.close()
  finally:
    client.close()
    hub.close(1.0)
  monkeypatch.delenv('SPARTAN_JOB_SECRET')
  monkeypatch.setenv('MASTER_ADDR', '10.1.2.3')
  with pytest.raises(rendezvous.RendezvousError, match='SPARTAN_JOB_SECRET'):
    rendezvous.Hub(2, 5.0)


def test_a_failed_wait_drops_its_round(job_env):
  """A round that times out (a rank never joins it) is removed from the hub's table by the rank that gives up: its
  payloads do not stay for the life of the hub."""
  from spartan_amd import rendezvous
  client, hub = rendezvous.join(0, 2, timeout_s=0.6)
  other, _ = rendezvous.join(1, 2, timeout_s=0.6)
  try:
    with pytest.raises(rendezvous.RendezvousError, match='timed out'):
      client.all_reduce(np.zeros(1 << 16, np.float32), 'ADD')
    assert not hub.rounds
  finally:
    other.close()
    client.close()
    hub.close(1.0)


def test_a_rank_that_never_arrives_times_the_round_out_with_its_number(job_env):
  """A hung rank (connected, silent) must not block the others for ever: the round fails after the deadline and
  names who was missing."""
  from spartan_amd import rendezvous
  hub = rendezvous.Hub(2, timeout_s=0.6)
  try:
    c0 = rendezvous.Client(0, 2, 5.0, port=hub.port)
    c1 = rendezvous.Client(1, 2, 5.0, port=hub.port)          # joins, then never calls the barrier
    t0 = time.time()
    with pytest.raises(rendezvous.RendezvousError, match=r'timed out after 1 s waiting for ranks \[1\]|timed out after 0 s waiting for ranks \[1\]'):
      c0.barrier()
    assert time.time() - t0 < 5
    c0.close()
    c1.close()
  finally:
    hub.close(1.0)


def test_hub_moves_past_a_taken_port_and_clients_pass_over_strangers(job_env):
  """MASTER_PORT is held by something that is not our hub (it accepts and says nothing useful): rank 0 binds the
  next port, and the clients find it there by the handshake."""
  stranger = socket.socket()
  stranger.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
  stranger.bind(('127.0.0.1', job_env))
  stranger.listen(8)

  def answer_garbage():
    stranger.settimeout(5)
    try:
      while True:
        conn, _ = stranger.accept()
        conn.sendall(b'HTTP/1.1 400')
        conn.close()
    except OSError:
      pass
  t = threading.Thread(target=answer_garbage, daemon=True)
  t.start()
  try:
    def body(c, r):
      c.barrier()
      return c._port
    ports = _threads(2, body)
    assert ports == [job_env + 1, job_env + 1], ports
  finally:
    stranger.close()


def test_agent_store_port_is_never_touched(job_env, monkeypatch):
  """Under torch.distributed.run (static rendezvous) MASTER_PORT is the launcher's TCPStore: the hub starts one up."""
  from spartan_amd import rendezvous
  monkeypatch.setenv('TORCHELASTIC_USE_AGENT_STORE', 'True')
  assert rendezvous.endpoint() == ('127.0.0.1', job_env + 1)
  monkeypatch.setenv('SPARTAN_RDZV_PORT', '4242')
  assert rendezvous.endpoint() == ('127.0.0.1', 4242)


def test_another_jobs_hub_is_passed_over(job_env, monkeypatch):
  """A hub of a job with another key (another WORLD_SIZE here) on our port: our rank 0 sits one port up, and our
  client goes there."""
  from spartan_amd import rendezvous
  other = rendezvous.Hub(5, timeout_s=5)             # same address and port, world of 5: another key
  try:
    assert other.port == job_env
    assert _threads(2, lambda c, r: c._port) == [job_env + 1, job_env + 1]
  finally:
    other.close(0.0)


_JOIN_PROBE = r'''
import json, os, sys
sys.path.insert(0, %(root)r)
import spartan_amd as sp
from spartan_amd import comm
err = None
try:
  sp.World.from_env(backend='rccl')        # World._join: library, device, rendezvous, RCCL -- in the product's order
except Exception as e:
  err = str(e)
paths = None
try:
  paths = comm.rccl_paths()
except Exception as e:
  paths = {'error': str(e)}
print('PROBE ' + json.dumps({'err': err, 'torch': 'torch' in sys.modules, 'mapped': comm.mapped_runtimes(), 'paths': paths}))
'''


def test_join_order_leaves_one_hip_runtime_and_binds_its_rccl():
  """World._join as a real `--gpus 2` rank runs it, in fresh interpreters: afterwards the process maps exactly one
  libamdhip64, one libhsa-runtime64, and the librccl beside them; torch was never imported.  (Without two GPUs the
  communicator itself cannot come up -- the job then stops with the reason, which is asserted too; with them this
  is the start-up of a real run.)"""
  import json
  port = _free_port()
  procs = []
  for rank in range(2):
    env = dict(os.environ)
    env.update({'RANK': str(rank), 'LOCAL_RANK': str(rank), 'WORLD_SIZE': '2', 'MASTER_ADDR': '127.0.0.1',
                'MASTER_PORT': str(port), 'SPARTAN_COMM_SELFTEST_S': '20'})
    env.pop('SPARTAN_DIST_BACKEND', None)
    env.pop('SPARTAN_RCCL_LIB', None)
    procs.append(subprocess.Popen([sys.executable, '-c', _JOIN_PROBE % {'root': ROOT}], env=env, cwd=ROOT,
                                  stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
  for rank, p in enumerate(procs):
    out = p.communicate(timeout=240)[0].decode('utf-8', 'replace')
    line = [l for l in out.splitlines() if l.startswith('PROBE ')]
    assert line, out[-3000:]
    rec = json.loads(line[-1][6:])
    assert rec['torch'] is False, rec
    mapped = rec['mapped']
    assert len(mapped['libamdhip64']) == 1 and len(mapped['libhsa-runtime64']) == 1, mapped
    assert len(mapped['librccl']) == 1, mapped
    runtime_dir = os.path.dirname(mapped['libamdhip64'][0])
    assert os.path.dirname(mapped['librccl'][0]) == runtime_dir, mapped
    assert rec['paths']['hip_runtime_path'] == rec['paths']['own_hip_runtime_path'] == mapped['libamdhip64'][0], rec
    assert rec['paths']['lib_path'] == mapped['librccl'][0], rec
    if rec['err'] is not None:                 # fewer than two GPUs: stopped loudly, never a quiet fallback
      assert 'did not come up' in rec['err'], rec['err']


def test_rccl_linked_against_another_runtime_is_refused():
  """The copy of RCCL PyTorch ships runs on PyTorch's own libamdhip64: in a process whose tiles live on the system
  runtime, libspartan_hip.so refuses it (named explicitly here; found by soname in a mixed process) with the reason."""
  try:
    import importlib.util
    spec = importlib.util.find_spec('torch')
  except Exception:
    spec = None
  if spec is None or not spec.origin:
    pytest.skip('no torch installation to take a foreign RCCL from')
  foreign = os.path.join(os.path.dirname(spec.origin), 'lib', 'librccl.so')
  if not os.path.exists(foreign):
    pytest.skip('torch ships no librccl.so here')
  code = ('import sys; sys.path.insert(0, %r)\n'
          'from spartan_amd import _hip\n'
          'ok = _hip.lib().sp_comm_available()\n'
          'print("AVAILABLE", ok, _hip.lib().sp_last_error().decode())\n' % ROOT)
  env = dict(os.environ, SPARTAN_RCCL_LIB=foreign)
  out = subprocess.run([sys.executable, '-c', code], env=env, cwd=ROOT, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=240).stdout.decode('utf-8', 'replace')
  assert 'AVAILABLE 0' in out and 'another HIP runtime' in out, out[-2000:]


@pytest.mark.gpu
def test_one_rank_rccl_communicator_in_a_torch_free_process():
  """tools/rccl_one_rank.py in a fresh interpreter on the GPU: sp_comm_* binds the RCCL beside the library's own HIP
  runtime, every primitive passes the start-up self-test on a one-rank communicator, one runtime of each kind is
  mapped and torch is never imported."""
  import json
  env = dict(os.environ)
  for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'SPARTAN_RCCL_LIB'):
    env.pop(k, None)
  p = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'rccl_one_rank.py')], env=env, cwd=ROOT,
                     stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
  out = p.stdout.decode('utf-8', 'replace')
  assert p.returncode == 0, (out[-2000:], p.stderr.decode('utf-8', 'replace')[-3000:])
  rec = json.loads(out[out.index('{'):out.rindex('}') + 1])
  assert rec['self_test'] == [True, 'ok'] and rec['torch_in_process'] is False
  assert all(len(v) == 1 for v in rec['mapped'].values()), rec['mapped']
  assert os.path.dirname(rec['paths']['lib_path']) == os.path.dirname(rec['paths']['own_hip_runtime_path'])
  assert rec['paths']['hip_runtime_path'] == rec['paths']['own_hip_runtime_path']
