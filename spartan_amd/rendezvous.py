"""The control plane of a multi-process job: rendezvous, barriers, small host objects, a key-value store.

Replaces the reference's ZeroMQ rendezvous and RPC bookkeeping -- workers register with the master at a known
address, the master waits for all of them, then hands every worker the worker table
(spartan/rpc/zeromq.py:242-253, spartan/worker.py:98-102, spartan/master.py:76-100) -- for a static world of one
process per GPU.  Only the standard library is used: a process that joins a job through here loads exactly the
libraries libspartan_hip.so itself depends on (one HIP runtime, one HSA runtime, and the RCCL of that ROCm), which
is what the RCCL data plane (spartan_amd/comm.py, sp_comm_*) needs -- the device pointers, streams and events it
is handed all belong to that one runtime.

Rank 0 serves (`Hub`, a thread per connection); every rank -- rank 0 too -- is a `Client`.  The hub offers

  collective rounds   barrier / all-gather / broadcast of pickled host objects and, for the host-staged debug
                      transport, reductions of NumPy arrays in rank order (deterministic)
  a mailbox           ordered point-to-point messages between two ranks (the staged form of grouped send/recv)
  a key-value store   the heartbeat's counters and verdicts (reads never block)

Where the hub listens: `SPARTAN_RDZV_PORT` if set, else `MASTER_PORT` -- or `MASTER_PORT + 1` when the job was
started by torch.distributed.run with its agent store (TORCHELASTIC_USE_AGENT_STORE: the launcher's own TCPStore
holds MASTER_PORT; it is never touched).  If that port is taken rank 0 moves up to 15 ports further; a client walks the
same ports and recognises its hub by a handshake carrying a key derived from MASTER_ADDR / MASTER_PORT / WORLD_SIZE
/ the launcher's run id, so a neighbouring job's hub (or anything else that listens there) is passed over.

Messages are pickles, as the reference's RPC payloads are (spartan/rpc/serialization*.pyx, cloudpickle): the hub is for
the ranks of ONE job on a trusted network.  It listens on the loopback interface when MASTER_ADDR is local; anything
that connects must present the job key before a single message is read, and SPARTAN_JOB_ID (any string, the same on
every rank) makes that key unguessable where the address is shared.

A rank whose connection drops without saying goodbye is recorded as gone: every round it has not joined and every
receive from it fail on the ranks waiting for them with the reason, instead of blocking them for ever.
"""
import collections
import hashlib
import os
import pickle
import socket
import struct
import threading
import time

import hmac

import numpy as np

_MAGIC = b'SPRDZV01'
_OK, _NO = b'SPRDZVOK', b'SPRDZVNO'
_PORT_SPAN = 16


class RendezvousError(RuntimeError):
  pass


def _job_key(world_size):
  """The job's NAME on the wire (not a secret: it is sent in clear so that a hub of another job on the same port
  range can be told apart)."""
  env = os.environ
  text = '|'.join([env.get('MASTER_ADDR', '127.0.0.1'), env.get('MASTER_PORT', '29500'), str(world_size),
                   env.get('TORCHELASTIC_RUN_ID', ''), env.get('SPARTAN_JOB_ID', '')])
  return hashlib.sha256(text.encode()).digest()


def _is_loopback(addr):
  return addr in ('127.0.0.1', 'localhost', '::1') or addr.startswith('127.')


def _job_secret(world_size):
  """What a peer must know to be let in (never sent: the hub sends a nonce, the peer answers with
  HMAC-SHA256(secret, nonce + name)).  $SPARTAN_JOB_SECRET, a token the launcher hands to every rank.  A job whose hub
  is reachable from other hosts must set it: after the handshake the hub unpickles what a peer sends.  A loopback job
  falls back to its name (anyone who can connect to 127.0.0.1 is this host already)."""
  secret = os.environ.get('SPARTAN_JOB_SECRET')
  if secret:
    return secret.encode()
  addr = os.environ.get('MASTER_ADDR', '127.0.0.1')
  if not _is_loopback(addr):
    raise RendezvousError('rendezvous: MASTER_ADDR=%s is not a loopback address: set SPARTAN_JOB_SECRET to a token only '
                          'the ranks of this job know (the hub executes what an admitted peer sends)' % addr)
  return _job_key(world_size)


def endpoint():
  """(address, first port) of the hub for this environment."""
  env = os.environ
  addr = env.get('MASTER_ADDR', '127.0.0.1')
  if env.get('SPARTAN_RDZV_PORT'):
    return addr, int(env['SPARTAN_RDZV_PORT'])
  port = int(env.get('MASTER_PORT', '29500'))
  if env.get('TORCHELASTIC_USE_AGENT_STORE', '').lower() in ('1', 'true'):
    port += 1                      # MASTER_PORT itself is the launcher's TCPStore
  return addr, port


def _send_msg(sock, obj):
  data = pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL)
  head = struct.pack('>Q', len(data))
  if len(data) < (1 << 16):
    sock.sendall(head + data)          # one segment for the small messages (TCP_NODELAY is on)
  else:
    sock.sendall(head)
    sock.sendall(data)


def _recv_exact(sock, n):
  buf = bytearray(n)
  view, got = memoryview(buf), 0
  while got < n:
    k = sock.recv_into(view[got:], n - got)
    if k == 0:
      raise EOFError('connection closed')
    got += k
  return buf


def _recv_msg(sock):
  (n,) = struct.unpack('>Q', bytes(_recv_exact(sock, 8)))
  return pickle.loads(_recv_exact(sock, n))


_NP_RED = {'ADD': np.add, 'MUL': np.multiply, 'MAX': np.maximum, 'MIN': np.minimum,
           'AND': np.logical_and, 'OR': np.logical_or}


def _reduce_in_rank_order(vals, reducer):
  if reducer not in _NP_RED:
    raise RendezvousError('unknown reducer %r (known: %s)' % (reducer, ', '.join(sorted(_NP_RED))))
  fn = _NP_RED[reducer]
  acc = np.array(vals[0], copy=True)
  for v in vals[1:]:
    acc = fn(acc, v).astype(acc.dtype, copy=False)
  return acc


class _Round(object):
  __slots__ = ('vals', 'results', 'done', 'taken')

  def __init__(self):
    self.vals, self.results, self.done, self.taken = {}, None, False, 0


class Hub(object):
  """The serving side (rank 0): listens, one daemon thread per connection."""

  def __init__(self, world_size, timeout_s):
    self.size = int(world_size)
    self.timeout_s = float(timeout_s)
    self.key = _job_key(world_size)
    self.secret = _job_secret(world_size)
    self.cond = threading.Condition()
    self.rounds = {}                                  # round key -> _Round
    self.mail = collections.defaultdict(collections.deque)   # (src, dst) -> payloads
    self.kv = {}
    self.gone = {}                                    # rank -> reason (connection lost without 'bye')
    self.byes = set()
    self.closing = False
    addr, base = endpoint()
    # the interface MASTER_ADDR names, not every interface of the host
    try:
      host = '127.0.0.1' if _is_loopback(addr) else socket.gethostbyname(addr)
    except OSError as e:
      raise RendezvousError('rendezvous: MASTER_ADDR=%r does not resolve on rank 0: %s' % (addr, e))
    self.sock = None
    last = None
    for port in range(base, base + _PORT_SPAN):
      s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
      s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
      try:
        s.bind((host, port))
      except OSError as e:
        s.close()
        last = e
        continue
      self.sock, self.port = s, port
      break
    if self.sock is None:
      raise RendezvousError('rendezvous: no free port in %d..%d on %r: %s' % (base, base + _PORT_SPAN - 1, host, last))
    self.sock.listen(4 * self.size + 8)
    self.thread = threading.Thread(target=self._accept_loop, name='spartan-rdzv-hub', daemon=True)
    self.thread.start()

  # -- serving ----------------------------------------------------------------------------------------------
  def _accept_loop(self):
    while not self.closing:
      try:
        conn, _ = self.sock.accept()
      except OSError:
        return
      threading.Thread(target=self._serve, args=(conn,), name='spartan-rdzv-conn', daemon=True).start()

  def _serve(self, conn):
    rank, primary, said_bye = None, False, False
    try:
      conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
      conn.settimeout(10.0)
      hello = bytes(_recv_exact(conn, len(_MAGIC) + 32))
      if hello[:len(_MAGIC)] != _MAGIC or hello[len(_MAGIC):] != self.key:
        conn.sendall(_NO)
        return
      nonce = os.urandom(16)
      conn.sendall(_OK + nonce)
      proof = bytes(_recv_exact(conn, 32))
      if not hmac.compare_digest(proof, hmac.new(self.secret, nonce + self.key, hashlib.sha256).digest()):
        conn.sendall(_NO)
        return
      conn.sendall(_OK)
      conn.settimeout(None)
      while True:
        msg = _recv_msg(conn)
        op = msg[0]
        if op == 'hello':
          rank, primary = int(msg[1]), bool(msg[2])
          if not 0 <= rank < self.size:
            _send_msg(conn, ('err', 'rank %d of %d' % (rank, self.size)))
            return
          _send_msg(conn, ('ok', self.size))
        elif op == 'bye':
          said_bye = True
          with self.cond:
            self.byes.add(rank)
            self.cond.notify_all()
          _send_msg(conn, ('ok', None))
          return
        else:
          try:
            _send_msg(conn, ('ok', self._handle(rank, msg)))
          except RendezvousError as e:
            _send_msg(conn, ('err', str(e)))
    except (EOFError, OSError, pickle.UnpicklingError, struct.error):
      pass
    finally:
      try:
        conn.close()
      except OSError:
        pass
      if primary and not said_bye and rank is not None:
        with self.cond:
          self.gone.setdefault(rank, 'rank %d left the job (its connection to the rendezvous hub closed)' % rank)
          self.cond.notify_all()

  def _wait(self, ready, what, ranks_needed):
    """Under self.cond: wait until ready() holds; fails when a rank that is needed has gone or the deadline passes."""
    deadline = time.time() + self.timeout_s
    while not ready():
      for r in ranks_needed():
        if r in self.gone:
          raise RendezvousError('%s: %s' % (what, self.gone[r]))
      left = deadline - time.time()
      if left <= 0:
        raise RendezvousError('%s: timed out after %.0f s waiting for ranks %s' % (what, self.timeout_s, sorted(ranks_needed())))
      self.cond.wait(min(left, 1.0))

  def _handle(self, rank, msg):
    op = msg[0]
    if op == 'set':
      with self.cond:
        self.kv[msg[1]] = msg[2]
      return None
    if op == 'get':
      with self.cond:
        return self.kv.get(msg[1])
    if op == 'del':
      with self.cond:
        self.kv.pop(msg[1], None)
      return None
    if op == 'send':
      with self.cond:
        self.mail[(rank, msg[1])].append(msg[2])
        self.cond.notify_all()
      return None
    if op == 'recv':
      src = msg[1]
      with self.cond:
        box = self.mail[(src, rank)]
        self._wait(lambda: len(box) > 0, 'receive from rank %d' % src, lambda: [src])
        return box.popleft()
    if op == 'coll':
      _, key, kind, payload = msg
      with self.cond:
        rnd = self.rounds.get(key)
        if rnd is None:
          rnd = self.rounds[key] = _Round()
        rnd.vals[rank] = payload
        if len(rnd.vals) == self.size:
          try:
            rnd.results = self._finish(kind, [rnd.vals[r] for r in range(self.size)])
          except Exception as e:       # a bad request (unknown reducer, ragged payloads): every rank of the round hears it
            rnd.results = _Failed('collective %r: %s: %s' % (key, type(e).__name__, e))
          rnd.vals, rnd.done = None, True
          self.cond.notify_all()
        else:
          try:
            self._wait(lambda: rnd.done, 'collective %r' % (key,),
                       lambda: [r for r in range(self.size) if rnd.vals is not None and r not in rnd.vals])
          except RendezvousError:
            # a rank of the round has gone or the wait timed out: `taken` can never reach `size`, so the round (and
            # the payloads it holds) is dropped by the first rank that gives up on it
            self.rounds.pop(key, None)
            raise
        if isinstance(rnd.results, _Failed):
          rnd.taken += 1
          if rnd.taken == self.size:
            self.rounds.pop(key, None)
          raise RendezvousError(rnd.results.why)
        out = rnd.results[rank] if isinstance(rnd.results, _PerRank) else rnd.results
        rnd.taken += 1
        if rnd.taken == self.size:
          del self.rounds[key]
        return out
    raise RendezvousError('unknown request %r' % (op,))

  def _finish(self, kind, vals):
    name = kind[0]
    if name == 'barrier':
      return None
    if name == 'gather':
      return vals
    if name == 'bcast':
      return vals[kind[1]]
    if name == 'allreduce':
      return _reduce_in_rank_order(vals, kind[1])
    if name == 'reduce':
      total = _reduce_in_rank_order(vals, kind[1])
      return _PerRank([total if r == kind[2] else None for r in range(self.size)])
    if name == 'reduce_scatter':
      total = _reduce_in_rank_order(vals, kind[1]).reshape(self.size, -1)
      return _PerRank([np.ascontiguousarray(total[r]) for r in range(self.size)])
    raise RendezvousError('unknown collective %r' % (name,))

  def close(self, wait_s=30.0):
    """Stop serving once every other rank has said goodbye (or has gone, or wait_s passed)."""
    deadline = time.time() + wait_s
    with self.cond:
      while len(self.byes | set(self.gone)) < self.size and time.time() < deadline:
        self.cond.wait(0.2)
    self.closing = True
    try:
      self.sock.close()
    except OSError:
      pass


class _PerRank(list):
  """Result of a round that differs by rank."""


class _Failed(object):
  """Result of a round whose reduction raised: every participant gets the error instead of a value."""

  def __init__(self, why):
    self.why = why


class Client(object):
  """One rank's side.  Every thread gets a connection of its own (the heartbeat's watcher talks to the store while
  the driver thread waits in a round); the first one made is the rank's primary connection -- the one whose loss
  tells the hub that the rank has gone."""

  def __init__(self, rank, world_size, timeout_s, port=None):
    self.rank, self.size = int(rank), int(world_size)
    self.timeout_s = float(timeout_s)
    self.key = _job_key(world_size)
    self.secret = _job_secret(world_size)
    self._tls = threading.local()
    self._lock = threading.Lock()
    self._all = []
    self._have_primary = False
    self._port = port                             # known once a hub answered (rank 0: the port it bound)
    self._seq = collections.defaultdict(int)
    self._conn()                                  # join now: a hub that cannot be reached is reported at start-up

  def _connect(self):
    addr, base = endpoint()
    ports = [self._port] if self._port else list(range(base, base + _PORT_SPAN))
    deadline = time.time() + self.timeout_s
    last = 'no attempt'
    while time.time() < deadline:
      for port in ports:
        s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        s.settimeout(3.0)
        try:
          s.connect((addr, port))
          s.sendall(_MAGIC + self.key)
          answer = bytes(_recv_exact(s, len(_OK)))
          if answer == _OK:
            nonce = bytes(_recv_exact(s, 16))
            s.sendall(hmac.new(self.secret, nonce + self.key, hashlib.sha256).digest())
            if bytes(_recv_exact(s, len(_OK))) == _OK:
              s.settimeout(None)
              s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
              self._port = port
              return s
            last = 'port %d: the hub of this job refused the proof (SPARTAN_JOB_SECRET differs between the ranks?)' % port
          else:
            last = 'port %d belongs to another job' % port
        except (OSError, EOFError) as e:
          last = 'port %d: %s' % (port, e)
        s.close()
      time.sleep(0.05)
    raise RendezvousError('rendezvous: rank %d could not reach the hub at %s:%d..%d within %.0f s (%s)'
                          % (self.rank, addr, base, base + _PORT_SPAN - 1, self.timeout_s, last))

  def _conn(self):
    s = getattr(self._tls, 'sock', None)
    if s is None:
      s = self._connect()
      with self._lock:
        primary = not self._have_primary
        self._have_primary = True
        self._all.append(s)
      _send_msg(s, ('hello', self.rank, primary))
      self._reply(s)
      self._tls.sock = s
    return s

  def _reply(self, s):
    try:
      status, value = _recv_msg(s)
    except (EOFError, OSError) as e:
      raise RendezvousError('rendezvous: rank %d lost the hub (%s) -- rank 0 has stopped' % (self.rank, e))
    if status != 'ok':
      raise RendezvousError('rendezvous: ' + str(value))
    return value

  def _call(self, *msg):
    s = self._conn()
    try:
      _send_msg(s, msg)
    except OSError as e:
      raise RendezvousError('rendezvous: rank %d lost the hub (%s) -- rank 0 has stopped' % (self.rank, e))
    return self._reply(s)

  # -- rounds (driver thread; every rank makes the same calls in the same order) ---------------------------------
  def _round(self, channel, kind, payload):
    self._seq[channel] += 1
    return self._call('coll', (channel, self._seq[channel]), kind, payload)

  def barrier(self):
    self._round('ctl', ('barrier',), None)

  def all_gather_object(self, obj):
    return self._round('ctl', ('gather',), obj)

  def broadcast_object(self, obj, src):
    return self._round('ctl', ('bcast', int(src)), obj if self.rank == src else None)

  def all_reduce(self, array, reducer):
    return self._round('data', ('allreduce', reducer), array)

  def reduce(self, array, root, reducer):
    return self._round('data', ('reduce', reducer, int(root)), array)

  def reduce_scatter(self, array, reducer):
    return self._round('data', ('reduce_scatter', reducer), array)

  # -- mailbox ------------------------------------------------------------------------------------------------
  def send(self, dst, payload):
    self._call('send', int(dst), payload)

  def recv(self, src):
    return self._call('recv', int(src))

  # -- key-value store (any thread) ---------------------------------------------------------------------------
  def set(self, key, value):
    self._call('set', key, value)

  def get(self, key):
    return self._call('get', key)

  def delete(self, key):
    self._call('del', key)

  def close(self):
    """Say goodbye on the primary connection and drop the others."""
    with self._lock:
      socks, self._all = self._all, []
    for i, s in enumerate(socks):
      try:
        if i == 0:
          _send_msg(s, ('bye',))
          _recv_msg(s)
        s.close()
      except (OSError, EOFError):
        pass
    self._tls = threading.local()


def join(rank, world_size, timeout_s=None):
  """Bring this rank into the job: (client, hub or None).  Rank 0 starts the hub first."""
  if timeout_s is None:
    timeout_s = float(os.environ.get('SPARTAN_RDZV_TIMEOUT_S', '1800'))
  hub = Hub(world_size, timeout_s) if rank == 0 else None
  try:
    client = Client(rank, world_size, min(timeout_s, float(os.environ.get('SPARTAN_RDZV_JOIN_S', '300'))),
                    port=hub.port if hub is not None else None)
  except Exception:
    if hub is not None:
      hub.close(0.0)
    raise
  client.timeout_s = timeout_s
  return client, hub
