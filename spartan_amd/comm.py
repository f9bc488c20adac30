"""The tile data plane between workers.

Replaces the reference's ZeroMQ RPC `get` / `update` exchange of tile payloads (spartan/blob_ctx.py:103-179,
spartan/worker.py:172-230, spartan/rpc/zeromq.py).  One process per GPU; every rank runs the same driver program
(SPMD), so every rank can derive the complete transfer schedule of an operation from array metadata alone and the
primitives below are collective calls made by all ranks in the same order:

  exchange        grouped point-to-point blocks: `fetch` of remote slabs (the all-to-all of A blocks in dot's
                  map2 join) and irregular `update`s
  reduce_scatter  `update(np.add)` of full partials into an evenly tiled target
  reduce          the same into a one-tile target (dot's default tile_hint)
  all_gather      `glom` / replicated fetch of one-tile-per-rank arrays
  broadcast       replicated fetch of a single tile

Transports:

  RcclTransport   HBM blobs between GPUs: the collective entry points of libspartan_hip.so (sp_comm_*,
                  include/spartan_hip.h), i.e. RCCL over xGMI called straight from the C-ABI.  Blocking calls
                  are enqueued on the compute stream (stream order is the only synchronisation); `async_` calls
                  run on a communication stream of their own, ordered against the compute stream by events, so
                  that kernels launched meanwhile overlap with the transfer.  There is no second GPU data plane:
                  if the RCCL binding does not come up or fails its start-up self-test the job fails, loudly.
  SocketTransport host arrays through the rendezvous hub (rendezvous.py): the NumPy tile backend of the CPU tests
                  and -- device arrays staged through the host -- a debug transport for several ranks sharing one
                  GPU.  Standard library only.
  TorchTransport  the same over a torch.distributed gloo group, for a caller that brought torch.distributed
                  (backend 'gloo'; the CPU tests cover it).

Small host objects (tile metadata, driver-level random draws, the RCCL rendezvous token), barriers and the
heartbeat's store are control plane: `SocketControl` (the rendezvous hub of rank 0; a job started this way never
imports torch, so the process holds ONE HIP runtime -- the one libspartan_hip.so and the RCCL it binds are linked
against) or `TorchControl` when the caller's process already runs an initialised torch.distributed group.
"""
import ctypes as C
import os
import time

import numpy as np


def _dist():
  """torch.distributed: the CONTROL plane (host objects, barriers, rendezvous) and the CPU-test transport.
  Imported on first use, so a single-process job on the HIP backend never loads torch."""
  import torch.distributed as dist
  return dist


def _torch_red(reducer):
  dist = _dist()
  return {'ADD': dist.ReduceOp.SUM, 'MUL': dist.ReduceOp.PRODUCT, 'MAX': dist.ReduceOp.MAX,
          'MIN': dist.ReduceOp.MIN}[reducer]


def rccl_paths():
  """{'lib_path', 'hip_runtime_path', 'own_hip_runtime_path'} of the RCCL data plane as libspartan_hip.so bound it
  (sp_comm_paths): the RCCL file, the HIP runtime that copy calls into, the HIP runtime of the library itself."""
  from . import _hip
  bufs = [C.create_string_buffer(4096) for _ in range(3)]
  _hip.check(_hip.lib().sp_comm_paths(bufs[0], bufs[1], bufs[2], 4096))
  vals = [b.value.decode('utf-8', 'replace') for b in bufs]
  return {'lib_path': vals[0], 'hip_runtime_path': vals[1], 'own_hip_runtime_path': vals[2]}


def mapped_runtimes():
  """Files of the HIP / HSA runtimes and of RCCL mapped into this process (/proc/self/maps), by library:
  {'libamdhip64': [...], 'libhsa-runtime64': [...], 'librccl': [...]} -- a job's process should show one of each."""
  import re
  found = {'libamdhip64': set(), 'libhsa-runtime64': set(), 'librccl': set()}
  try:
    with open('/proc/self/maps') as fh:
      for line in fh:
        m = re.search(r'(/\S*/(libamdhip64|libhsa-runtime64|librccl)[^/\s]*)$', line.strip())
        if m:
          found[m.group(2)].add(os.path.realpath(m.group(1)))
  except OSError:
    pass
  return {k: sorted(v) for k, v in found.items()}


def _need_dense(*tensors):
  """Collectives take whole dense buffers: the RCCL transport hands RCCL a base pointer and a count.  (The host
  transports would copy a strided view and hide the mistake: `World` checks for every transport, so that the CPU
  tests see what a GPU job would do.)"""
  for t in tensors:
    ok = t.is_contiguous() if hasattr(t, 'is_contiguous') else t.flags['C_CONTIGUOUS']
    if not ok:
      raise AssertionError('a collective was handed a strided view of shape %s: make it contiguous first' % (tuple(t.shape),))


def gpu_count():
  """HIP devices visible to libspartan_hip.so (0 when the library is not built or no device answers)."""
  try:
    from . import _hip
    n = C.c_int(0)
    return n.value if _hip.lib().sp_device_count(C.byref(n)) == 0 else 0
  except Exception:
    return 0


class _Done(object):
  """Handle of an asynchronous transfer: wait() makes the CURRENT stream wait for it.  The handle keeps the
  arrays the transfer reads / writes alive: the tile store reuses freed memory in the order of the compute stream,
  so nothing a side stream still works on may be given back before the compute stream has waited for it."""

  def __init__(self, event, keep):
    self.event = event
    self.keep = keep

  def wait(self):
    from . import devarray as D
    D.current_stream().wait_event(self.event)
    self.keep = None

  def __del__(self):
    if self.keep:                      # dropped without wait(): hold the memory until the transfer is over
      try:
        from . import devarray as D
        D.keep_alive_until(self.event, self.keep)
      except Exception:
        pass


class RcclTransport(object):
  """sp_comm_* of libspartan_hip.so on device arrays."""
  name = 'rccl'
  device_native = True

  def __init__(self, world_size, rank, uid):
    from . import _hip
    from . import devarray as D
    self._hip = _hip
    self._D = D
    self.lib = _hip.lib()
    self.size, self.rank = world_size, rank
    handle = C.c_void_p()
    _hip.check(self.lib.sp_comm_init(world_size, rank, uid, C.byref(handle)))
    self.comm = handle
    # asynchronous transfers run here; high priority, so that a collective's few workgroups are placed as soon as
    # compute workgroups retire instead of queueing behind a whole GEMM launch (its peers on the other GPUs wait)
    self.side = D.Stream(high_priority=True)

  @staticmethod
  def unique_id():
    from . import _hip
    buf = C.create_string_buffer(_hip.SP_COMM_UID_BYTES)
    _hip.check(_hip.lib().sp_comm_unique_id(buf, _hip.SP_COMM_UID_BYTES))
    return buf.raw

  def close(self, abort=False):
    if self.comm is not None:
      (self.lib.sp_comm_abort if abort else self.lib.sp_comm_destroy)(self.comm)
      self.comm = None

  # -- helpers
  def _dt(self, t):
    from . import kernels
    return self._hip.sp_dtype(kernels.np_dtype_of(t))

  def _red(self, reducer):
    return self._hip.REDUCER[reducer]

  @staticmethod
  def _p(t):
    return C.c_void_p(t.data_ptr())

  def _launch(self, fn, tensors, async_):
    """Run fn(stream) on the compute stream, or on the side stream behind everything the compute stream has
    been given so far."""
    D = self._D
    cur = D.current_stream()
    if not async_:
      self._hip.check(fn(cur.ptr))
      return None
    self.side.wait_stream(cur)
    self._hip.check(fn(self.side.ptr))
    return _Done(D.Event().record(self.side), list(tensors))

  # -- primitives (contiguous device arrays)
  def exchange(self, sends, recvs, async_=False):
    ns, nr = len(sends), len(recvs)
    sp = (C.c_int32 * max(ns, 1))(*[d for d, _ in sends])
    rp = (C.c_int32 * max(nr, 1))(*[s for s, _ in recvs])
    sptr = self._hip.ptr_array([t.data_ptr() for _, t in sends])
    rptr = self._hip.ptr_array([t.data_ptr() for _, t in recvs])
    sb = self._hip.i64_array([t.nbytes for _, t in sends])
    rb = self._hip.i64_array([t.nbytes for _, t in recvs])
    return self._launch(lambda st: self.lib.sp_comm_all_to_all_blocks(self.comm, ns, sp, sptr, sb, nr, rp, rptr, rb, st),
                        [t for _, t in sends] + [t for _, t in recvs], async_)

  def all_gather_into(self, out, tensor, async_=False):
    _need_dense(out, tensor)
    return self._launch(lambda st: self.lib.sp_comm_all_gather(self.comm, self._p(tensor), self._p(out),
                                                               tensor.size, self._dt(tensor), st),
                        [out, tensor], async_)

  def reduce_scatter(self, out, inp, reducer, async_=False):
    _need_dense(out, inp)
    return self._launch(lambda st: self.lib.sp_comm_reduce_scatter(self.comm, self._p(inp), self._p(out), out.size,
                                                                   self._dt(inp), self._red(reducer), st),
                        [out, inp], async_)

  def all_reduce(self, tensor, reducer):
    _need_dense(tensor)
    self._launch(lambda st: self.lib.sp_comm_all_reduce(self.comm, self._p(tensor), self._p(tensor), tensor.size,
                                                        self._dt(tensor), self._red(reducer), st), [tensor], False)

  def reduce(self, tensor, dst, reducer):
    _need_dense(tensor)
    self._launch(lambda st: self.lib.sp_comm_reduce(self.comm, self._p(tensor), self._p(tensor), tensor.size,
                                                    self._dt(tensor), self._red(reducer), dst, st), [tensor], False)

  def broadcast(self, tensor, src):
    _need_dense(tensor)
    self._launch(lambda st: self.lib.sp_comm_bcast(self.comm, self._p(tensor), tensor.size, self._dt(tensor),
                                                   src, st), [tensor], False)

  # -- start-up self-test
  def self_test(self, timeout_s=60.0):
    """Every primitive once on small buffers, results checked, with a deadline (a transport that hangs or
    miscomputes must be found here, not in the middle of a job).  Returns (ok, message)."""
    D = self._D
    n, r = self.size, self.rank
    stream = D.Stream()
    try:
      words = 1 << 14
      up = D.from_numpy
      a = up(np.full(words, float(r + 1), np.float32))
      parts = up(np.arange(n * words, dtype=np.float32) + r)
      rs = D.empty((words,), np.float32)
      mine = up(np.full(words, r, np.int64))
      ag = D.empty((n * words,), np.int64)
      b = up(np.full(words, float(r), np.float64))
      red = up(np.full(words, float(r + 1), np.float32))
      nxt, prv = (r + 1) % n, (r - 1) % n
      out_ring = up(np.full(words, r, np.int32))
      in_ring = D.empty((words,), np.int32)
      D.synchronize()
      with D.use_stream(stream):
        self.all_reduce(a, 'ADD')
        self.reduce_scatter(rs, parts, 'ADD')
        self.all_gather_into(ag, mine)
        self.broadcast(b, n - 1)
        self.reduce(red, 0, 'MAX')
        if n > 1:
          self.exchange([(nxt, out_ring)], [(prv, in_ring)])
        else:
          from . import kernels
          kernels.stream_copy(in_ring, out_ring)
      deadline = time.time() + timeout_s
      while not stream.query():
        if time.time() > deadline:
          return False, 'RCCL self-test did not complete within %.0f s' % timeout_s
        time.sleep(0.002)
      self._hip.check(self.lib.sp_comm_async_error(self.comm))
      base = np.arange(words, dtype=np.float64)
      checks = [
          ('all_reduce', a.numpy(), np.full(words, n * (n + 1) / 2.0)),
          ('reduce_scatter', rs.numpy(), n * (base + r * words) + n * (n - 1) / 2.0),
          ('all_gather', ag.numpy(), np.repeat(np.arange(n), words)),
          ('broadcast', b.numpy(), np.full(words, float(n - 1))),
          ('exchange', in_ring.numpy(), np.full(words, prv)),
      ]
      if r == 0:
        checks.append(('reduce', red.numpy(), np.full(words, float(n))))
      for name, got, want in checks:
        if not np.array_equal(got.astype(np.float64), want.astype(np.float64)):
          return False, 'RCCL self-test: wrong result from %s' % name
      return True, 'ok'
    except Exception as e:   # HipError and friends: reported to the caller, which stops the job
      return False, 'RCCL self-test failed: %s' % (e,)


class TorchTransport(object):
  """A torch.distributed gloo group.  Payloads are the NumPy backend's CPU tensors (the CPU tests), or -- `staged`
  -- device arrays copied through the host (a debug transport for several ranks sharing one GPU; never the
  transport of a real job, which is RcclTransport or nothing)."""
  device_native = False

  def __init__(self, group, size, rank, staged):
    self.group, self.size, self.rank, self.staged = group, size, rank, staged
    self.name = 'torch-staged' if staged else 'torch'
    self.device_native = False

  def close(self, abort=False):
    pass

  @staticmethod
  def _stage(t):
    """The host tensor gloo moves: a device array is downloaded, a CPU tensor is used as it is."""
    import torch
    if isinstance(t, np.ndarray):
      # the NumPy backend's tile, wrapped without a copy: what gloo receives lands in the array itself
      return torch.from_numpy(t if t.flags['C_CONTIGUOUS'] and t.flags['WRITEABLE'] else np.array(t, order='C'))
    return torch.from_numpy(np.ascontiguousarray(t.numpy()))

  @staticmethod
  def _unstage(dst, host):
    got = host.numpy()
    if isinstance(dst, np.ndarray):
      if not np.shares_memory(dst, got):
        dst[...] = got.reshape(dst.shape)
    else:
      dst.upload(got)

  def exchange(self, sends, recvs, async_=False):
    dist = _dist()
    ops, staged = [], []
    for dst, t in sends:
      ops.append(dist.P2POp(dist.isend, self._stage(t), dst, group=self.group))
    for src, t in recvs:
      h = self._stage(t)
      staged.append((t, h))
      ops.append(dist.P2POp(dist.irecv, h, src, group=self.group))
    for req in dist.batch_isend_irecv(ops):
      req.wait()
    for t, h in staged:
      self._unstage(t, h)
    return None

  def all_gather_into(self, out, tensor, async_=False):
    import torch
    dist = _dist()
    mine = self._stage(tensor).contiguous()
    parts = [torch.empty(mine.shape, dtype=mine.dtype) for _ in range(self.size)]
    dist.all_gather(parts, mine, group=self.group)
    self._unstage(out, torch.cat([p.reshape(-1) for p in parts]).view(tuple(out.shape)))
    return None

  def reduce_scatter(self, out, inp, reducer, async_=False):
    # gloo has no reduce_scatter: all_reduce + take our piece
    tmp = self._stage(inp).clone()
    _dist().all_reduce(tmp, op=_torch_red(reducer), group=self.group)
    self._unstage(out, tmp.view(self.size, -1)[self.rank].reshape(tuple(out.shape)).contiguous())
    return None

  def all_reduce(self, tensor, reducer):
    h = self._stage(tensor)
    _dist().all_reduce(h, op=_torch_red(reducer), group=self.group)
    self._unstage(tensor, h)

  def reduce(self, tensor, dst, reducer):
    h = self._stage(tensor)
    _dist().reduce(h, dst, op=_torch_red(reducer), group=self.group)
    self._unstage(tensor, h)

  def broadcast(self, tensor, src):
    h = self._stage(tensor)
    _dist().broadcast(h, src, group=self.group)
    self._unstage(tensor, h)


class SocketTransport(object):
  """Host arrays through the rendezvous hub: reductions are done by the hub in rank order, blocks between two
  ranks go through its mailbox in the order they were sent.  Payloads are the NumPy backend's arrays (CPU tests)
  or device arrays copied through the host (`staged`: several ranks sharing one GPU).  A debug / test transport --
  never the data plane of a GPU job, which is RcclTransport or nothing."""
  device_native = False

  def __init__(self, client, size, rank, staged):
    self.client, self.size, self.rank, self.staged = client, size, rank, staged
    self.name = 'socket-staged' if staged else 'socket'

  def close(self, abort=False):
    pass

  @staticmethod
  def _host(t):
    return np.ascontiguousarray(t if isinstance(t, np.ndarray) else t.numpy())

  @staticmethod
  def _put(dst, got):
    if isinstance(dst, np.ndarray):
      dst[...] = np.asarray(got).reshape(dst.shape)
    else:
      dst.upload(np.ascontiguousarray(np.asarray(got).reshape(tuple(dst.shape))))

  def exchange(self, sends, recvs, async_=False):
    for dst, t in sends:
      self.client.send(dst, self._host(t))
    for src, t in recvs:
      self._put(t, self.client.recv(src))
    return None

  def all_gather_into(self, out, tensor, async_=False):
    parts = self.client._round('data', ('gather',), self._host(tensor))
    self._put(out, np.concatenate([np.asarray(p).reshape(-1) for p in parts]))
    return None

  def reduce_scatter(self, out, inp, reducer, async_=False):
    self._put(out, self.client.reduce_scatter(self._host(inp), reducer))
    return None

  def all_reduce(self, tensor, reducer):
    self._put(tensor, self.client.all_reduce(self._host(tensor), reducer))

  def reduce(self, tensor, dst, reducer):
    got = self.client.reduce(self._host(tensor), dst, reducer)
    if self.rank == dst:
      self._put(tensor, got)

  def broadcast(self, tensor, src):
    self._put(tensor, self.client.broadcast_object(self._host(tensor) if self.rank == src else None, src))


class SocketControl(object):
  """Control plane over the rendezvous hub (rendezvous.py): standard library only."""
  name = 'socket'

  def __init__(self, rank, size):
    import atexit
    from . import rendezvous
    self.client, self.hub = rendezvous.join(rank, size)
    self.rank, self.size = rank, size
    # a process that ends without World.close() still says goodbye, and rank 0 keeps serving until the others
    # have: its exit must not cut off a reply another rank is waiting for
    atexit.register(self.close)

  def barrier(self):
    self.client.barrier()

  def broadcast_object(self, obj, src):
    return self.client.broadcast_object(obj, src)

  def all_gather_object(self, obj):
    return self.client.all_gather_object(obj)

  def store(self):
    return self.client                      # set / get / delete; get of a key never set returns None

  def close(self):
    if self.client is not None:
      self.client.close()
      self.client = None
    if self.hub is not None:
      self.hub.close()
      self.hub = None


class _TorchStore(object):
  """The process group's rendezvous store (TCPStore); reads of a key that was never set must not block."""

  def __init__(self):
    self._store = _dist().distributed_c10d._get_default_store()

  def set(self, key, value):
    self._store.set(key, value)

  def get(self, key):
    if not self._store.check([key]):
      return None
    return self._store.get(key).decode()

  def delete(self, key):
    try:
      self._store.delete_key(key)
    except Exception:
      pass


class TorchControl(object):
  """Control plane of a caller that runs torch.distributed: its default (gloo) group."""
  name = 'torch'

  def __init__(self, group=None):
    self.group = group

  def barrier(self):
    _dist().barrier(group=self.group)

  def broadcast_object(self, obj, src):
    box = [obj]
    _dist().broadcast_object_list(box, src=src, group=self.group)
    return box[0]

  def all_gather_object(self, obj):
    dist = _dist()
    out = [None] * dist.get_world_size(self.group)
    dist.all_gather_object(out, obj, group=self.group)
    return out

  def store(self):
    return _TorchStore()

  def close(self):
    pass


class World(object):
  """The set of worker processes: a transport for tile payloads, a control plane for host objects and barriers, and
  counters of what crossed ranks (the tests assert on them)."""

  def __init__(self, rank=0, size=1, group=None, transport=None, control=None):
    self.rank = rank
    self.size = size
    self.group = group
    self.control = control                  # SocketControl / TorchControl (None in a 1-process world)
    self.transport = transport
    if size > 1 and control is None:
      self.control = TorchControl(group)    # a caller that built the world around its own torch.distributed group
    if transport is None and size > 1:
      self.transport = TorchTransport(group, size, rank, staged=False)
    self.stats = {'p2p_bytes': 0, 'collective_bytes': 0, 'p2p_msgs': 0, 'collectives': 0, 'sparse_blocks': 0}
    self.note = ''                          # how the transport was chosen (bench.py prints it)

  @property
  def distributed(self):
    return self.size > 1

  # `staged`: debug transport for several ranks sharing one GPU (tests/mp_worker.py sets it)
  @property
  def staged(self):
    return bool(getattr(self.transport, 'staged', False))

  @staged.setter
  def staged(self, value):
    if isinstance(self.transport, (TorchTransport, SocketTransport)):
      kind = 'torch' if isinstance(self.transport, TorchTransport) else 'socket'
      self.transport.staged = bool(value)
      self.transport.name = kind + '-staged' if value else kind

  # -- construction -----------------------------------------------------------
  @staticmethod
  def from_env(backend=None):
    """Join the job described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (what torch.distributed.run or
    bench.py's own launcher set), or return the 1-process world.  backend:

      'rccl'    sp_comm_* of the C-ABI: the data plane of a GPU job; if it does not come up or fails its self-test
                the job STOPS -- there is no second GPU transport to fall back to.  Control plane: the rendezvous
                hub (no torch in the process).
      'socket'  host arrays through the rendezvous hub: CPU tile backend, or device arrays staged through the host
                -- a debug transport for ranks sharing one GPU.  No torch in the process.
      'gloo'    the same over torch.distributed (imports torch; for callers that live in a torch job).

    Default: 'rccl' with GPUs; without, 'gloo' in a process that already runs torch.distributed, else 'socket'.
    SPARTAN_DIST_BACKEND overrides.  A caller whose process already
    runs an initialised torch.distributed group keeps it as the control plane."""
    ws = int(os.environ.get('WORLD_SIZE', '1'))
    dist = _initialized_torch_dist()
    if dist is not None:
      if dist.get_world_size() <= 1:
        return World(dist.get_rank(), dist.get_world_size(), None)
      return World._join(backend, dist.get_world_size(), dist.get_rank())
    if ws <= 1:
      return World(0, 1, None)
    return World._join(backend, ws, int(os.environ['RANK']))

  @staticmethod
  def _join(backend, ws, rank):
    gpus = gpu_count()
    # (the default follows the control plane: a caller that brought an initialised torch.distributed group keeps it,
    # and on a box without GPUs its data plane is gloo over that group, not the hub's sockets)
    backend = backend or os.environ.get('SPARTAN_DIST_BACKEND') or (
        'rccl' if gpus else ('gloo' if _initialized_torch_dist() is not None else 'socket'))
    if backend not in ('rccl', 'socket', 'gloo'):
      raise ValueError("unknown data-plane backend %r (known: 'rccl', 'socket', 'gloo')" % backend)
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    local = int(os.environ.get('LOCAL_RANK', rank))
    if gpus:
      from . import _hip
      _hip.check(_hip.lib().sp_set_device(local % gpus))
    if backend == 'gloo' or _initialized_torch_dist() is not None:
      dist = _dist()
      if not dist.is_initialized():
        dist.init_process_group('gloo', rank=rank, world_size=ws)
      control = TorchControl(None)
    else:
      control = SocketControl(rank, ws)
    if backend == 'gloo':
      w = World(rank, ws, None, TorchTransport(None, ws, rank, staged=bool(gpus)), control=control)
      w.note = 'gloo'
      return w
    if backend == 'socket':
      if not isinstance(control, SocketControl):
        raise RuntimeError("backend 'socket' in a process that runs torch.distributed: use 'gloo' there")
      w = World(rank, ws, None, SocketTransport(control.client, ws, rank, staged=bool(gpus)), control=control)
      w.note = 'socket'
      return w
    transport, note = _try_rccl(ws, rank, control)
    if transport is None:
      control.close()
      raise RuntimeError('the RCCL data plane (sp_comm_* of libspartan_hip.so) did not come up: %s.  There is no '
                         'fallback transport for tiles in HBM; set SPARTAN_DIST_BACKEND=socket only to debug on '
                         'one GPU.' % note)
    w = World(rank, ws, None, transport, control=control)
    w.note = note
    return w

  def close(self):
    if self.transport is not None:
      self.transport.close()
      self.transport = None
    if self.control is not None:
      self.control.close()                  # (rank 0 keeps its hub up until the other ranks have said goodbye)
      self.control = None

  # -- primitives -------------------------------------------------------------
  def barrier(self):
    if self.distributed:
      self.control.barrier()

  def exchange(self, sends, recvs):
    """sends: [(dst_rank, tensor)], recvs: [(src_rank, tensor)]; contiguous tensors.  All ranks call this with
    mutually consistent lists (blocks between two ranks in the same order on both sides).  One grouped launch."""
    if not sends and not recvs:
      return
    assert self.distributed, 'exchange() with remote peers in a 1-process world'
    for _, t in list(sends) + list(recvs):
      assert t.is_contiguous() if hasattr(t, 'is_contiguous') else t.flags['C_CONTIGUOUS']
    for _, t in sends:
      self.stats['p2p_bytes'] += t.nbytes
      self.stats['p2p_msgs'] += 1
    self.transport.exchange(sends, recvs)

  def exchange_async(self, sends, recvs):
    """exchange() without waiting: returns a handle whose wait() makes the current stream wait (None when the
    transport completed it already)."""
    if not sends and not recvs:
      return None
    for _, t in sends:
      self.stats['p2p_bytes'] += t.nbytes
      self.stats['p2p_msgs'] += 1
    return self.transport.exchange(sends, recvs, async_=True)

  def _count(self, nbytes):
    self.stats['collectives'] += 1
    self.stats['collective_bytes'] += int(nbytes)

  def all_gather_into(self, out, tensor):
    _need_dense(out, tensor)
    self._count(tensor.nbytes * (self.size - 1))
    self.transport.all_gather_into(out, tensor)

  def all_gather_into_async(self, out, tensor):
    """all_gather_into without waiting: returns a handle whose wait() makes the CURRENT STREAM wait for the
    result (the collective runs on a communication stream, so kernels launched meanwhile overlap with it).
    Host-side transports complete immediately and return None."""
    _need_dense(out, tensor)
    self._count(tensor.nbytes * (self.size - 1))
    return self.transport.all_gather_into(out, tensor, async_=True)

  def reduce_scatter(self, out, inp, reducer):
    """out[rank piece] = reduce over ranks of inp (inp = size equal pieces)."""
    _need_dense(out, inp)
    self._count(inp.nbytes * (self.size - 1) // self.size)
    self.transport.reduce_scatter(out, inp, reducer)

  def reduce_scatter_async(self, out, inp, reducer):
    _need_dense(out, inp)
    self._count(inp.nbytes * (self.size - 1) // self.size)
    return self.transport.reduce_scatter(out, inp, reducer, async_=True)

  def all_reduce(self, tensor, reducer):
    _need_dense(tensor)
    self._count(2 * tensor.nbytes * (self.size - 1) // self.size)
    self.transport.all_reduce(tensor, reducer)

  def reduce(self, tensor, dst, reducer):
    _need_dense(tensor)
    self._count(tensor.nbytes)
    self.transport.reduce(tensor, dst, reducer)

  def broadcast(self, tensor, src):
    _need_dense(tensor)
    self._count(tensor.nbytes)
    self.transport.broadcast(tensor, src)

  # -- host objects (control plane) ----------------------------------------------
  def broadcast_object(self, obj, src):
    if not self.distributed:
      return obj
    return self.control.broadcast_object(obj if self.rank == src else None, src)

  def all_gather_object(self, obj):
    if not self.distributed:
      return [obj]
    return self.control.all_gather_object(obj)

  def store(self):
    """The job's key-value store (the heartbeat's counters and verdicts): set / get / delete, reads never block."""
    return self.control.store()


def _initialized_torch_dist():
  """torch.distributed if the CALLER imported and initialised it; never imports torch."""
  import sys
  dist = sys.modules.get('torch.distributed')
  try:
    if dist is not None and dist.is_available() and dist.is_initialized():
      return dist
  except Exception:
    pass
  return None


def _try_rccl(ws, rank, control):
  """Bring up the direct RCCL transport and make it prove itself; (transport, note) or (None, why not)."""
  why = ''
  transport = None
  try:
    from . import _hip
    if not _hip.lib().sp_comm_available():
      why = _hip.lib().sp_last_error().decode('utf-8', 'replace')
  except Exception as e:
    why = str(e)
  def _agree(ok):
    return all(control.all_gather_object(bool(ok)))

  if not _agree(not why):
    return None, 'sp_comm unavailable: %s' % (why or 'on another rank')
  token = None
  if rank == 0:
    try:
      token = ('uid', RcclTransport.unique_id())
    except Exception as e:                 # reported to every rank: all of them stop with the reason
      token = ('error', 'sp_comm_unique_id on rank 0: %s' % (e,))
  kind, uid = control.broadcast_object(token, 0)
  if kind == 'error':
    return None, uid
  try:
    transport = RcclTransport(ws, rank, uid)
  except Exception as e:
    why = 'sp_comm_init: %s' % (e,)
  if not _agree(transport is not None):
    if transport is not None:
      transport.close(abort=True)
    return None, why or 'sp_comm_init failed on another rank'
  ok, msg = transport.self_test(float(os.environ.get('SPARTAN_COMM_SELFTEST_S', '60')))
  if not _agree(ok):
    transport.close(abort=True)
    return None, msg if not ok else 'self-test failed on another rank'
  return transport, 'sp_comm_* over RCCL (self-test passed)'
