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

Two transports carry them:

  RcclTransport   HBM blobs between GPUs: the collective entry points of libspartan_hip.so (sp_comm_*,
                  include/spartan_hip.h), i.e. RCCL over xGMI called straight from the C-ABI.  Blocking calls
                  are enqueued on the compute stream (stream order is the only synchronisation); `async_` calls
                  run on a communication stream of their own, ordered against the compute stream by events, so
                  that kernels launched meanwhile overlap with the transfer.
  TorchTransport  torch.distributed process groups: gloo for the CPU tests (NumPy tile backend), gloo with
                  host staging as a debug transport for several ranks sharing one GPU, and ProcessGroupNCCL as
                  the fallback when the direct binding does not pass its start-up self-test.

Small host objects (tile metadata, driver-level random draws, the RCCL rendezvous token) always travel over a
gloo group: that is control plane, not tile data.
"""
import ctypes as C
import os
import time

import numpy as np
import torch
import torch.distributed as dist


def _torch_red(reducer):
  return {'ADD': dist.ReduceOp.SUM, 'MUL': dist.ReduceOp.PRODUCT, 'MAX': dist.ReduceOp.MAX,
          'MIN': dist.ReduceOp.MIN}[reducer]


class _Done(object):
  """Handle of an asynchronous transfer: wait() makes the CURRENT stream wait for it."""

  def __init__(self, event, keep):
    self.event = event
    self.keep = keep          # tensors the transfer reads / writes

  def wait(self):
    torch.cuda.current_stream().wait_event(self.event)
    self.keep = None


class RcclTransport(object):
  """sp_comm_* of libspartan_hip.so on device tensors."""
  name = 'rccl'
  device_native = True

  def __init__(self, world_size, rank, uid):
    from . import _hip
    self._hip = _hip
    self.lib = _hip.lib()
    self.size, self.rank = world_size, rank
    handle = C.c_void_p()
    _hip.check(self.lib.sp_comm_init(world_size, rank, uid, C.byref(handle)))
    self.comm = handle
    # asynchronous transfers run here; high priority, so that a collective's few workgroups are placed as soon as
    # compute workgroups retire instead of queueing behind a whole GEMM launch (its peers on the other GPUs wait)
    self.side = torch.cuda.Stream(priority=-1)

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
    cur = torch.cuda.current_stream()
    if not async_:
      self._hip.check(fn(C.c_void_p(cur.cuda_stream)))
      return None
    self.side.wait_stream(cur)
    self._hip.check(fn(C.c_void_p(self.side.cuda_stream)))
    for t in tensors:
      t.record_stream(self.side)             # the caching allocator must not hand the memory out early
    done = torch.cuda.Event()
    done.record(self.side)
    return _Done(done, tensors)

  # -- primitives (contiguous device tensors)
  def exchange(self, sends, recvs, async_=False):
    ns, nr = len(sends), len(recvs)
    sp = (C.c_int32 * max(ns, 1))(*[d for d, _ in sends])
    rp = (C.c_int32 * max(nr, 1))(*[s for s, _ in recvs])
    sptr = self._hip.ptr_array([t.data_ptr() for _, t in sends])
    rptr = self._hip.ptr_array([t.data_ptr() for _, t in recvs])
    sb = self._hip.i64_array([t.numel() * t.element_size() for _, t in sends])
    rb = self._hip.i64_array([t.numel() * t.element_size() for _, t in recvs])
    return self._launch(lambda st: self.lib.sp_comm_all_to_all_blocks(self.comm, ns, sp, sptr, sb, nr, rp, rptr, rb, st),
                        [t for _, t in sends] + [t for _, t in recvs], async_)

  def all_gather_into(self, out, tensor, async_=False):
    return self._launch(lambda st: self.lib.sp_comm_all_gather(self.comm, self._p(tensor), self._p(out),
                                                               tensor.numel(), self._dt(tensor), st),
                        [out, tensor], async_)

  def reduce_scatter(self, out, inp, reducer, async_=False):
    return self._launch(lambda st: self.lib.sp_comm_reduce_scatter(self.comm, self._p(inp), self._p(out), out.numel(),
                                                                   self._dt(inp), self._red(reducer), st),
                        [out, inp], async_)

  def all_reduce(self, tensor, reducer):
    self._launch(lambda st: self.lib.sp_comm_all_reduce(self.comm, self._p(tensor), self._p(tensor), tensor.numel(),
                                                        self._dt(tensor), self._red(reducer), st), [tensor], False)

  def reduce(self, tensor, dst, reducer):
    self._launch(lambda st: self.lib.sp_comm_reduce(self.comm, self._p(tensor), self._p(tensor), tensor.numel(),
                                                    self._dt(tensor), self._red(reducer), dst, st), [tensor], False)

  def broadcast(self, tensor, src):
    self._launch(lambda st: self.lib.sp_comm_bcast(self.comm, self._p(tensor), tensor.numel(), self._dt(tensor),
                                                   src, st), [tensor], False)

  # -- start-up self-test
  def self_test(self, timeout_s=60.0):
    """Every primitive once on small buffers, results checked, with a deadline (a transport that hangs or
    miscomputes must be found here, not in the middle of a job).  Returns (ok, message)."""
    n, r = self.size, self.rank
    dev = torch.device('cuda', torch.cuda.current_device())
    stream = torch.cuda.Stream()
    try:
      with torch.cuda.stream(stream):
        words = 1 << 14
        a = torch.full((words,), float(r + 1), dtype=torch.float32, device=dev)
        self.all_reduce(a, 'ADD')
        parts = torch.arange(n * words, dtype=torch.float32, device=dev) + r
        rs = torch.empty(words, dtype=torch.float32, device=dev)
        self.reduce_scatter(rs, parts, 'ADD')
        mine = torch.full((words,), r, dtype=torch.int64, device=dev)
        ag = torch.empty(n * words, dtype=torch.int64, device=dev)
        self.all_gather_into(ag, mine)
        b = torch.full((words,), float(r), dtype=torch.float64, device=dev)
        self.broadcast(b, n - 1)
        red = torch.full((words,), float(r + 1), dtype=torch.float32, device=dev)
        self.reduce(red, 0, 'MAX')
        nxt, prv = (r + 1) % n, (r - 1) % n
        out_ring = torch.full((words,), r, dtype=torch.int32, device=dev)
        in_ring = torch.empty(words, dtype=torch.int32, device=dev)
        if n > 1:
          self.exchange([(nxt, out_ring)], [(prv, in_ring)])
        else:
          in_ring.copy_(out_ring)
      deadline = time.time() + timeout_s
      done = C.c_int32(0)
      while True:
        self._hip.check(self.lib.sp_stream_query(C.c_void_p(stream.cuda_stream), C.byref(done)))
        if done.value:
          break
        if time.time() > deadline:
          return False, 'RCCL self-test did not complete within %.0f s' % timeout_s
        time.sleep(0.002)
      self._hip.check(self.lib.sp_comm_async_error(self.comm))
      base = np.arange(words, dtype=np.float64)
      checks = [
          ('all_reduce', a.cpu().numpy(), np.full(words, n * (n + 1) / 2.0)),
          ('reduce_scatter', rs.cpu().numpy(), n * (base + r * words) + n * (n - 1) / 2.0),
          ('all_gather', ag.cpu().numpy(), np.repeat(np.arange(n), words)),
          ('broadcast', b.cpu().numpy(), np.full(words, float(n - 1))),
          ('exchange', in_ring.cpu().numpy(), np.full(words, prv)),
      ]
      if r == 0:
        checks.append(('reduce', red.cpu().numpy(), np.full(words, float(n))))
      for name, got, want in checks:
        if not np.array_equal(got.astype(np.float64), want.astype(np.float64)):
          return False, 'RCCL self-test: wrong result from %s' % name
      return True, 'ok'
    except Exception as e:   # HipError and friends: reported, the caller falls back
      return False, 'RCCL self-test failed: %s' % (e,)


class TorchTransport(object):
  """torch.distributed process group (gloo: CPU tensors, or device tensors staged through the host; nccl)."""
  device_native = False

  def __init__(self, group, size, rank, staged):
    self.group, self.size, self.rank, self.staged = group, size, rank, staged
    self.name = 'torch-staged' if staged else 'torch'
    self.device_native = not staged

  def close(self, abort=False):
    pass

  def _stage(self, t):
    return t.cpu() if (self.staged and t.is_cuda) else t

  @staticmethod
  def _unstage(dst, host):
    if host is not dst:
      dst.copy_(host)

  def exchange(self, sends, recvs, async_=False):
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
    if self.staged or not tensor.is_cuda:
      parts = [torch.empty(tensor.shape, dtype=tensor.dtype) for _ in range(self.size)]
      dist.all_gather(parts, self._stage(tensor).contiguous(), group=self.group)
      out.copy_(torch.cat([p.reshape(-1) for p in parts]).view(out.shape))
      return None
    return dist.all_gather_into_tensor(out, tensor, group=self.group, async_op=True) if async_ else \
        dist.all_gather_into_tensor(out, tensor, group=self.group)

  def reduce_scatter(self, out, inp, reducer, async_=False):
    if inp.is_cuda and not self.staged:
      return dist.reduce_scatter_tensor(out, inp, op=_torch_red(reducer), group=self.group, async_op=True) if async_ \
          else dist.reduce_scatter_tensor(out, inp, op=_torch_red(reducer), group=self.group)
    # gloo has no reduce_scatter: all_reduce + take our piece (CPU tests / debug transport)
    tmp = self._stage(inp).clone()
    dist.all_reduce(tmp, op=_torch_red(reducer), group=self.group)
    out.copy_(tmp.view(self.size, -1)[self.rank].view_as(out))
    return None

  def all_reduce(self, tensor, reducer):
    h = self._stage(tensor)
    dist.all_reduce(h, op=_torch_red(reducer), group=self.group)
    self._unstage(tensor, h)

  def reduce(self, tensor, dst, reducer):
    h = self._stage(tensor)
    dist.reduce(h, dst, op=_torch_red(reducer), group=self.group)
    self._unstage(tensor, h)

  def broadcast(self, tensor, src):
    h = self._stage(tensor)
    dist.broadcast(h, src, group=self.group)
    self._unstage(tensor, h)


class World(object):
  """The set of worker processes: a transport for tile payloads, a gloo group for host objects, and counters
  of what crossed ranks (the tests assert on them)."""

  def __init__(self, rank=0, size=1, group=None, transport=None, control=None):
    self.rank = rank
    self.size = size
    self.group = group
    self.control = control                  # gloo group for objects / barriers (None: the default group)
    self.transport = transport
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
    if isinstance(self.transport, TorchTransport):
      self.transport.staged = bool(value)
      self.transport.device_native = not value
      self.transport.name = 'torch-staged' if value else 'torch'

  # -- construction -----------------------------------------------------------
  @staticmethod
  def from_env(backend=None):
    """Join the job described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torch.distributed.run), or
    return the 1-process world.  backend: 'rccl' (sp_comm_* of the C-ABI; the default on GPUs, falls back to
    'nccl' if its self-test fails), 'nccl' (torch's ProcessGroupNCCL), 'gloo' (CPU tensors; device tensors are
    staged through the host -- a debug transport for ranks sharing one GPU).  SPARTAN_DIST_BACKEND overrides."""
    ws = int(os.environ.get('WORLD_SIZE', '1'))
    if dist.is_available() and dist.is_initialized():
      ws = dist.get_world_size()
      rank = dist.get_rank()
      if ws <= 1:
        return World(rank, ws, None)
    elif ws <= 1:
      return World(0, 1, None)
    else:
      rank = int(os.environ['RANK'])
    backend = backend or os.environ.get('SPARTAN_DIST_BACKEND') or ('rccl' if torch.cuda.is_available() else 'gloo')
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    local = int(os.environ.get('LOCAL_RANK', rank))
    if torch.cuda.is_available():
      torch.cuda.set_device(local % torch.cuda.device_count())
    if not dist.is_initialized():
      # the default group is gloo: host objects, barriers, and the CPU / staged data plane
      dist.init_process_group('gloo', rank=rank, world_size=ws)
    if backend == 'gloo':
      w = World(rank, ws, None, TorchTransport(None, ws, rank, staged=torch.cuda.is_available()))
      w.note = 'gloo'
      return w
    if backend == 'rccl':
      transport, note = _try_rccl(ws, rank)
      if transport is not None:
        w = World(rank, ws, None, transport)
        w.note = note
        return w
    else:
      note = 'requested'
    group = dist.new_group(backend='nccl', device_id=torch.device('cuda', torch.cuda.current_device()))
    w = World(rank, ws, None, TorchTransport(group, ws, rank, staged=False))
    w.note = 'torch ProcessGroupNCCL (%s)' % note
    return w

  def close(self):
    if self.transport is not None:
      self.transport.close()
      self.transport = None

  # -- primitives -------------------------------------------------------------
  def barrier(self):
    if self.distributed:
      dist.barrier(group=self.control)

  def exchange(self, sends, recvs):
    """sends: [(dst_rank, tensor)], recvs: [(src_rank, tensor)]; contiguous tensors.  All ranks call this with
    mutually consistent lists (blocks between two ranks in the same order on both sides).  One grouped launch."""
    if not sends and not recvs:
      return
    assert self.distributed, 'exchange() with remote peers in a 1-process world'
    for _, t in list(sends) + list(recvs):
      assert t.is_contiguous()
    for _, t in sends:
      self.stats['p2p_bytes'] += t.numel() * t.element_size()
      self.stats['p2p_msgs'] += 1
    self.transport.exchange(sends, recvs)

  def exchange_async(self, sends, recvs):
    """exchange() without waiting: returns a handle whose wait() makes the current stream wait (None when the
    transport completed it already)."""
    if not sends and not recvs:
      return None
    for _, t in sends:
      self.stats['p2p_bytes'] += t.numel() * t.element_size()
      self.stats['p2p_msgs'] += 1
    return self.transport.exchange(sends, recvs, async_=True)

  def _count(self, nbytes):
    self.stats['collectives'] += 1
    self.stats['collective_bytes'] += int(nbytes)

  def all_gather_into(self, out, tensor):
    self._count(tensor.numel() * tensor.element_size() * (self.size - 1))
    self.transport.all_gather_into(out, tensor)

  def all_gather_into_async(self, out, tensor):
    """all_gather_into without waiting: returns a handle whose wait() makes the CURRENT STREAM wait for the
    result (the collective runs on a communication stream, so kernels launched meanwhile overlap with it).
    Host-side transports complete immediately and return None."""
    self._count(tensor.numel() * tensor.element_size() * (self.size - 1))
    return self.transport.all_gather_into(out, tensor, async_=True)

  def reduce_scatter(self, out, inp, reducer):
    """out[rank piece] = reduce over ranks of inp (inp = size equal pieces)."""
    self._count(inp.numel() * inp.element_size() * (self.size - 1) // self.size)
    self.transport.reduce_scatter(out, inp, reducer)

  def reduce_scatter_async(self, out, inp, reducer):
    self._count(inp.numel() * inp.element_size() * (self.size - 1) // self.size)
    return self.transport.reduce_scatter(out, inp, reducer, async_=True)

  def all_reduce(self, tensor, reducer):
    self._count(2 * tensor.numel() * tensor.element_size() * (self.size - 1) // self.size)
    self.transport.all_reduce(tensor, reducer)

  def reduce(self, tensor, dst, reducer):
    self._count(tensor.numel() * tensor.element_size())
    self.transport.reduce(tensor, dst, reducer)

  def broadcast(self, tensor, src):
    self._count(tensor.numel() * tensor.element_size())
    self.transport.broadcast(tensor, src)

  # -- host objects (control plane) ----------------------------------------------
  def broadcast_object(self, obj, src):
    if not self.distributed:
      return obj
    box = [obj if self.rank == src else None]
    dist.broadcast_object_list(box, src=src, group=self.control)
    return box[0]

  def all_gather_object(self, obj):
    if not self.distributed:
      return [obj]
    out = [None] * self.size
    dist.all_gather_object(out, obj, group=self.control)
    return out


def _agree(ok):
  """True iff every rank says ok (gloo)."""
  flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
  dist.all_reduce(flag, op=dist.ReduceOp.MIN)
  return bool(flag.item())


def _try_rccl(ws, rank):
  """Bring up the direct RCCL transport and make it prove itself; (transport, note) or (None, why not)."""
  why = ''
  transport = None
  try:
    from . import _hip
    if not _hip.lib().sp_comm_available():
      why = _hip.lib().sp_last_error().decode('utf-8', 'replace')
  except Exception as e:
    why = str(e)
  if not _agree(not why):
    return None, 'sp_comm unavailable: %s' % (why or 'on another rank')
  box = [RcclTransport.unique_id() if rank == 0 else None]
  dist.broadcast_object_list(box, src=0)
  try:
    transport = RcclTransport(ws, rank, box[0])
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
