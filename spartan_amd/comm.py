"""The tile data plane between workers: torch.distributed over RCCL/xGMI.

Replaces the reference's ZeroMQ RPC `get`/`update` exchange
(spartan/blob_ctx.py:103-179, spartan/worker.py:172-230, spartan/rpc/zeromq.py)
for tile payloads.  One process per GPU (rank == worker rank); backend "nccl"
(= RCCL on ROCm) for HBM blobs, "gloo" for the CPU tests.

Every rank runs the same driver program (SPMD), so every rank can derive the
complete transfer schedule of an operation from array metadata alone; the
primitives below are therefore collective calls made by all ranks in the same
order:
  exchange        -- grouped point-to-point (ncclSend/ncclRecv batch): `fetch`
                     of remote slabs (all-to-all of A blocks in dot's map2 join)
                     and irregular `update`s
  reduce_scatter  -- `update(np.add)` of full partials into an evenly tiled target
  reduce          -- the same into a one-tile target (dot's default tile_hint)
  all_gather      -- `glom` / replicated fetch of one-tile-per-rank arrays
  broadcast       -- replicated fetch of a single tile
"""
import os

import torch
import torch.distributed as dist

_RED = None


def _red_ops():
  global _RED
  if _RED is None:
    _RED = {'ADD': dist.ReduceOp.SUM, 'MUL': dist.ReduceOp.PRODUCT, 'MAX': dist.ReduceOp.MAX,
            'MIN': dist.ReduceOp.MIN}
  return _RED


class World(object):
  """The set of worker processes (a thin veneer over a torch.distributed group)."""

  def __init__(self, rank=0, size=1, group=None):
    self.rank = rank
    self.size = size
    self.group = group
    self.stats = {'p2p_bytes': 0, 'collective_bytes': 0, 'p2p_msgs': 0, 'collectives': 0}
    self.staged = False   # debug transport: stage device tensors through the host (see from_env)

  @property
  def distributed(self):
    return self.size > 1

  # -- construction -----------------------------------------------------------
  @staticmethod
  def from_env(backend=None):
    """Join the job described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT
    (torch.distributed.run), or return the 1-process world."""
    if dist.is_available() and dist.is_initialized():
      return World(dist.get_rank(), dist.get_world_size(), None)
    ws = int(os.environ.get('WORLD_SIZE', '1'))
    if ws <= 1:
      return World(0, 1, None)
    rank = int(os.environ['RANK'])
    if backend is None:
      # SPARTAN_DIST_BACKEND=gloo is a DEBUG transport: HBM blobs are staged through
      # host memory so that the N>1 code path can be exercised with several ranks
      # sharing one GPU (RCCL refuses two ranks on one device).  Never the default.
      backend = os.environ.get('SPARTAN_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    if backend == 'nccl':
      local = int(os.environ.get('LOCAL_RANK', rank))
      torch.cuda.set_device(local)
      dist.init_process_group(backend, rank=rank, world_size=ws, device_id=torch.device('cuda', local))
    else:
      if torch.cuda.is_available():
        ndev = torch.cuda.device_count()
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', rank)) % ndev)
      dist.init_process_group(backend, rank=rank, world_size=ws)
    w = World(rank, ws, None)
    w.staged = backend != 'nccl'
    return w

  # -- primitives -------------------------------------------------------------
  def barrier(self):
    if self.distributed:
      dist.barrier(group=self.group)

  def _stage(self, t):
    return t.cpu() if (self.staged and t.is_cuda) else t

  def _unstage(self, dst, host):
    if host is not dst:
      dst.copy_(host)

  def exchange(self, sends, recvs):
    """sends: [(dst_rank, tensor)], recvs: [(src_rank, tensor)]; contiguous
    tensors.  All ranks call this with mutually consistent lists (same global
    order).  One grouped launch (ncclGroupStart/End under RCCL)."""
    if not sends and not recvs:
      return
    assert self.distributed, 'exchange() with remote peers in a 1-process world'
    ops = []
    staged = []
    # a single deterministic order on every rank: interleave as listed
    for dst, t in sends:
      assert t.is_contiguous()
      ops.append(dist.P2POp(dist.isend, self._stage(t), dst, group=self.group))
      self.stats['p2p_bytes'] += t.numel() * t.element_size()
      self.stats['p2p_msgs'] += 1
    for src, t in recvs:
      assert t.is_contiguous()
      h = self._stage(t)
      staged.append((t, h))
      ops.append(dist.P2POp(dist.irecv, h, src, group=self.group))
    for req in dist.batch_isend_irecv(ops):
      req.wait()
    for t, h in staged:
      self._unstage(t, h)

  def all_gather(self, out_tensors, tensor):
    self.stats['collectives'] += 1
    self.stats['collective_bytes'] += tensor.numel() * tensor.element_size() * (self.size - 1)
    dist.all_gather(out_tensors, tensor, group=self.group)

  def all_gather_into(self, out, tensor):
    self.stats['collectives'] += 1
    self.stats['collective_bytes'] += tensor.numel() * tensor.element_size() * (self.size - 1)
    if self.staged or not tensor.is_cuda:
      parts = [torch.empty(tensor.shape, dtype=tensor.dtype) for _ in range(self.size)]
      dist.all_gather(parts, self._stage(tensor).contiguous(), group=self.group)
      out.copy_(torch.cat([p.reshape(-1) for p in parts]).view(out.shape))
      return
    dist.all_gather_into_tensor(out, tensor, group=self.group)

  def all_gather_into_async(self, out, tensor):
    """all_gather_into without waiting: returns a handle whose wait() makes the CURRENT STREAM wait
    for the result (the collective runs on RCCL's own stream, so kernels launched meanwhile overlap
    with it).  The staged / CPU debug transports complete immediately and return None."""
    if self.staged or not tensor.is_cuda:
      self.all_gather_into(out, tensor)
      return None
    self.stats['collectives'] += 1
    self.stats['collective_bytes'] += tensor.numel() * tensor.element_size() * (self.size - 1)
    return dist.all_gather_into_tensor(out, tensor, group=self.group, async_op=True)

  def reduce_scatter(self, out, inp, reducer):
    """out[rank chunk] = reduce over ranks of inp (inp = size equal chunks)."""
    self.stats['collectives'] += 1
    self.stats['collective_bytes'] += inp.numel() * inp.element_size() * (self.size - 1) // self.size
    if inp.is_cuda and not self.staged:
      dist.reduce_scatter_tensor(out, inp, op=_red_ops()[reducer], group=self.group)
    else:
      # gloo has no reduce_scatter: all_reduce + take our chunk (CPU tests / debug transport)
      tmp = self._stage(inp).clone()
      dist.all_reduce(tmp, op=_red_ops()[reducer], group=self.group)
      out.copy_(tmp.view(self.size, -1)[self.rank].view_as(out))

  def all_reduce(self, tensor, reducer):
    self.stats['collectives'] += 1
    self.stats['collective_bytes'] += 2 * tensor.numel() * tensor.element_size() * (self.size - 1) // self.size
    h = self._stage(tensor)
    dist.all_reduce(h, op=_red_ops()[reducer], group=self.group)
    self._unstage(tensor, h)

  def reduce(self, tensor, dst, reducer):
    self.stats['collectives'] += 1
    self.stats['collective_bytes'] += tensor.numel() * tensor.element_size()
    h = self._stage(tensor)
    dist.reduce(h, dst, op=_red_ops()[reducer], group=self.group)
    self._unstage(tensor, h)

  def broadcast(self, tensor, src):
    self.stats['collectives'] += 1
    self.stats['collective_bytes'] += tensor.numel() * tensor.element_size()
    h = self._stage(tensor)
    dist.broadcast(h, src, group=self.group)
    self._unstage(tensor, h)

  def broadcast_object(self, obj, src):
    if not self.distributed:
      return obj
    box = [obj if self.rank == src else None]
    dist.broadcast_object_list(box, src=src, group=self.group)
    return box[0]

  def all_gather_object(self, obj):
    if not self.distributed:
      return [obj]
    out = [None] * self.size
    dist.all_gather_object(out, obj, group=self.group)
    return out
