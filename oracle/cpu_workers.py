"""TEST / BENCH INFRASTRUCTURE -- the CPU baseline of bench.py.  Not part of the product path.

The reference's execution model on host cores: W worker processes, one per
physical core, each pinned to its core and limited to one BLAS / OpenMP thread (spartan/worker.py:40,385-387:
"one worker == one core", `taskset` under use_single_core), every worker holding its own tiles and running the
NumPy tile bodies of the path on them; the parent plays the owner / master side: it merges what the workers push
(Tile.merge: the first update replaces, later ones are added -- spartan/array/tile.pyx:263-268).

Only bench.py (cpu_baseline) imports this module.  The tile bodies are the oracle's restatement of the
reference's mappers: dot_map2_mapper `tiles[0].dot(tiles[1])` (dot.py:195-217), the un-fused NumPy evaluation of
x*x+x (local.py:115-127), `_sum_local` (mathematics.py:126-127), the lreg step (sgd.py:34-39,
linear_regression.py:10-16) and the k-means map2 tile bodies (k_means_.py:52-97).
"""
import os
import subprocess
import sys
import tempfile
import time
from multiprocessing.connection import Client, Listener

SEED = 20150708


def physical_cores():
  """One logical CPU of every physical core this process may run on."""
  allowed = sorted(os.sched_getaffinity(0))
  seen, picks = set(), []
  try:
    for cpu in allowed:
      base = '/sys/devices/system/cpu/cpu%d/topology/' % cpu
      key = (open(base + 'physical_package_id').read().strip(), open(base + 'core_id').read().strip())
      if key not in seen:
        seen.add(key)
        picks.append(cpu)
  except (IOError, OSError):
    picks = allowed
  return picks


def _worker(conn, cpu, index, count):
  try:
    os.sched_setaffinity(0, {cpu})
  except OSError:
    pass
  import numpy as np   # (after the thread limits in the environment took effect)
  rng = np.random.RandomState(SEED + index)
  state = {}
  while True:
    msg = conn.recv()
    op = msg[0]
    if op == 'stop':
      return
    if op == 'make_dot':          # worker `index` owns rows index*n/count.. of A and of B; the K-split needs
      n = msg[1]                  # the column slab A[:, k_w] (fetched from every row tile in the reference)
      kw = n // count
      state['a_slab'] = (rng.rand(n, kw) * 2 - 1).astype(np.float32)
      state['b_rows'] = (rng.rand(kw, n) * 2 - 1).astype(np.float32)
      conn.send(None)
    elif op == 'dot':             # dot_map2_mapper: the M x N partial, pushed to the owner of the one target tile
      part = state['a_slab'].dot(state['b_rows'])
      conn.send(None)
      conn.send_bytes(part)
    elif op == 'dot_compute':     # the same partial, kept in the worker's memory until the owner asks for it
      state['part'] = state['a_slab'].dot(state['b_rows'])
      conn.send(None)
    elif op == 'dot_push':        # ... and copied into a slot of the shared transfer ring (the "send")
      path, offset = msg[1], msg[2]
      part = state.pop('part')
      ring = np.memmap(path, dtype=np.float32, mode='r+', offset=offset, shape=part.shape)
      ring[...] = part
      del ring
      conn.send(None)
    elif op == 'make_tile':
      rows, cols = msg[1], msg[2]
      state['x'] = rng.rand(rows, cols).astype(np.float32)
      conn.send(None)
    elif op == 'map':             # MapExpr x*x+x, evaluated node by node on the tile (one temporary per node)
      x = state['x']
      state['y'] = np.add(np.multiply(x, x), x)
      conn.send(None)
    elif op == 'sum0':            # ReduceExpr sum(axis=0): local partial, combined by the owner
      conn.send(state['x'].sum(axis=0))
    elif op == 'make_lreg':
      rows, dim = msg[1], msg[2]
      state['X'] = rng.rand(rows, dim).astype(np.float32)
      state['yv'] = rng.rand(rows, 1).astype(np.float32)
      conn.send(None)
    elif op == 'lreg':            # yp = dot(X, w); sum(X * (yp - y), axis=0)
      w = msg[1]
      yp = state['X'].dot(w)
      conn.send((state['X'] * (yp - state['yv'])).sum(axis=0))
    elif op == 'make_kmeans':
      rows, dim = msg[1], msg[2]
      state['P'] = rng.rand(rows, dim).astype(np.float32)
      conn.send(None)
    elif op == 'kmeans':          # labels = argmin(cdist), bincount, per-cluster sums
      from scipy.spatial.distance import cdist
      centers = msg[1]
      k = centers.shape[0]
      P = state['P']
      labels = np.argmin(cdist(P, centers), axis=1)
      counts = np.bincount(labels, minlength=k)
      sums = np.zeros((k, P.shape[1]), dtype=P.dtype)
      for c in range(k):
        sums[c] = P[labels == c].sum(axis=0)
      conn.send((counts, sums))
    else:
      conn.send(ValueError(op))


class Workers(object):
  """W pinned single-thread worker processes and the master-side combine."""

  def __init__(self, max_workers=64):
    cpus = physical_cores()[:max_workers]
    self.count = len(cpus)
    # fresh interpreters (the parent holds a HIP context and torch: never fork it, never re-import its main
    # module), started with the thread limits already in their environment; they call back on a local socket
    env = dict(os.environ)
    env.update({'OMP_NUM_THREADS': '1', 'OPENBLAS_NUM_THREADS': '1', 'MKL_NUM_THREADS': '1'})
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    self._dir = tempfile.mkdtemp(prefix='sp_cpu_')
    address = os.path.join(self._dir, 'sock')
    listener = Listener(address, family='AF_UNIX')
    self.procs = [subprocess.Popen([sys.executable, '-m', 'oracle.cpu_workers', address, str(i), str(self.count),
                                    str(cpu)], cwd=root, env=env) for i, cpu in enumerate(cpus)]
    conns = {}
    for _ in cpus:
      c = listener.accept()
      conns[c.recv()] = c
    listener.close()
    self.conns = [conns[i] for i in range(self.count)]

  def all(self, *msg):
    for c in self.conns:
      c.send(msg)
    return [c.recv() for c in self.conns]

  def close(self):
    for c in self.conns:
      try:
        c.send(('stop',))
      except (OSError, BrokenPipeError):
        pass
    for p in self.procs:
      try:
        p.wait(timeout=5)
      except subprocess.TimeoutExpired:
        p.kill()
    import shutil
    shutil.rmtree(self._dir, ignore_errors=True)

  # ---- the timed programs: each returns (seconds, what was computed) -------------------------------------------
  def dot(self, n):
    import numpy as np
    n = n // self.count * self.count
    self.all('make_dot', n)
    target = np.empty((n, n), np.float32)
    scratch = np.empty((n, n), np.float32)
    t0 = time.perf_counter()
    for c in self.conns:
      c.send(('dot',))
    for c in self.conns:
      c.recv()
    t_compute = time.perf_counter() - t0
    for i, c in enumerate(self.conns):           # the owner merges the partials as they arrive
      if i == 0:
        c.recv_bytes_into(target.reshape(-1).view(np.uint8))
      else:
        c.recv_bytes_into(scratch.reshape(-1).view(np.uint8))
        np.add(target, scratch, out=target)
    return time.perf_counter() - t0, t_compute, n

  def dot_shared(self, n, slots=4):
    """The K-split dot with the partials travelling through SHARED MEMORY (a ring of `slots` M x N buffers in a
    memory-mapped file) instead of pickled pipe messages: every worker computes its M x N partial in parallel and
    keeps it; the owner of the one target tile then takes them one after the other -- the worker copies its partial
    into a free slot (the transfer), the owner merges it (first write replaces, later ones add: tile.pyx:263-268)
    while the next worker fills the next slot.  What is timed is the reference's model -- W GEMMs, W M x N
    transfers, W - 1 merges at one owner -- not the cost of pickling through a pipe.
    Returns (seconds end to end, seconds until the last GEMM finished, n)."""
    import numpy as np
    n = n // self.count * self.count
    self.all('make_dot', n)
    path = os.path.join(self._dir, 'ring')
    slot_bytes = n * n * 4
    with open(path, 'wb') as f:
      f.truncate(slot_bytes * slots)
    ring = np.memmap(path, dtype=np.float32, mode='r+', shape=(slots, n, n))
    target = np.empty((n, n), np.float32)
    try:
      t0 = time.perf_counter()
      self.all('dot_compute')
      t_compute = time.perf_counter() - t0
      waiting = list(range(self.count))
      in_flight = []                       # (worker, slot), in the order the pushes were requested
      free = list(range(slots))
      merged = 0
      while merged < self.count:
        while waiting and free:
          w, s_ = waiting.pop(0), free.pop(0)
          self.conns[w].send(('dot_push', path, s_ * slot_bytes))
          in_flight.append((w, s_))
        w, s_ = in_flight.pop(0)
        self.conns[w].recv()
        if merged == 0:
          target[...] = ring[s_]
        else:
          np.add(target, ring[s_], out=target)
        merged += 1
        free.append(s_)
      return time.perf_counter() - t0, t_compute, n
    finally:
      del ring
      try:
        os.remove(path)
      except OSError:
        pass

  def map_and_sum(self, rows_total, cols):
    import numpy as np
    rows = rows_total // self.count
    self.all('make_tile', rows, cols)
    t0 = time.perf_counter()
    self.all('map')
    t_map = time.perf_counter() - t0
    t0 = time.perf_counter()
    parts = self.all('sum0')
    total = parts[0]
    for p in parts[1:]:
      total = np.add(total, p)
    t_sum = time.perf_counter() - t0
    return t_map, t_sum, rows * self.count

  def lreg_steps(self, rows_total, dim, steps=3):
    """`steps` gradient steps on the WHOLE array, row-tiled over the workers (sgd.py:34-39): per step every worker
    runs its tile body, the owner adds the (dim,) partials and steps w.  Returns (seconds per step, rows)."""
    import numpy as np
    rows = rows_total // self.count
    self.all('make_lreg', rows, dim)
    w = np.random.RandomState(SEED).rand(dim, 1).astype(np.float32)
    t0 = time.perf_counter()
    for _ in range(steps):
      parts = self.all('lreg', w)
      grad = parts[0]
      for p in parts[1:]:
        grad = np.add(grad, p)
      w = w - grad.reshape(dim, 1) * 1e-6
    return (time.perf_counter() - t0) / steps, rows * self.count

  def kmeans_iterations(self, rows_total, dim, k, iterations=1):
    """Lloyd iterations (k_means_.py:52-97): labels, counts and per-cluster sums per worker, combined by the owner.
    Returns (seconds per iteration, rows)."""
    import numpy as np
    rows = rows_total // self.count
    self.all('make_kmeans', rows, dim)
    centers = np.random.RandomState(SEED).rand(k, dim)
    t0 = time.perf_counter()
    for _ in range(iterations):
      parts = self.all('kmeans', centers)
      counts, sums = parts[0]
      for c, s in parts[1:]:
        counts = counts + c
        sums = sums + s
      counts[counts == 0] = 1
      centers = sums / counts.reshape(k, 1)
    return (time.perf_counter() - t0) / iterations, rows * self.count


if __name__ == '__main__':
  _address, _index, _count, _cpu = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
  _conn = Client(_address, family='AF_UNIX')
  _conn.send(_index)
  _worker(_conn, _cpu, _index, _count)
