#!/usr/bin/env python
"""Headline benchmark: BASELINE.json's metric
    "spartan.dot TFLOP/s + map/reduce HBM GB/s at 1/2/4/8 MI355X".

One "step" = one `spartan.dot(A, B).force()` through the whole tile path (lazy DAG -> per-tile fp32 MFMA GEMM
launches -> merge into the target), operands already resident in HBM:

  --gpus 1 : BASELINE configs[1] -- dot 8192 x 8192 x 8192 fp32, one tile.  The same line carries the north-star
             shape on one GPU (`northstar_32768`: dot 32768^3, timed with HIP events around whole `.force()` calls).
  --gpus N : the north star, STRONG scaling: dot 32768 x 32768 x 32768 fp32 with A, B and the result row-tiled one
             tile per GPU (`tile_hint=(M/N, N)`).  rows <= cols, so this is the reference's K-split map2 join
             (dot.py:286-290): the all-to-all of A's column slabs, p GEMMs per column chunk of the partial, and
             one reduce-scatter per chunk overlapped with the next chunk's GEMMs (spartan_amd/expr/dot.ksplit_plan).
             `value` is whole-job TFLOP/s = 2 * 32768^3 * steps / max-over-ranks wall time; `dot_breakdown`
             gives kernel-only time and the bytes each GPU moved per step against the xGMI link rates.

The line also carries the roofline of the dominant kernel (sp_gemm_kernel, MFMA-bound; durations from HIP events
on the launch stream), the fused-map / reduce HBM rates, the k-means and sparse tile kernels, and the CPU
baseline: the oracle's tile bodies on W = min(physical cores, 64) pinned one-thread worker processes.
"""
import argparse
import os as _os
_os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # dmabuf IPC between the per-GPU processes (RCCL)
_os.environ.setdefault('NCCL_DEBUG_FILE', '/dev/stderr')     # (see _claim_stdout)
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import spartan_amd as sp  # noqa: E402
from spartan_amd import _hip, kernels  # noqa: E402

MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
HBM_PEAK_GBPS = 8000.0         # MI355X_MICROARCH.md: HBM3E spec peak
SEED = 20150708
SETUP_LAUNCHES = 8              # untimed set-up steps before the W warm-up steps (see main)
NORTH_STAR = 32768
XGMI_LINK_GBPS = 153.0          # per link, 7 links per GPU (MI355X_MICROARCH.md / task brief)


def device_uniform(ex, lo, hi, seed):
  g = torch.Generator(device='cuda')
  g.manual_seed(seed + 1000003 * ex.ul[0])
  t = torch.rand(ex.shape, dtype=torch.float32, device='cuda', generator=g)
  return t * (hi - lo) + lo


def time_steps(ctx, step, steps, warmup):
  for _ in range(warmup):
    step()
  ctx.world.barrier()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(steps):
    step()
  torch.cuda.synchronize()
  ctx.world.barrier()
  dt = time.perf_counter() - t0
  if ctx.world.distributed:
    t = torch.tensor([dt], dtype=torch.float64, device='cuda')
    ctx.world.all_reduce(t, 'MAX')
    dt = float(t.item())
  return dt


def event_time(fn, iters, warmup=4):
  for _ in range(warmup):
    fn()
  # programs outside the prebuilt kernel library are specialised at run time on a
  # background thread (include/spartan_hip.h sp_jit_*): let the warm-up's requests land
  _hip.lib().sp_jit_wait()
  fn()
  torch.cuda.synchronize()
  e0, e1 = kernels.Event(), kernels.Event()
  e0.record()
  for _ in range(iters):
    fn()
  e1.record()
  e1.synchronize()
  return e0.elapsed_ms(e1) / iters


def hbm_section(ctx):
  """Fused map and reduce rates on one 2 GiB fp32 tile (HBM-bound kernels)."""
  rows, cols = 8192, 65536            # BASELINE configs[2] tile: 8192 x 65536 fp32
  n = rows * cols
  X = sp.from_tile_fn((rows, cols), np.float32, lambda ex: device_uniform(ex, 0.0, 1.0, SEED + 7)).force()
  Xv = sp.Val(val=X)
  out = {}
  t = X.tiles
  x = ctx.tile(list(t.values())[0]).data
  y = torch.empty_like(x)
  ms = event_time(lambda: kernels.stream_copy(y, x), 10)
  out['stream_copy_GBps'] = round(2 * 4.0 * n / ms / 1e6, 1)
  del y
  ms = event_time(lambda: (Xv * Xv + Xv).optimized().force(), 10)
  out['map_xx_plus_x_GBps'] = round(8.0 * n / ms / 1e6, 1)          # 4*(n_in+1)*E bytes
  ms = event_time(lambda: (Xv + 1).force(), 10)
  out['map_x_plus_1_GBps'] = round(8.0 * n / ms / 1e6, 1)
  # fused trees outside the prebuilt library: run-time specialised kernels (csrc/sp_jit.hip)
  ms = event_time(lambda: (((Xv * Xv + Xv) * 0.5 - Xv) / (Xv + 2.0)).optimized().force(), 10)
  out['map_5op_chain_jit_GBps'] = round(8.0 * n / ms / 1e6, 1)
  ms = event_time(lambda: sp.sum((Xv - 0.5) * (Xv - 0.5), axis=0).optimized().force(), 10)
  out['sum_sq_dev_axis0_jit_GBps'] = round(4.0 * n / ms / 1e6, 1)
  for axis in (None, 0, 1):
    ms = event_time(lambda: sp.sum(Xv, axis).force(), 10)
    out['sum_axis%s_GBps' % axis] = round(4.0 * n / ms / 1e6, 1)    # 4*E bytes
  ms = event_time(lambda: sp.argmax(Xv, 1).force(), 10)
  out['argmax_axis1_GBps'] = round(4.0 * n / ms / 1e6, 1)
  del X, Xv, x
  torch.cuda.empty_cache()
  # the reference's DEFAULT dtype is float64 (its builders make np.float arrays): same tile shape halved
  Xd = sp.astype(sp.from_tile_fn((rows, cols // 2), np.float32,
                                 lambda ex: device_uniform(ex, 0.0, 1.0, SEED + 9)), np.float64).force()
  Xdv = sp.Val(val=Xd)
  nd = rows * (cols // 2)
  ms = event_time(lambda: (Xdv * Xdv + Xdv).optimized().force(), 10)
  out['map_xx_plus_x_f64_jit_GBps'] = round(16.0 * nd / ms / 1e6, 1)      # 8*(n_in+1)*E bytes
  ms = event_time(lambda: sp.sum(Xdv, 0).force(), 10)
  out['sum_axis0_f64_jit_GBps'] = round(8.0 * nd / ms / 1e6, 1)
  del Xd, Xdv
  torch.cuda.empty_cache()
  # one linear-regression step on a BASELINE configs[4] per-GPU tile (125000 x 4096 fp32):
  # yp = dot(X, w); grad = sum(X * (yp - y), axis=0)  (sgd.py:34-39) -- X streamed twice
  N, D = 125000, 4096
  Xl = sp.from_tile_fn((N, D), np.float32, lambda ex: device_uniform(ex, 0.0, 1.0, SEED + 11))
  yl = sp.from_tile_fn((N, 1), np.float32, lambda ex: device_uniform(ex, 0.0, 1.0, SEED + 12))
  w = np.random.RandomState(SEED).rand(D, 1).astype(np.float32)

  def lreg_step():
    yp = sp.dot(Xl, w)
    return sp.sum(Xl * (yp - yl), axis=0).optimized().force()
  ms = event_time(lreg_step, 10)
  out['lreg_step_GBps'] = round(2 * 4.0 * N * D / ms / 1e6, 1)     # 2*4*N*D bytes
  out['lreg_step_ms'] = round(ms, 4)
  out['frac_of_measured_copy'] = {k: round(v / out['stream_copy_GBps'], 3) for k, v in out.items()
                                  if k.endswith('_GBps') and k != 'stream_copy_GBps'}
  out['hbm_peak_GBps'] = HBM_PEAK_GBPS
  out['tile'] = '%dx%d fp32' % (rows, cols)
  return out


def kmeans_section(ctx):
  """One k-means iteration on a BASELINE configs[3] per-GPU tile (1 250 000 x 256 fp32 points,
  k = 1024): distance+argmin is MFMA-bound (2*n*k*d flop), the accumulate HBM-bound (4*n*d bytes)."""
  from spartan_amd.examples.sklearn.cluster import KMeans
  n, k, d = 1250000, 1024, 256
  X = sp.from_tile_fn((n, d), np.float32, lambda ex: device_uniform(ex, 0.0, 1.0, SEED + 21)).force()
  Xv = sp.Val(val=X)
  x = ctx.tile(list(X.tiles.values())[0]).data
  centers = np.random.RandomState(SEED).rand(k, d)
  cdev = ctx.backend.from_numpy(centers)
  labels = torch.empty(n, dtype=torch.int64, device=x.device)
  out = {'tile': '%dx%d fp32, k=%d' % (n, d, k)}
  ms = event_time(lambda: kernels.nearest_center(x, cdev, labels), 20, warmup=3)
  out['assign_ms'] = round(ms, 3)
  out['assign_TFLOPs'] = round(2.0 * n * k * d / ms / 1e9, 1)          # 2*N*K*D flop
  out['assign_frac_of_mfma_peak'] = round(2.0 * n * k * d / ms / 1e9 / MFMA_F32_PEAK_TFLOPS, 3)
  kernels.nearest_center(x, cdev, labels, _hip.NEAREST_FUSED_UNCHECKED)
  out['assign_rechecked_points'] = int((labels < 0).sum().item())   # re-done by the exact fp64 kernel
  kernels.nearest_center(x, cdev, labels)
  sums = torch.empty(k, d, dtype=torch.float32, device=x.device)
  counts = torch.empty(k, dtype=torch.int64, device=x.device)

  def accumulate():
    kernels.bincount(labels, k, counts)
    kernels.segment_sum(x, labels, k, sums)
  ms = event_time(accumulate, 10)
  out['accumulate_ms'] = round(ms, 3)
  out['accumulate_GBps'] = round(4.0 * n * d / ms / 1e6, 1)             # 4*N*D bytes
  km = KMeans(k, 1)
  t = []
  for _ in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    km.fit(Xv, centers, implementation='map2', reducer=np.add)   # glom of counts / centres synchronises
    t.append(time.perf_counter() - t0)
  out['iteration_ms'] = round(min(t[1:]) * 1e3, 3)                     # driver program end to end
  return out


def sparse_section(ctx):
  """the sparse multiply of tests/benchmark_pagerank.py on one worker's tile -- 900 000 pages,
  10 out-links per page, 90 % of them inside one of 8 sites -- p <- W . p with W a device CSR tile.
  HBM-bound: 8 B per stored entry (value + column index) + 16 B per row (indptr, y, x read once)."""
  from spartan_amd import sparse as S
  n, deg, sites = 900000, 10, 8
  g = torch.Generator(device='cuda')
  g.manual_seed(SEED + 31)
  cols = torch.arange(n, device='cuda', dtype=torch.int64).repeat_interleave(deg)
  local = (cols // (n // sites)) * (n // sites) + torch.randint(0, n // sites, (n * deg,), device='cuda', generator=g)
  far = torch.randint(0, n, (n * deg,), device='cuda', generator=g)
  rows = torch.where(torch.rand(n * deg, device='cuda', generator=g) <= 0.9, local, far).int()
  cols = cols.int()
  vals = torch.ones(n * deg, device='cuda', dtype=torch.float32)
  out = {'tile': '%dx%d fp32 CSR, %d links per page' % (n, n, deg)}
  ms = event_time(lambda: S.from_coo((n, n), np.float32, rows, cols, vals), 3, warmup=1)
  out['coo_to_csr_ms'] = round(ms, 3)
  W = S.from_coo((n, n), np.float32, rows, cols, vals)
  del rows, cols, vals, local, far
  x = torch.rand((n, 1), device='cuda', dtype=torch.float32, generator=g)
  y = torch.empty((n, 1), device='cuda', dtype=torch.float32)
  alg = W.nnz * 8 + n * 16
  ms = event_time(lambda: S.spmm(W, x, out=y), 20, warmup=3)
  out.update({'nnz': W.nnz, 'spmv_ms': round(ms, 4), 'spmv_GBps': round(alg / ms / 1e6, 1),
              'spmv_bytes_per_launch': alg})
  # the driver program: 5 iterations of p = dot(wts, p) through the expression API on the same tile
  if True:
    wts = sp.from_tile_fn((n, n), np.float32, lambda ex: W, sparse=True).force()
    p = sp.from_tile_fn((n, 1), np.float32, lambda ex: x).force()
    t = []
    for _ in range(3):
      torch.cuda.synchronize()
      t0 = time.perf_counter()
      q = sp.Val(val=p)
      for _ in range(5):
        q = sp.dot(sp.Val(val=wts), q).optimized()
      q.force()
      torch.cuda.synchronize()
      t.append(time.perf_counter() - t0)
    out['five_iterations_ms'] = round(min(t[1:]) * 1e3, 3)
  return out


def dist_section(ctx):
  """N > 1: BASELINE configs[2] -- an array of N row tiles of 8192 x 65536 fp32 (2 GiB per GPU), reduced along
  every axis through the expression API: the per-tile kernels plus the RCCL combine (reduce to the owner for
  axis=None, reduce-scatter for axis=0, nothing for axis=1 / argmax axis=1 / the fused map).  GB/s = whole-job
  algorithmic bytes / wall-clock, barrier + synchronize on both sides, max over ranks."""
  p = ctx.world.size
  R = int(os.environ.get('SPARTAN_BENCH_DIST_ROWS', '8192'))
  C = 65536
  X = sp.from_tile_fn((R * p, C), np.float32, lambda ex: device_uniform(ex, 0.0, 1.0, SEED + 41), tile_hint=(R, C)).force()
  Xv = sp.Val(val=X)
  E = float(R) * p * C
  out = {'array': '%d x %d fp32, %d row tiles of %d x %d' % (R * p, C, p, R, C)}
  keep = []
  progs = (('sum_axisNone', lambda: sp.sum(Xv), 4.0), ('sum_axis0', lambda: sp.sum(Xv, 0), 4.0),
           ('sum_axis1', lambda: sp.sum(Xv, 1), 4.0), ('argmax_axis1', lambda: sp.argmax(Xv, 1), 4.0),
           ('map_xx_plus_x', lambda: (Xv * Xv + Xv).optimized(), 8.0))
  for name, build, bpe in progs:
    def step():
      keep[:] = [build().force()]
    dt = time_steps(ctx, step, 5, 2)
    out[name + '_GBps'] = round(bpe * E * 5 / dt / 1e9, 1)
    del keep[:]
  return out


def guarded(fn, timeout_s, rank, fallback_line):
  """Run an informational section; if it does not come back (a collective that never completes), rank 0 prints the
  line it already has and every rank leaves -- the headline measurement is never lost to an extra."""
  import threading
  done = threading.Event()

  def watchdog():
    if not done.wait(timeout_s):
      fallback_line['extras_error'] = 'FAILED: section did not complete within %d s (hung collective?)' % timeout_s
      _emit(fallback_line, rank)
      os._exit(0)   # the headline (measured before this section) stands; the failure is in the line itself
  threading.Thread(target=watchdog, daemon=True).start()
  try:
    res = fn()
  except Exception as e:   # informational: report, do not fail the run
    res = {'error': '%s: %s' % (type(e).__name__, str(e)[:300])}
  done.set()
  return res


def northstar_section(ctx):
  """dot 32768 x 32768 x 32768 fp32 on ONE GPU (a single 4 GiB tile per operand): the north-star shape, timed
  around whole `spartan.dot(A, B).force()` calls with HIP events on the launch stream."""
  n = NORTH_STAR
  A = sp.from_tile_fn((n, n), np.float32, lambda ex: device_uniform(ex, -1.0, 1.0, SEED + 51))
  B = sp.from_tile_fn((n, n), np.float32, lambda ex: device_uniform(ex, -1.0, 1.0, SEED + 52))
  A.force()
  B.force()
  keep = []

  def step():
    keep[:] = [sp.dot(A, B).force()]
  step()
  torch.cuda.synchronize()
  ms = []
  for _ in range(3):
    e0, e1 = kernels.Event(), kernels.Event()
    e0.record()
    step()
    e1.record()
    e1.synchronize()
    ms.append(e0.elapsed_ms(e1))
  del keep[:]
  avg = sum(ms) / len(ms)
  flop = 2.0 * n ** 3
  return {'workload': 'spartan.dot %dx%dx%d fp32, one tile' % (n, n, n), 'calls': len(ms),
          'ms_per_call': round(avg, 2), 'min_ms': round(min(ms), 2), 'TFLOPs': round(flop / avg / 1e9, 2),
          'frac_of_mfma_peak': round(flop / avg / 1e9 / MFMA_F32_PEAK_TFLOPS, 4), 'flop_per_call': flop}


def cpu_baseline():
  """The reference's execution model on the host cores of this box: W = min(physical
  cores, 64) worker processes, one per core, pinned, one BLAS thread each (spartan/worker.py:40,385-387), running
  the oracle's NumPy tile bodies on bounded samples of the BASELINE shapes; the parent merges like the owner of
  the target tile.  A reported baseline, not a target."""
  from oracle import cpu_workers
  t_all = time.perf_counter()
  pool = cpu_workers.Workers(64)
  try:
    W = pool.count
    n = 4096
    t_dot, t_dot_compute, n = pool.dot(n)                       # K-split, one target tile (dot.py:277-290)
    rows, cols = 65536, 16384                                    # configs[2] scaled 1/4: 4 GiB fp32 in all
    t_map, t_sum, rows = pool.map_and_sum(rows, cols)
    ln, ld = 125000, 4096                                        # configs[4] scaled 1/8: one per-GPU tile
    t_lreg, ln = pool.lreg_step(ln, ld)
    kn, kd, kk = 2000 * W, 256, 1024                             # configs[3]: k and d as given, 2000 points / worker
    t_km, kn = pool.kmeans_iteration(kn, kd, kk)
  finally:
    pool.close()
  e = float(rows) * cols
  return {'value': round(2.0 * n ** 3 / t_dot / 1e12, 4), 'unit': 'TFLOP/s', 'cores': W, 'kind': 'port',
          'workers': '%d processes pinned to %d physical cores, 1 BLAS thread each' % (W, W),
          'dot': {'shape': '%dx%dx%d fp32, K-split over %d workers, one target tile' % (n, n, n, W),
                  'seconds': round(t_dot, 3), 'gemm_seconds': round(t_dot_compute, 3),
                  'TFLOPs': round(2.0 * n ** 3 / t_dot / 1e12, 4),
                  'note': 'the partials travel to the owner through pipes and are added there one by one, as the '
                          'reference pickles them to the owner of its single target tile (dot.py:277-278)'},
          'map_xx_plus_x_GBps': round(8.0 * e / t_map / 1e9, 2), 'sum_axis0_GBps': round(4.0 * e / t_sum / 1e9, 2),
          'map_sum_shape': '%dx%d fp32 in %d row tiles' % (rows, cols, W),
          'lreg_step': {'shape': '%dx%d fp32' % (ln, ld), 'seconds': round(t_lreg, 4),
                        'GBps': round(2 * 4.0 * ln * ld / t_lreg / 1e9, 2)},
          'kmeans_iteration': {'shape': '%dx%d points, k=%d' % (kn, kd, kk), 'seconds': round(t_km, 3),
                               'TFLOPs_of_2nkd': round(2.0 * kn * kk * kd / t_km / 1e12, 4)},
          'sample': 'oracle tile bodies (NumPy / BLAS / scipy cdist) on %d pinned one-thread workers: dot %d^3 '
                    'K-split; x*x+x and sum(axis=0) on %dx%d; one lreg step on %dx%d; one k-means iteration on '
                    '%dx%d, k=%d -- scaled from the BASELINE shapes to stay within ~20 s' %
                    (W, n, rows, cols, ln, ld, kn, kd, kk),
          'wall_seconds': round(time.perf_counter() - t_all, 1)}


_REAL_STDOUT = None


def _claim_stdout():
  """stdout carries ONE JSON line and nothing else.  Native libraries write there too -- RCCL's version banner
  (NCCL_DEBUG=VERSION is set on the GPU boxes; it sits in the C stdout buffer until exit), gloo's "[Gloo] Rank ..."
  lines -- so file descriptor 1 is pointed at stderr for the rest of the process and the line is written to the
  saved descriptor."""
  global _REAL_STDOUT
  if _REAL_STDOUT is None:
    sys.stdout.flush()
    _REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)


def _emit(line, rank):
  """The ONE line of stdout, from rank 0."""
  sys.stdout.flush()
  if rank == 0:
    os.write(_REAL_STDOUT if _REAL_STDOUT is not None else 1, (json.dumps(line) + '\n').encode())


def main():
  _claim_stdout()
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=20)
  ap.add_argument('--warmup', type=int, default=3)
  ap.add_argument('--size', type=int, default=0, help='matrix order (default: 8192 on one GPU, 32768 on several)')
  ap.add_argument('--no-extras', action='store_true', help='skip the map/reduce, north-star and CPU-baseline sections')
  args = ap.parse_args()

  world = sp.World.from_env()
  if world.size != args.gpus:
    raise SystemExit('--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run --nproc-per-node %d)'
                     % (args.gpus, world.size, args.gpus))
  ctx = sp.initialize('hip', world=world)
  p = world.size
  n = args.size or (8192 if p == 1 else NORTH_STAR)
  if p == 1:
    A = sp.from_tile_fn((n, n), np.float32, lambda ex: device_uniform(ex, -1.0, 1.0, SEED))
    B = sp.from_tile_fn((n, n), np.float32, lambda ex: device_uniform(ex, -1.0, 1.0, SEED + 1))
    hint = None
    tag = {8192: 'BASELINE configs[1]', NORTH_STAR: 'north-star shape'}.get(n, 'custom size')
    workload = 'spartan.dot %dx%dx%d fp32, one tile (%s)' % (n, n, n, tag)
    parallelism = 'single tile'
    scaling = 'weak'
  else:
    if n % p:
      raise SystemExit('--size %d is not a multiple of --gpus %d' % (n, p))
    hint = (n // p, n)
    A = sp.from_tile_fn((n, n), np.float32, lambda ex: device_uniform(ex, -1.0, 1.0, SEED), tile_hint=hint)
    B = sp.from_tile_fn((n, n), np.float32, lambda ex: device_uniform(ex, -1.0, 1.0, SEED + 1), tile_hint=hint)
    workload = ('spartan.dot %dx%dx%d fp32 (north star), A, B and the result row-tiled %d x (%dx%d)'
                % (n, n, n, p, n // p, n))
    parallelism = ('K-split map2 join, 1 worker per GPU: all-to-all of A blocks, %d GEMMs per column chunk, one '
                   'asynchronous reduce-scatter per chunk behind the next chunk (transport: %s)'
                   % (p, world.note or getattr(world.transport, 'name', '?')))
    scaling = 'strong'
  A.force()
  B.force()

  keep = []

  def step():
    keep[:] = [sp.dot(A, B, tile_hint=hint).force()]

  # set-up launches (untimed, before the W warm-up steps): library load, allocator warm-up, and the
  # device's one-off dispatch stall (~30 ms, seen once per process about 50 ms into the first sustained
  # MFMA load on these boxes: profiles/r01_notes.md) -- so neither lands in the timed steps
  for _ in range(SETUP_LAUNCHES if p == 1 else 2):
    step()
  torch.cuda.synchronize()
  ctx.backend.gemm_events = []
  stats0 = dict(world.stats)
  dt = time_steps(ctx, step, args.steps, args.warmup)
  torch.cuda.synchronize()
  stats1 = dict(world.stats)
  all_events = ctx.backend.gemm_events
  ctx.backend.gemm_events = None
  # launches per step: 1 on one GPU; p per column chunk of the partial on several (dot.ksplit_plan)
  per_step = max(1, len(all_events) // (args.steps + args.warmup))
  events = all_events[-args.steps * per_step:]
  kernel_ms = [e0.elapsed_ms(e1) for (e0, e1, _, _, _) in events]
  flops_launch = 2.0 * events[0][2] * events[0][3] * events[0][4]
  avg_ms = sum(kernel_ms) / len(kernel_ms)
  achieved = flops_launch / (avg_ms * 1e-3) / 1e12

  flop_step = 2.0 * n * n * n
  value = flop_step * args.steps / dt / 1e12
  line = {
      'metric': 'spartan.dot TFLOP/s (+ map/reduce HBM GB/s)', 'value': round(value, 2), 'unit': 'TFLOP/s',
      'n_gpus': p, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3),
      'higher_is_better': True, 'scaling': scaling, 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
      'config': {'workload': workload, 'parallelism': parallelism, 'flop_per_step': flop_step,
                 'setup_launches': SETUP_LAUNCHES if p == 1 else 2,
                 'inputs': 'uniform[-1,1) fp32 generated on device, resident in HBM'},
      'roofline': {'bound': 'mfma', 'kernel': 'sp_gemm_glds_kernel<256x128x16, 4 waves> (v_mfma_f32_32x32x2_f32, k-tiles by global_load_lds)',
                   'achieved': round(achieved, 2), 'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                   'frac': round(achieved / MFMA_F32_PEAK_TFLOPS, 4),
                   'flop_per_launch': flops_launch, 'avg_launch_ms': round(avg_ms, 4),
                   'launches_per_step': per_step, 'traffic': None},
  }
  traffic_file = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
  if os.path.exists(traffic_file):
    try:
      line['roofline']['traffic'] = json.load(open(traffic_file)).get('gemm_%d' % n)
      line['roofline']['traffic_source'] = ('profiles/pmc_traffic.json: 2*FETCH_SIZE + WRITE_SIZE of a separate rocprofv3 '
                                            '--pmc run of this kernel and shape (not of this run); algorithmic floor '
                                            '%d bytes' % (12 * n * n))
    except Exception:
      pass
  if p > 1:
    # this rank's share of a step: kernel-only time vs wall, and the bytes it moved against the xGMI rates
    sent = {k: (stats1[k] - stats0[k]) / float(args.steps + args.warmup) for k in stats0}
    step_s = dt / args.steps
    line['dot_breakdown'] = {
        'gemm_kernel_ms_per_step': round(sum(kernel_ms) / args.steps, 3),
        'step_ms': round(step_s * 1e3, 3),
        'kernel_only_TFLOPs_whole_job': round(flop_step / (sum(kernel_ms) / args.steps * 1e-3) / 1e12, 2),
        'all_to_all_bytes_per_gpu_per_step': int(sent['p2p_bytes']),
        'reduce_scatter_bytes_per_gpu_per_step': int(sent['collective_bytes']),
        'exchange_GBps_per_gpu_if_serial': round((sent['p2p_bytes'] + sent['collective_bytes']) / step_s / 1e9, 1),
        'xgmi_GBps_per_gpu': {'one_link': XGMI_LINK_GBPS, 'seven_links': 7 * XGMI_LINK_GBPS},
        'note': 'bytes are what ONE GPU sends per step ((p-1)/p of its 4*M*N partial and of its A tile); the '
                'reduce-scatters run on a communication stream behind the GEMMs of the next chunk',
    }
  if world.rank == 0 and not args.no_extras and p == 1:
    del keep[:]
    torch.cuda.empty_cache()
    line['northstar_%d' % NORTH_STAR] = northstar_section(ctx)
    torch.cuda.empty_cache()
    line['hbm'] = hbm_section(ctx)
    torch.cuda.empty_cache()
    line['kmeans'] = kmeans_section(ctx)
    torch.cuda.empty_cache()
    line['sparse'] = sparse_section(ctx)
    torch.cuda.empty_cache()
    line['cpu_baseline'] = cpu_baseline()
  if world.distributed:
    line['comm'] = dict(world.stats)
    line['comm']['transport'] = world.note or getattr(world.transport, 'name', '?')
    if not args.no_extras:
      del keep[:]
      del A, B
      torch.cuda.empty_cache()
      line['hbm_dist'] = guarded(lambda: dist_section(ctx), 240, world.rank, dict(line))
  world.barrier()
  _emit(line, world.rank)
  sp.shutdown()
  world.close()
  if world.distributed:
    torch.distributed.destroy_process_group()


if __name__ == '__main__':
  main()
