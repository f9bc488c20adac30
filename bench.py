#!/usr/bin/env python
"""Headline benchmark: BASELINE.json's metric
    "spartan.dot TFLOP/s + map/reduce HBM GB/s at 1/2/4/8 MI355X".

One "step" = one `spartan.dot(A, B).force()` through the whole tile path (lazy DAG -> per-tile fp32 MFMA GEMM
launches -> merge into the target), operands already resident in HBM:

  --gpus 1 : BASELINE configs[1] -- dot 8192 x 8192 x 8192 fp32, one tile.  The same line carries the north-star
             shape on one GPU (`northstar_32768`: dot 32768^3, timed with HIP events around whole `.force()` calls),
             the HBM-bound kernels, the k-means / lreg workloads of configs[3] / [4] on a per-GPU tile, and
             `ksplit_rank_emulation`: ONE rank's share of the p-GPU north-star step for p = 2, 4, 8, with every
             collective replaced by device copies of the same byte count on the communication stream (no multi-GPU
             hardware is needed to see what the GEMMs lose to a concurrent transfer and what the pipeline exposes).
  --gpus N : the north star, STRONG scaling: dot 32768 x 32768 x 32768 fp32 with A, B and the result row-tiled one
             tile per GPU (`tile_hint=(M/N, N)`).  rows <= cols, so this is the reference's K-split map2 join
             (dot.py:286-290): the all-to-all of A's column slabs into one slab buffer, one GEMM per column chunk of
             the partial, and one reduce-scatter per chunk overlapped with the next chunk's GEMM
             (spartan_amd/expr/dot.ksplit_pipeline).  `value` is whole-job TFLOP/s = 2 * 32768^3 * steps /
             max-over-ranks wall time; `dot_breakdown` gives kernel-only time and the bytes each GPU moved per step.

The line also carries the roofline of the dominant kernel (MFMA-bound GEMM; durations from HIP events on the launch
stream) and the CPU baseline: the oracle's tile bodies on W = min(physical cores, 64) pinned one-thread worker
processes.  stdout gets the COMPACT line (tools/bench_line.py, < 10 KB: contract keys, compact roofline with every
HBM section as [GB/s, fraction of measured copy], cpu_baseline, one summary per extra section); the detailed record
-- raw sections, every emulation case, and `profile_table` (what every timed section launched: kernel, launches,
algorithmic units per launch -- the key tools/roofline.py uses to recompute each fraction from a rocprofv3 kernel
trace of this very command) -- goes to gpurun_out/bench_detail_n<N>.json ($SP_BENCH_DETAIL) and to stderr.

No torch in this process at any N: the control plane of a multi-rank run is the rendezvous hub of rank 0
(spartan_amd/rendezvous.py, standard library sockets); the data plane is RCCL called from libspartan_hip.so.
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

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import spartan_amd as sp  # noqa: E402
from spartan_amd import _hip, kernels  # noqa: E402
from spartan_amd import devarray as D  # noqa: E402
from tools import bench_line  # noqa: E402

MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
MFMA_BF16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16, dense
HBM_PEAK_GBPS = 8000.0         # MI355X_MICROARCH.md: HBM3E spec peak
SEED = 20150708
SETUP_LAUNCHES = 8              # untimed set-up steps before the W warm-up steps (see main)
NORTH_STAR = 32768
XGMI_LINK_GBPS = 153.0          # per link, 7 links per GPU (MI355X_MICROARCH.md / task brief)

GEMM_KERNEL = 'sp_gemm_glds_kernel'
PROFILE_TABLE = []              # one entry per timed section (see the module docstring)


def clocks():
  """Host clocks in ns, one of which is the time base of rocprofv3's kernel timestamps on this box
  (tools/roofline.py finds out which by counting the launches that fall into the windows)."""
  return {'monotonic': time.clock_gettime_ns(time.CLOCK_MONOTONIC), 'boottime': time.clock_gettime_ns(time.CLOCK_BOOTTIME),
          'monotonic_raw': time.clock_gettime_ns(time.CLOCK_MONOTONIC_RAW), 'realtime': time.time_ns()}


def note_section(label, kernel, launches, units, unit, bound, t0, t1):
  """One timed section: every launch of `kernel` issued between the host times t0 and t1 (the device was idle at
  both) belongs to it."""
  PROFILE_TABLE.append({'label': label, 'kernel': kernel, 'launches': int(launches), 'units_per_launch': units,
                        'unit': unit, 'bound': bound, 't0': t0, 't1': t1})


def device_uniform(ex, lo, hi, seed):
  """One tile of uniform [lo, hi) fp32 from the library's counter-based generator (sp_random_fill)."""
  out = D.empty(ex.shape, np.float32)
  kernels.random_fill(out, 'uniform', seed + 1000003 * (ex.ul[0] if ex.ul else 0), 0)
  if (lo, hi) != (0.0, 1.0):
    out = out * np.float32(hi - lo) + np.float32(lo)
  return out


def time_steps(ctx, step, steps, warmup):
  for _ in range(warmup):
    step()
  ctx.world.barrier()
  D.synchronize()
  t0 = time.perf_counter()
  for _ in range(steps):
    step()
  D.synchronize()
  ctx.world.barrier()
  dt = time.perf_counter() - t0
  if ctx.world.distributed:
    dt = max(ctx.world.all_gather_object(dt))       # max over ranks (control plane)
  return dt


def event_time(fn, iters, warmup=4, section=None):
  """Average milliseconds of fn() over `iters` calls, by HIP events on the launch stream.
  section = (label, kernel substring, algorithmic units per launch, unit, bound[, launches per call])."""
  D.synchronize()
  t0 = clocks()
  for _ in range(warmup):
    fn()
  # programs outside the prebuilt kernel library are specialised at run time on a
  # background thread (include/spartan_hip.h sp_jit_*): let the warm-up's requests land
  _hip.lib().sp_jit_wait()
  fn()
  D.synchronize()
  e0, e1 = D.Event(), D.Event()
  e0.record()
  for _ in range(iters):
    fn()
  e1.record()
  e1.synchronize()
  if os.environ.get('SP_BENCH_TRACE'):
    import gc
    live, pooled = D.blob_stats()
    sys.stderr.write('event_time %s: %.3f ms/call, live blobs %d, pooled %.1f GiB, gc %s\n'
                     % (section[0] if section else '-', e0.elapsed_ms(e1) / iters, live, pooled / 2.0 ** 30, gc.get_count()))
  if section is not None:
    per_call = section[5] if len(section) > 5 else 1
    note_section(section[0], section[1], (warmup + 1 + iters) * per_call, section[2], section[3], section[4], t0, clocks())
  return e0.elapsed_ms(e1) / iters


def hbm_section(ctx):
  """Fused map and reduce rates on one 2 GiB fp32 tile (HBM-bound kernels)."""
  rows, cols = 8192, 65536            # BASELINE configs[2] tile: 8192 x 65536 fp32
  n = rows * cols
  X = sp.from_tile_fn((rows, cols), np.float32, lambda ex: device_uniform(ex, 0.0, 1.0, SEED + 7)).force()
  Xv = sp.Val(val=X)
  out = {}
  x = ctx.tile(list(X.tiles.values())[0]).data
  y = D.empty(x.shape, x.dtype)
  ms = event_time(lambda: kernels.stream_copy(y, x), 10, section=('stream copy 2 GiB', 'sp_stream_copy_kernel', 8.0 * n, 'bytes', 'hbm'))
  out['stream_copy_GBps'] = round(2 * 4.0 * n / ms / 1e6, 1)
  del y
  ms = event_time(lambda: (Xv * Xv + Xv).optimized().force(), 10,
                  section=('map x*x+x', 'sp_map_kernel', 8.0 * n, 'bytes', 'hbm'))
  out['map_xx_plus_x_GBps'] = round(8.0 * n / ms / 1e6, 1)          # 4*(n_in+1)*E bytes
  ms = event_time(lambda: (Xv + 1).force(), 10, section=('map x+1', 'sp_map_kernel', 8.0 * n, 'bytes', 'hbm'))
  out['map_x_plus_1_GBps'] = round(8.0 * n / ms / 1e6, 1)
  # fused trees outside the prebuilt library: run-time specialised kernels (csrc/sp_jit.hip).  The FIRST call of
  # the process is timed on its own: with the code object in csrc/jit_seed (built by __graft_entry__.build()) it runs
  # specialised at once; without, it runs on the interpreter tier while hipRTC compiles in the background.
  chain = lambda: (((Xv * Xv + Xv) * 0.5 - Xv) / (Xv + 2.0)).optimized().force()     # noqa: E731
  # the process's first two launches of the program, one at a time on an idle device.  The expression is built and
  # optimised before the events (host work, ~0.3 ms, that a loop overlaps with the previous launch): the timed
  # force() lowers it, finds the code object (preloaded from csrc/jit_seed by the backend) and launches.
  # (both results stay alive while they are timed: two free tiles in the store first, so that neither call measures
  #  a 2 GiB hipMalloc -- results die with their last reference now, the pool no longer holds leftovers)
  spare = [D.empty(x.shape, x.dtype) for _ in range(2)]
  del spare
  pending = [(((Xv * Xv + Xv) * 0.5 - Xv) / (Xv + 2.0)).optimized() for _ in range(2)]
  for tag, e in zip(('first', 'second'), pending):
    D.synchronize()
    e0, e1 = D.Event(), D.Event()
    e0.record()
    e.force()
    e1.record()
    e1.synchronize()
    out['map_5op_chain_%s_call_GBps' % tag] = round(8.0 * n / e0.elapsed_ms(e1) / 1e6, 1)
  del pending
  ms = event_time(chain, 10)
  out['map_5op_chain_jit_GBps'] = round(8.0 * n / ms / 1e6, 1)
  # the fused map -> reduce the same way: its first launch of the process, alone on an idle device (seeded by
  # build(): specialised at once; an unseeded program starts on the interpreter tier below)
  # (the first REDUCTION of a process pays ~12 ms of one-time host work -- module imports, the first workspace --
  # whatever it reduces: a 64 x 64 one takes that, so that the figure below is about the program, not the process)
  sp.sum(sp.ones((64, 64)) * 2.0, axis=0).optimized().glom()
  first = sp.sum((Xv - 0.5) * (Xv - 0.5), axis=0).optimized()
  D.synchronize()
  e0, e1 = D.Event(), D.Event()
  e0.record()
  first.force()
  e1.record()
  e1.synchronize()
  out['sum_sq_dev_axis0_first_call_GBps'] = round(4.0 * n / e0.elapsed_ms(e1) / 1e6, 1)
  del first
  ms = event_time(lambda: sp.sum((Xv - 0.5) * (Xv - 0.5), axis=0).optimized().force(), 10)
  out['sum_sq_dev_axis0_jit_GBps'] = round(4.0 * n / ms / 1e6, 1)
  # the same two programs on the interpreter tier (what a program nobody seeded runs on until hipRTC is done, and
  # every small tile): run-time specialisation switched off around the measurement
  _hip.lib().sp_jit_configure(0, -1)
  try:
    ms = event_time(chain, 10, section=('map 5-op chain, interpreted', 'DynProg', 8.0 * n, 'bytes', 'hbm'))
    out['map_5op_chain_interpreter_GBps'] = round(8.0 * n / ms / 1e6, 1)
    ms = event_time(lambda: sp.sum((Xv - 0.5) * (Xv - 0.5), axis=0).optimized().force(), 10)
    out['sum_sq_dev_axis0_interpreter_GBps'] = round(4.0 * n / ms / 1e6, 1)
  finally:
    _hip.lib().sp_jit_configure(1, -1)
  for axis, kern in ((None, 'sp_reduce_rows_kernel'), (0, 'sp_reduce_cols_kernel'), (1, 'sp_reduce_rows_kernel')):
    ms = event_time(lambda: sp.sum(Xv, axis).force(), 10,
                    section=('sum axis=%s' % axis, kern, 4.0 * n, 'bytes', 'hbm'))
    out['sum_axis%s_GBps' % axis] = round(4.0 * n / ms / 1e6, 1)    # 4*E bytes
  ms = event_time(lambda: sp.argmax(Xv, 1).force(), 10,
                  section=('argmax axis=1', 'sp_reduce_rows_kernel', 4.0 * n, 'bytes', 'hbm'))
  out['argmax_axis1_GBps'] = round(4.0 * n / ms / 1e6, 1)
  # the rest of SURVEY 8d's C3 list: the index reductions along the other axes, max / min (4*E bytes each)
  for name, build in (('argmax_axisNone', lambda: sp.argmax(Xv)), ('argmax_axis0', lambda: sp.argmax(Xv, 0)),
                      ('argmin_axis1', lambda: sp.argmin(Xv, 1)), ('max_axisNone', lambda: sp.max(Xv)),
                      ('max_axis0', lambda: sp.max(Xv, 0)), ('max_axis1', lambda: sp.max(Xv, 1)),
                      ('min_axisNone', lambda: sp.min(Xv)), ('min_axis0', lambda: sp.min(Xv, 0))):
    ms = event_time(lambda: build().force(), 10)
    out[name + '_GBps'] = round(4.0 * n / ms / 1e6, 1)
  del X, Xv, x
  D.trim_pool()
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
  D.trim_pool()
  out['frac_of_measured_copy'] = {k: round(v / out['stream_copy_GBps'], 3) for k, v in out.items()
                                  if k.endswith('_GBps') and k != 'stream_copy_GBps'}
  out['hbm_peak_GBps'] = HBM_PEAK_GBPS
  out['tile'] = '%dx%d fp32' % (rows, cols)
  return out


def probed_peaks():
  """Dense matrix-pipe rates measured on THIS GPU by tools/mfma_peak_probe (built by __graft_entry__.build()): every CU
  issuing nothing but independent MFMAs.  {} when the probe is not there."""
  import subprocess
  if _PROBED:
    return _PROBED
  exe = os.path.join(ROOT, 'tools', 'mfma_peak_probe')
  if not os.path.exists(exe):
    return {}
  D.synchronize()
  try:
    text = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=120).stdout.decode()
  except Exception:
    return {}
  best = {}
  for line in text.splitlines():
    if line.startswith('PROBE '):
      f = dict(kv.split('=') for kv in line.split()[2:])
      name = line.split()[1]
      best[name] = max(best.get(name, 0.0), float(f['TFLOPs']))
  _PROBED.update(best)
  return best


_PROBED = {}


MFMA_F64_SPEC_TFLOPS = 78.6     # AMD's MI355X figure for the fp64 matrix pipe (not in MI355X_MICROARCH.md: probed below)


def dot_f64_section(ctx):
  """The reference's own dot benchmark is float64 (tests/benchmark_dot.py:17-27; its builders default to float64):
  spartan.dot 8192^3 fp64 on one tile against the fp64 matrix rate PROBED on this GPU (a loop of
  v_mfma_f64_16x16x4_f64 on every CU).  The benchmark's sqrt(p) x sqrt(p) GRID tiling is not timed: on grid tiles
  the reference's join multiplies one-index-thick slabs (extent.pyx:545-552) -- not the matrix product, reproduced
  bit for bit and pinned by tests/test_dot_grid_tiles.py -- so there is no GEMM there to price."""
  n = 8192
  mk = lambda seed, hint=None: sp.astype(sp.from_tile_fn((n, n), np.float32, lambda ex: device_uniform(ex, -1.0, 1.0, seed),   # noqa: E731
                                                          tile_hint=hint), np.float64).force()
  peaks = probed_peaks()
  probed = peaks.get('v_mfma_f64_16x16x4_f64')
  peak = probed or MFMA_F64_SPEC_TFLOPS
  A, B = mk(SEED + 51), mk(SEED + 52)
  Av, Bv = sp.Val(val=A), sp.Val(val=B)
  flop = 2.0 * n * n * n
  ms = event_time(lambda: sp.dot(Av, Bv).force(), 5, warmup=2, section=('dot 8192^3 fp64', 'sp_dgemm_kernel', flop, 'flop', 'mfma_f64'))
  out = {'workload': 'spartan.dot %dx%dx%d fp64, one tile' % (n, n, n), 'ms_per_call': round(ms, 3),
         'TFLOPs': round(flop / ms / 1e9, 2), 'peak_TFLOPs': round(peak, 2),
         'peak_is': ('probed on this GPU (tools/mfma_peak_probe: v_mfma_f64_16x16x4_f64 on every CU; the same probe gives '
                     '%.1f TFLOP/s for v_mfma_f32_32x32x2_f32)' % peaks.get('v_mfma_f32_32x32x2_f32', float('nan')))
         if probed else 'spec (probe binary not built)',
         'frac_of_f64_mfma_peak': round(flop / ms / 1e9 / peak, 4)}
  del A, B, Av, Bv
  D.trim_pool()
  return out


def gemm_shapes_section(ctx):
  """sp_gemm_f32 (through kernels.gemm_f32, the call spartan.dot makes per tile) on shapes other than the headline's:
  the per-chunk GEMMs of the 8-GPU north-star pipeline (a 4096-row tile times 4096 / 8192-column chunks over the
  whole K), and cubes whose 256 x 128 tiles do not fill the 512 resident workgroups (the balanced kernel + fix-up
  launch, where its cost model picks it).  {shape: [TFLOP/s, fraction of the fp32 MFMA peak]}; HIP events."""
  out = {}
  for m, n, k in ((4096, 4096, 32768), (4096, 8192, 32768), (4096, 4096, 4096), (2048, 2048, 2048), (2304, 2304, 2304),
                  (3072, 3072, 3072), (5000, 5000, 5000)):
    a = device_uniform(type('E', (), {'shape': (m, k), 'ul': (0, 0)})(), -1.0, 1.0, SEED + 61)
    b = device_uniform(type('E', (), {'shape': (k, n), 'ul': (0, 0)})(), -1.0, 1.0, SEED + 62)
    c = D.empty((m, n), np.float32)
    iters = max(3, min(200, int(4e12 / (2.0 * m * n * k))))
    ms = event_time(lambda: kernels.gemm_f32(a, b, c), iters, warmup=3)
    tf = 2.0 * m * n * k / ms / 1e9
    out['%dx%dx%d' % (m, n, k)] = [round(tf, 1), round(tf / MFMA_F32_PEAK_TFLOPS, 3)]
    del a, b, c
    D.trim_pool()
  return out


def host_section(ctx):
  """Host time of the driver-side paths (what the device waits for between launches): microseconds until force()
  RETURNS on a 16 MiB tile (launches are asynchronous; the tile is small enough for the queue never to fill), and the
  state of the two tables that make a repeated DAG cheap -- optimised DAGs by structure (expr/plan.py) and lowered
  programs by operator structure (backend_hip._lowered)."""
  import importlib
  plan = importlib.import_module('spartan_amd.expr.plan')
  X = sp.from_tile_fn((1024, 4096), np.float32, lambda ex: device_uniform(ex, 0.0, 1.0, SEED + 3)).force()
  Xv = sp.Val(val=X)

  def host_us(fn, reps=200):
    for _ in range(20):
      fn()
    D.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
      fn()
    dt = time.perf_counter() - t0
    D.synchronize()
    return round(dt / reps * 1e6, 1)
  out = {'tile': '1024x4096 fp32',
         'x_plus_1_force_us': host_us(lambda: (Xv + 1).force()),
         'xx_plus_x_optimized_force_us': host_us(lambda: (Xv * Xv + Xv).optimized().force()),
         'chain5_optimized_force_us': host_us(lambda: (((Xv * Xv + Xv) * 0.5 - Xv) / (Xv + 2.0)).optimized().force()),
         'sum_axis0_force_us': host_us(lambda: sp.sum(Xv, 0).force()),
         'plan_table': dict(plan.stats), 'lowered_program_hits': ctx.backend.lowering_hits}
  return out


def lreg_section(ctx, copy_gbps):
  """BASELINE configs[4] on the per-GPU tile (125 000 x 4096 fp32): the benchmark's 100 gradient steps
  (tests/benchmark_lreg.py:22-29 -> examples/lreg.fit), after 2 untimed ones; a step is yp = dot(X, w);
  grad = sum(X * (yp - y), axis=0) (sgd.py:34-39) -- two passes over X as stated, one after the optimizer's rewrite --
  then the glom of the (D,) gradient and the update of w on the driver."""
  from spartan_amd.examples import lreg
  N, Dm = 125000, 4096
  Xl = sp.from_tile_fn((N, Dm), np.float32, lambda ex: device_uniform(ex, 0.0, 1.0, SEED + 11)).force()
  yl = sp.from_tile_fn((N, 1), np.float32, lambda ex: device_uniform(ex, 0.0, 1.0, SEED + 12)).force()
  Xv, yv = sp.Val(val=Xl), sp.Val(val=yl)
  w = np.random.RandomState(SEED).rand(Dm, 1).astype(np.float32)

  def lreg_step():
    yp = sp.dot(Xv, w)
    return sp.sum(Xv * (yp - yv), axis=0).optimized().force()
  # The expression states two passes over X (2*4*N*D bytes per step); the optimizer's one-pass rewrite
  # (expr/rowdot.py -> sp_rowdot_colsum_f32) reads X once: rates below are of the bytes actually streamed.
  from spartan_amd.expr.rowdot import RowDotColSumExpr
  one_pass = isinstance(sp.sum(Xv * (sp.dot(Xv, w) - yv), axis=0).optimized(), RowDotColSumExpr)
  passes = 1 if one_pass else 2
  step_bytes = passes * 4.0 * N * Dm
  ms = event_time(lreg_step, 10, section=(('lreg gradient, one pass over X', 'sp_rowdot_colsum_kernel', step_bytes, 'bytes', 'hbm')
                                          if one_pass else None))
  out = {'tile': '%dx%d fp32' % (N, Dm), 'passes_over_X_per_step': passes, 'step_kernels_ms': round(ms, 4),
         'step_kernels_GBps': round(step_bytes / ms / 1e6, 1)}
  alpha = 1e-10               # (the example's default 1e-6 diverges on a 125 000-row tile of uniform data: gradients ~6e7)
  w = lreg.fit(Xv, yv, 2, alpha=alpha, w=w)
  D.synchronize()
  t0 = time.perf_counter()
  w = lreg.fit(Xv, yv, 100, alpha=alpha, w=w)
  D.synchronize()
  dt = time.perf_counter() - t0
  out.update({'steps': 100, 'warmup_steps': 2, 'hundred_steps_ms': round(dt * 1e3, 2), 'ms_per_step': round(dt * 10, 4),
              'GBps': round(100 * step_bytes / dt / 1e9, 1),
              'frac_of_measured_copy': round(100 * step_bytes / dt / 1e9 / copy_gbps, 3),
              'two_pass_equivalent_GBps': round(100 * 2 * 4.0 * N * Dm / dt / 1e9, 1),
              'weights_finite': bool(np.isfinite(w).all()),
              'note': 'whole driver loop: %s + glom of the gradient + host update of w per step; two_pass_equivalent_GBps '
                      'counts X twice, as SURVEY 8d and the two-launch form do' %
                      ('one pass over X (2 launches: row pass + partial sums)' if one_pass else '2 launches')})
  return out


def kmeans_section(ctx):
  """BASELINE configs[3] on the per-GPU tile (1 250 000 x 256 fp32 points, k = 1024): distance + argmin is
  MFMA-bound (2*n*k*d flop), the accumulate HBM-bound (4*n*d bytes); then the benchmark as specified -- 10 timed
  iterations after 2 warm-up ones -- through KMeans.fit."""
  from spartan_amd.examples.sklearn.cluster import KMeans
  n, k, d = 1250000, 1024, 256
  X = sp.from_tile_fn((n, d), np.float32, lambda ex: device_uniform(ex, 0.0, 1.0, SEED + 21)).force()
  Xv = sp.Val(val=X)
  x = ctx.tile(list(X.tiles.values())[0]).data
  centers = np.random.RandomState(SEED).rand(k, d)
  cdev = ctx.backend.from_numpy(centers)
  labels = D.empty((n,), np.int64)
  out = {'tile': '%dx%d fp32, k=%d' % (n, d, k)}
  flop = 2.0 * n * k * d                                                # 2*N*K*D flop
  # the default tier (csrc/kmeans_split.hpp): every fp32 operand cut into two bf16 numbers, the contraction as three
  # exact-product bf16 MFMAs per 16 features with fp32 accumulation -- a FILTER with a proven error window; the points
  # inside it are re-decided in fp64, so the labels are argmin(cdist)'s.  Timed as a fit runs it (the points' images
  # made once: `prepared`) and as one stand-alone call (which cuts the points first).
  prepared = kernels.prepare_points(x)
  ms = event_time(lambda: kernels.nearest_center(x, cdev, labels, prepared=prepared), 20, warmup=3,
                  section=('k-means assign (first pass, bf16-split MFMA)', 'sp_nearest_split_kernel<false, false,', 3.0 * flop, 'flop', 'mfma_bf16'))
  out['assign_ms'] = round(ms, 3)
  out['assign_TFLOPs'] = round(flop / ms / 1e9, 1)                      # useful fp32 flops per second
  # information only, NOT a roofline fraction (above 1: the contraction does not run on the fp32 matrix pipe this
  # peak belongs to); the fraction of the pipe it does run on is assign_split.frac_of_bf16_peak
  out['assign_useful_flops_over_fp32_mfma_peak'] = round(flop / ms / 1e9 / MFMA_F32_PEAK_TFLOPS, 3)
  out['assign_split'] = {'instruction': 'v_mfma_f32_32x32x16_bf16 x 3 per 16 features (hi*hi, hi*mid, mid*hi), fp32 accumulate',
                         'issued_TFLOPs': round(3.0 * flop / ms / 1e9, 1), 'bf16_peak_TFLOPs': MFMA_BF16_PEAK_TFLOPS,
                         'frac_of_bf16_peak': round(3.0 * flop / ms / 1e9 / MFMA_BF16_PEAK_TFLOPS, 3)}
  # beside the guide's figure: what this GPU's bf16 pipe delivers when every CU issues nothing but that MFMA
  # (tools/mfma_peak_probe) -- information; frac_of_bf16_peak above stays the roofline fraction
  probed = probed_peaks().get('v_mfma_f32_32x32x16_bf16')
  if probed:
    out['assign_split']['probed_bf16_TFLOPs'] = round(probed, 1)
    out['assign_split']['frac_of_probed_bf16'] = round(3.0 * flop / ms / 1e9 / probed, 3)
  del prepared
  ms = event_time(lambda: kernels.nearest_center(x, cdev, labels), 10, warmup=2)
  out['assign_standalone_ms'] = round(ms, 3)                            # + cutting the points (once per call)
  kernels.nearest_center(x, cdev, labels, _hip.NEAREST_SPLIT_UNCHECKED)
  out['assign_rechecked_points'] = int((labels < 0).sum().item())       # re-decided exactly (fp64 cdist)
  # the fp32-MFMA filter of rounds 2-4 (SP_NEAREST_FUSED; SP_KM_SPLIT=0 makes it the default again)
  ms = event_time(lambda: kernels.nearest_center(x, cdev, labels, _hip.NEAREST_FUSED), 10, warmup=2,
                  section=('k-means assign (first pass, fp32 MFMA tier)', 'sp_nearest_nt_kernel<true, false, false>', flop, 'flop', 'mfma'))
  kernels.nearest_center(x, cdev, labels, _hip.NEAREST_FUSED_UNCHECKED)
  out['assign_fp32_tier'] = {'ms': round(ms, 3), 'TFLOPs': round(flop / ms / 1e9, 1),
                             'frac_of_fp32_mfma_peak': round(flop / ms / 1e9 / MFMA_F32_PEAK_TFLOPS, 3),
                             'rechecked_points': int((labels < 0).sum().item())}
  kernels.nearest_center(x, cdev, labels)
  sums = D.empty((k, d), np.float32)
  counts = D.empty((k,), np.int64)

  def accumulate():
    # counts and per-cluster sums of one assignment, as a fit's iteration gets them: ONE call (sp_segment_sum_counts:
    # the counting sort inside has the counts), 5 launches -- histogram, column scan, rank, segment sums, combine
    kernels.segment_sum(x, labels, k, sums, counts)
  ms = event_time(accumulate, 10, section=('k-means segment sums', 'sp_segment_sum_kernel', 4.0 * n * d, 'bytes', 'hbm'))
  out['accumulate_ms'] = round(ms, 3)
  out['accumulate_GBps'] = round(4.0 * n * d / ms / 1e6, 1)             # 4*N*D bytes
  out['accumulate_launches'] = 5

  def accumulate_two_calls():            # (rounds 1-5: sp_bincount_i64 + sp_segment_sum, 7 launches with the memset)
    kernels.bincount(labels, k, counts)
    kernels.segment_sum(x, labels, k, sums)
  out['accumulate_two_calls_ms'] = round(event_time(accumulate_two_calls, 10), 3)
  # the benchmark as the reference runs it (tests/benchmark_kmeans.py -> KMeans(k, n_iter).fit): ONE fit of 10
  # iterations, after one of 2 -- inside a fit the centers stay on the worker between iterations (k_means_.py)
  c, _ = KMeans(k, 2).fit(Xv, centers, implementation='map2', reducer=np.add)
  D.synchronize()
  t0 = time.perf_counter()
  c, _ = KMeans(k, 10).fit(Xv, c, implementation='map2', reducer=np.add)   # (returns the centers as a host array)
  D.synchronize()
  dt = time.perf_counter() - t0
  out['ten_iterations_ms'] = round(dt * 1e3, 2)
  out['iteration_ms'] = round(dt * 1e2, 3)                             # driver program end to end, mean of the 10
  out['iterations'] = {'timed': 10, 'warmup': 2, 'how': 'one KMeans(k, 10).fit after one KMeans(k, 2).fit'}
  out['centers_finite'] = bool(np.isfinite(c).all())
  return out


def sparse_section(ctx):
  """the sparse multiply of tests/benchmark_pagerank.py on one worker's tile -- 900 000 pages,
  10 out-links per page, 90 % of them inside one of 8 sites -- p <- W . p with W a device CSR tile.
  HBM-bound: 8 B per stored entry (value + column index) + 16 B per row (indptr, y, x read once)."""
  from spartan_amd import sparse as S
  n, deg, sites = 900000, 10, 8
  rng = np.random.RandomState(SEED + 31)
  cols = np.repeat(np.arange(n, dtype=np.int64), deg)
  local = (cols // (n // sites)) * (n // sites) + rng.randint(0, n // sites, size=n * deg)
  far = rng.randint(0, n, size=n * deg)
  rows = D.from_numpy(np.where(rng.rand(n * deg) <= 0.9, local, far).astype(np.int32))
  cols = D.from_numpy(cols.astype(np.int32))
  vals = D.full((n * deg,), 1, np.float32)
  out = {'tile': '%dx%d fp32 CSR, %d links per page' % (n, n, deg)}
  ms = event_time(lambda: S.from_coo((n, n), np.float32, rows, cols, vals), 3, warmup=1)
  out['coo_to_csr_ms'] = round(ms, 3)
  W = S.from_coo((n, n), np.float32, rows, cols, vals)
  del rows, cols, vals, local, far
  x = D.from_numpy(rng.rand(n, 1).astype(np.float32))
  y = D.empty((n, 1), np.float32)
  alg = W.nnz * 8 + n * 16
  D.synchronize()
  t0 = time.perf_counter()
  blocked = S.spmv_block_plan(W) is not False      # the column-blocked copy of the entries: once per tile
  D.synchronize()
  out['spmv_block_plan_ms'] = round((time.perf_counter() - t0) * 1e3, 3)
  out['spmv_kernel'] = 'sp_bsp_spmv_kernel' if blocked else 'sp_csr_spmv_planned_kernel'
  ms = event_time(lambda: S.spmm(W, x, out=y), 20, warmup=3, section=('CSR x vector 900k', out['spmv_kernel'], float(alg), 'bytes', 'hbm'))
  out.update({'nnz': W.nnz, 'spmv_ms': round(ms, 4), 'spmv_GBps': round(alg / ms / 1e6, 1),
              'spmv_bytes_per_launch': alg})
  # the driver program: 5 iterations of p = dot(wts, p) through the expression API on the same tile
  wts = sp.from_tile_fn((n, n), np.float32, lambda ex: W, sparse=True).force()
  p = sp.from_tile_fn((n, 1), np.float32, lambda ex: x).force()
  t = []
  for _ in range(3):
    D.synchronize()
    t0 = time.perf_counter()
    q = sp.Val(val=p)
    for _ in range(5):
      q = sp.dot(sp.Val(val=wts), q).optimized()
    q.force()
    D.synchronize()
    t.append(time.perf_counter() - t0)
  out['five_iterations_ms'] = round(min(t[1:]) * 1e3, 3)
  return out


def dist_section(ctx, p=None):
  """N > 1 (p = None) or ONE GPU hosting p logical workers: BASELINE configs[2] -- an array of N row tiles of 8192 x 65536 fp32 (2 GiB per GPU), reduced along
  every axis through the expression API: the per-tile kernels plus the RCCL combine (reduce to the owner for
  axis=None, reduce-scatter for axis=0, nothing for axis=1 / argmax axis=1 / the fused map).  GB/s = whole-job
  algorithmic bytes / wall-clock, barrier + synchronize on both sides, max over ranks."""
  p = p or ctx.world.size
  R = int(os.environ.get('SPARTAN_BENCH_DIST_ROWS', '8192'))
  C = 65536
  X = sp.from_tile_fn((R * p, C), np.float32, lambda ex: device_uniform(ex, 0.0, 1.0, SEED + 41), tile_hint=(R, C)).force()
  Xv = sp.Val(val=X)
  E = float(R) * p * C
  out = {'array': '%d x %d fp32, %d row tiles of %d x %d' % (R * p, C, p, R, C)}
  keep = []
  progs = (('sum_axisNone', lambda: sp.sum(Xv), 4.0), ('sum_axis0', lambda: sp.sum(Xv, 0), 4.0),
           ('sum_axis1', lambda: sp.sum(Xv, 1), 4.0), ('argmax_axisNone', lambda: sp.argmax(Xv), 4.0),
           ('argmax_axis0', lambda: sp.argmax(Xv, 0), 4.0), ('argmax_axis1', lambda: sp.argmax(Xv, 1), 4.0),
           ('map_xx_plus_x', lambda: (Xv * Xv + Xv).optimized(), 8.0))
  for name, build, bpe in progs:
    def step():
      keep[:] = [build().force()]
    dt = time_steps(ctx, step, 10, 3)
    out[name + '_GBps'] = round(bpe * E * 10 / dt / 1e9, 1)
    del keep[:]
  return out


def lreg_dist_section(ctx, p=None, steps=None):
  """N > 1 (p = None) or one GPU hosting p logical workers: BASELINE configs[4] -- X 1 000 000 x 4096 fp32 row-tiled over the N GPUs (strong scaling: rows / N per
  GPU), 100 gradient steps after 2 untimed ones through examples/lreg.fit: per step one pass over the rank's rows,
  the (D,) partial gradients combined by reduce-scatter + all-gather (glom), w updated on every rank's driver."""
  from spartan_amd.examples import lreg
  p = p or ctx.world.size
  N = int(os.environ.get('SPARTAN_BENCH_LREG_ROWS', '1000000'))
  Dm = int(os.environ.get('SPARTAN_BENCH_LREG_COLS', '4096'))
  N -= N % p
  hint = (N // p, Dm)
  Xl = sp.from_tile_fn((N, Dm), np.float32, lambda ex: device_uniform(ex, 0.0, 1.0, SEED + 11), tile_hint=hint).force()
  yl = sp.from_tile_fn((N, 1), np.float32, lambda ex: device_uniform(ex, 0.0, 1.0, SEED + 12), tile_hint=(N // p, 1)).force()
  Xv, yv = sp.Val(val=Xl), sp.Val(val=yl)
  w = ctx.world.broadcast_object(np.random.RandomState(SEED).rand(Dm, 1).astype(np.float32), 0)
  alpha = 1e-11
  w = lreg.fit(Xv, yv, 2, alpha=alpha, w=w)
  steps = steps or int(os.environ.get('SPARTAN_BENCH_LREG_STEPS', '100'))
  box = [w]

  def run():
    box[0] = lreg.fit(Xv, yv, 1, alpha=alpha, w=box[0])
  dt = time_steps(ctx, run, steps, 0)
  return {'array': '%d x %d fp32, %d row tiles of %d rows (configs[4], strong scaling)' % (N, Dm, p, N // p),
          'steps': steps, 'warmup_steps': 2, 'ms_per_step': round(dt / steps * 1e3, 4),
          'two_pass_equivalent_GBps': round(steps * 2 * 4.0 * N * Dm / dt / 1e9, 1),
          'streamed_GBps': round(steps * 4.0 * N * Dm / dt / 1e9, 1),
          'weights_finite': bool(np.isfinite(box[0]).all())}


def kmeans_dist_section(ctx, p=None):
  """N > 1 (p = None) or one GPU hosting p logical workers: BASELINE configs[3] -- 10 000 000 x 256 fp32 points row-tiled over the N GPUs, k = 1024: 10 timed
  Lloyd iterations after 2 untimed ones through KMeans.fit ('map2', reducer np.add): assign + accumulate per tile,
  counts and sums combined across the ranks, centers re-derived on every rank's driver."""
  from spartan_amd.examples.sklearn.cluster import KMeans
  p = p or ctx.world.size
  n = int(os.environ.get('SPARTAN_BENCH_KMEANS_POINTS', '10000000'))
  k, d = int(os.environ.get('SPARTAN_BENCH_KMEANS_K', '1024')), 256
  n -= n % p
  X = sp.from_tile_fn((n, d), np.float32, lambda ex: device_uniform(ex, 0.0, 1.0, SEED + 21), tile_hint=(n // p, d)).force()
  Xv = sp.Val(val=X)
  box = [ctx.world.broadcast_object(np.random.RandomState(SEED).rand(k, d), 0)]
  box[0], _ = KMeans(k, 2).fit(Xv, box[0], implementation='map2', reducer=np.add)

  def run():          # ONE fit of 10 iterations (the centers stay on the workers in between)
    box[0], _ = KMeans(k, 10).fit(Xv, box[0], implementation='map2', reducer=np.add)
  dt = time_steps(ctx, run, 1, 0)
  return {'array': '%d x %d fp32 points, k=%d, %d row tiles of %d rows (configs[3], strong scaling)' % (n, d, k, p, n // p),
          'iterations': {'timed': 10, 'warmup': 2}, 'iteration_ms': round(dt * 1e2, 3),
          'assign_TFLOPs_whole_job_incl_everything': round(10 * 2.0 * n * k * d / dt / 1e12, 1)}


def collectives_section(ctx):
  """N > 1: every collective of the data plane timed ONCE PER KIND on the message size the workloads give it, before
  the pipeline runs -- so that the first multi-GPU run says which exchange is slow (or hangs: the section runs under
  `guarded`) instead of only a low headline.  Per kind: [ms, GB/s one GPU sends (its share of the algorithm's
  traffic / time), fraction of the 7 x 153 GB/s a GPU's xGMI links carry together].  Sizes are those of the north
  star at p ranks: reduce-scatter of a 512 MiB partial chunk (dot.ksplit_pipeline), all-to-all of 64 MiB A blocks,
  all-gather of a (4096,) gradient (lreg), broadcast of a 16 KiB centre block; on the staged debug transport
  (ranks sharing a GPU) everything is 1/64 of that.  Stands where the reference sends UpdateReq / GetReq messages
  over ZeroMQ (spartan/array/distarray.py:372-422, blob_ctx.py:163-179)."""
  world = ctx.world
  p = world.size
  real = getattr(world.transport, 'name', '') == 'rccl'
  scale = 1 if real else 64
  MiB = 1 << 20
  out = {'is': '{kind: [bytes per GPU message, ms, GB/s sent per GPU, fraction of 7 links]}',
         'links_GBps': {'one': XGMI_LINK_GBPS, 'seven': 7 * XGMI_LINK_GBPS}, 'transport': world.note or '?',
         'message_scale': '1' if real else '1/64 (staged transport)'}

  def timed(fn, reps=3):
    fn()                                        # (first call: communicator channels, staging buffers)
    return time_steps(ctx, fn, reps, 0) / reps

  def entry(name, nbytes, sent, dt):
    gbps = sent / dt / 1e9
    out[name] = [int(nbytes), round(dt * 1e3, 3), round(gbps, 1), round(gbps / (7 * XGMI_LINK_GBPS), 3)]
  # reduce-scatter: every rank holds a 512 MiB partial, keeps the sum of its 1/p
  n = 512 * MiB // scale // 4 // p * p
  inp, res = D.full((n,), 1.0, np.float32), D.empty((n // p,), np.float32)
  entry('reduce_scatter', n * 4, n * 4 * (p - 1) / p, timed(lambda: world.reduce_scatter(res, inp, 'ADD')))
  ok = bool((res[:4].numpy() == p).all())
  del inp, res
  # all-to-all: one 64 MiB block to and from every other rank, one grouped launch
  b = 64 * MiB // scale // 4
  send = [D.full((b,), float(world.rank), np.float32) for _ in range(p)]
  recv = [D.empty((b,), np.float32) for _ in range(p)]
  sends = [(r, send[r]) for r in range(p) if r != world.rank]
  recvs = [(r, recv[r]) for r in range(p) if r != world.rank]
  entry('all_to_all', b * 4, b * 4 * (p - 1), timed(lambda: world.exchange(sends, recvs)))
  ok = ok and all(float(recv[r][:1].numpy()[0]) == r for r in range(p) if r != world.rank)
  del send, recv, sends, recvs
  # all-gather of the (4096,) pieces of a gradient; broadcast of 16 KiB
  g, gall = D.full((4096 // p * p // p,), 1.0, np.float32), D.empty((4096 // p * p,), np.float32)
  entry('all_gather', g.nbytes, g.nbytes * (p - 1), timed(lambda: world.all_gather_into(gall, g), reps=20))
  small = D.full((4096,), 3.0, np.float32)
  entry('broadcast', small.nbytes, small.nbytes, timed(lambda: world.broadcast(small, 0), reps=20))
  out['values_ok'] = ok
  D.trim_pool()
  return out


def rccl_report(world):
  """What carried the tile payloads between the ranks: `ranks` is the size of the RCCL communicator every rank
  joined and self-tested (0 when the job ran on the staged debug transport)."""
  import ctypes
  t = world.transport
  v = ctypes.c_int(0)
  version = None
  try:
    if _hip.lib().sp_comm_available() and _hip.lib().sp_comm_version(ctypes.byref(v)) == 0:
      version = v.value
  except Exception:
    pass
  from spartan_amd import comm
  rep = {'ranks': 0, 'version': version, 'visible_gpus': comm_gpu_count(),
         'control_plane': getattr(world.control, 'name', None), 'torch_in_process': 'torch' in sys.modules}
  try:
    rep.update(comm.rccl_paths())            # lib_path, hip_runtime_path (RCCL's), own_hip_runtime_path
  except Exception as e:
    rep['lib_path'] = rep['hip_runtime_path'] = None
    rep['paths_error'] = str(e)[:300]
  rep['mapped'] = comm.mapped_runtimes()     # one file per library when the process is single-runtime
  if getattr(t, 'name', '') == 'rccl':
    rep.update({'ranks': t.size, 'self_test': 'passed'})
  else:
    rep['self_test'] = 'not run: transport is %s (ranks share devices)' % getattr(t, 'name', '?')
  return rep


def comm_gpu_count():
  from spartan_amd import comm
  return comm.gpu_count()


def guarded(fn, timeout_s, rank, fallback_line):
  """Run an informational section under a deadline.  If it does not come back (a collective that never completes)
  rank 0 still prints the line it has -- the headline was measured before this section -- with the failure in
  `extras_error`, and every rank exits NON-ZERO: a hung extra is a failed run."""
  import threading
  done = threading.Event()

  def watchdog():
    if not done.wait(timeout_s):
      fallback_line['extras_error'] = 'FAILED: section did not complete within %d s (hung collective?)' % timeout_s
      _emit(fallback_line, rank)
      os._exit(3)
  threading.Thread(target=watchdog, daemon=True).start()
  try:
    res = fn()
  except Exception as e:   # informational: report in the line, do not lose the headline
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
  D.synchronize()
  t0 = clocks()
  step()
  D.synchronize()
  ms = []
  for _ in range(3):
    e0, e1 = D.Event(), D.Event()
    e0.record()
    step()
    e1.record()
    e1.synchronize()
    ms.append(e0.elapsed_ms(e1))
  del keep[:]
  avg = sum(ms) / len(ms)
  flop = 2.0 * n ** 3
  D.synchronize()
  note_section('dot 32768^3 (north star)', GEMM_KERNEL, 4, flop, 'flop', 'mfma', t0, clocks())
  return {'workload': 'spartan.dot %dx%dx%d fp32, one tile' % (n, n, n), 'calls': len(ms),
          'ms_per_call': round(avg, 2), 'min_ms': round(min(ms), 2), 'TFLOPs': round(flop / avg / 1e9, 2),
          'frac_of_mfma_peak': round(flop / avg / 1e9 / MFMA_F32_PEAK_TFLOPS, 4), 'flop_per_call': flop}


# ---- one rank of the p-GPU K-split step, on one GPU ---------------------------------------------------------------
class _SideWork(object):
  def __init__(self, event, keep):
    self.event, self.keep = event, keep

  def wait(self):
    D.current_stream().wait_event(self.event)
    self.keep = None


def ksplit_rank_emulation(ctx, one_gpu_ms):
  """What ONE rank of the p-GPU north-star step does (spartan_amd/expr/dot.ksplit_pipeline, the very function the
  multi-GPU job runs), measured on this GPU for p = 2, 4, 8: its (32768/p x 32768) tile of A, its rows of B, the
  slab buffer, the chunk GEMMs, the pastes -- with the transport replaced by device copies of the SAME byte counts on
  a high-priority side stream: per all-to-all block one copy (a block leaving + a block landing), per
  reduce-scatter (p-1)/p of the chunk copied out and the rank's own piece produced.  The copies run on
  `copy_workgroups` workgroups, the way a collective's channels occupy a bounded share of the CUs.
  Reported per (p, chunk columns): the step with and without the transfers, the GEMM kernels' own time in both,
  and the speed-up over the one-GPU step this rank's time would allow if the wire were no slower than the copies."""
  import importlib
  dot_mod = importlib.import_module('spartan_amd.expr.dot')     # (spartan_amd.expr.dot the ATTRIBUTE is the dot() builder)
  be = ctx.backend
  n = NORTH_STAR
  wg = int(os.environ.get('SPARTAN_EMULATED_COMM_WORKGROUPS', '32'))
  side = D.Stream(high_priority=True)
  out = {'workload': 'one rank of dot %d^3 fp32 row-tiled over p GPUs' % n, 'copy_workgroups': wg,
         'one_gpu_step_ms': round(one_gpu_ms, 2), 'cases': []}
  flop = 2.0 * n ** 3
  for p in (2, 4, 8):
    mb = kb = n // p
    my_a = device_uniform(sp.extent.from_shape((mb, n)), -1.0, 1.0, SEED + 61)
    my_b = device_uniform(sp.extent.from_shape((kb, n)), -1.0, 1.0, SEED + 62)
    scratch = D.empty((n, 8192), np.float32)
    for nc in (2048, 4096, 8192):
      moved = {'bytes': 0}

      def exchange(sends, recvs, on=True):
        if not on:
          return None
        side.wait_stream(D.current_stream())
        for (_, src), (_, dst) in zip(sends, recvs):
          kernels.stream_copy(dst, src, max_workgroups=wg, stream=side)
          moved['bytes'] += src.nbytes
        return _SideWork(D.Event().record(side), [t for _, t in sends] + [t for _, t in recvs])

      def reduce_scatter(piece, part, on=True):
        if not on:
          return None
        side.wait_stream(D.current_stream())
        away = part.nbytes // p * (p - 1)                    # what this rank sends: every row block but its own
        kernels.stream_copy(scratch, part, nbytes=away, max_workgroups=wg, stream=side)
        kernels.stream_copy(piece, part, nbytes=piece.nbytes, max_workgroups=wg, stream=side)   # its reduced piece
        moved['bytes'] += away
        return _SideWork(D.Event().record(side), [piece, part])
      res = {}
      for mode in ('compute_only', 'with_transfers'):
        on = mode == 'with_transfers'
        step_ms, gemm_ms = [], []
        for it in range(3):
          moved['bytes'] = 0
          be.gemm_events = []
          e0, e1 = D.Event(), D.Event()
          D.synchronize()
          e0.record()
          result = dot_mod.ksplit_pipeline(be, p, 1 % p, my_a, my_b, np.dtype(np.float32),
                                           lambda s, r: exchange(s, r, on), lambda o, q: reduce_scatter(o, q, on), nc)
          e1.record()
          D.synchronize()
          if it:
            step_ms.append(e0.elapsed_ms(e1))
            gemm_ms.append(sum(a.elapsed_ms(b) for (a, b, _, _, _) in be.gemm_events))
          del result
        be.gemm_events = None
        res[mode] = (sum(step_ms) / len(step_ms), sum(gemm_ms) / len(gemm_ms))
      step_on, gemm_on = res['with_transfers']
      step_off, gemm_off = res['compute_only']
      out['cases'].append({
          'p': p, 'chunk_cols': nc, 'gemms_per_step': (n // nc - 1) + 3,
          'step_ms': round(step_on, 2), 'step_ms_compute_only': round(step_off, 2),
          'gemm_kernels_ms': round(gemm_on, 2), 'gemm_kernels_ms_compute_only': round(gemm_off, 2),
          'gemm_TFLOPs_compute_only': round(flop / p / gemm_off / 1e9, 1),
          'gemm_frac_of_mfma_peak_compute_only': round(flop / p / gemm_off / 1e9 / MFMA_F32_PEAK_TFLOPS, 3),
          'gemm_slowdown_under_transfers': round(gemm_on / gemm_off, 3),
          'transfer_bytes_per_step': moved['bytes'],
          'implied_speedup_bound': round(one_gpu_ms / step_on, 2)})
    del my_a, my_b, scratch
    D.trim_pool()
  best8 = max((c for c in out['cases'] if c['p'] == 8), key=lambda c: c['implied_speedup_bound'])
  out['implied_8gpu_speedup_upper_bound'] = best8['implied_speedup_bound']
  out['best_chunk_cols_at_p8'] = best8['chunk_cols']
  out['note'] = ('an UPPER bound, not a scaling measurement: link bandwidth and RCCL latency are not modelled, only '
                 'the CUs and HBM bandwidth a concurrent transfer takes from the GEMMs and what the pipeline leaves '
                 'exposed; target >= 6x at 8 GPUs')
  return out


def _host_memory_gib():
  """(available, total) host RAM in GiB (MemAvailable of /proc/meminfo)."""
  info = {}
  try:
    for line in open('/proc/meminfo'):
      k, v = line.split(':', 1)
      info[k] = float(v.split()[0]) / (1 << 20)
  except (IOError, OSError, ValueError):
    return 0.0, 0.0
  return info.get('MemAvailable', 0.0), info.get('MemTotal', 0.0)


def cpu_baseline():
  """The reference's execution model on the host cores of this box: W = min(physical
  cores, 64) worker processes, one per core, pinned, one BLAS thread each (spartan/worker.py:40,385-387), running
  the oracle's NumPy tile bodies ON THE BASELINE SHAPES -- configs[1] whole, configs[2] whole, configs[4] whole
  for 3 steps, configs[3] on one per-GPU tile for 2 iterations (the bounded sample: the whole array is ~45 s per
  iteration) -- scaled down only where the host's free memory forces it (the reason is printed); the parent
  merges like the owner of the target tile.  A reported baseline, not a target."""
  from oracle import cpu_workers
  t_all = time.perf_counter()
  avail, total = _host_memory_gib()
  avail = avail or 32.0               # (no /proc/meminfo: assume a small host)
  scaled = []
  pool = cpu_workers.Workers(64)
  try:
    W = pool.count
    # configs[1]: dot 8192^3, K-split over W workers, one target tile (dot.py:277-290): W partials of 256 MiB stay
    # in the workers, 4 ring slots + the target with the owner
    n = 8192
    need = (W + 5) * n * n * 4 / float(1 << 30) + 1.0
    if avail and need > 0.6 * avail:
      while n > 1024 and (W + 5) * n * n * 4 / float(1 << 30) + 1.0 > 0.6 * avail:
        n //= 2
      scaled.append('dot: %d^3 instead of 8192^3 (%.0f GiB free, the K-split partials of %d workers need %.0f GiB)'
                    % (n, avail, W, need))
    t_dot, t_dot_compute, n = pool.dot_shared(n)
    # configs[2]: 65536 x 65536 fp32 = 16 GiB of input, x*x+x keeps a temporary and a result per tile
    rows, cols = 65536, 65536
    need = 3.2 * rows * cols * 4 / float(1 << 30)
    if avail and need > 0.6 * avail:
      while rows > 4096 and 3.2 * rows * cols * 4 / float(1 << 30) > 0.6 * avail:
        rows //= 2
      scaled.append('map / sum: %d x %d instead of 65536 x 65536 (%.0f GiB free, %.0f GiB needed)' % (rows, cols, avail, need))
    t_map, t_sum, rows = pool.map_and_sum(rows, cols)
    # configs[4]: the WHOLE 1 000 000 x 4096 array (16 GB over the W workers), 3 gradient steps
    ln, ld, lsteps = 1000000, 4096, 3
    need = 2.2 * ln * ld * 4 / float(1 << 30)
    if avail and need > 0.6 * avail:
      while ln > 125000 and 2.2 * ln * ld * 4 / float(1 << 30) > 0.6 * avail:
        ln //= 2
      scaled.append('lreg: %d rows instead of 1000000 (%.0f GiB free, %.0f GiB needed)' % (ln, avail, need))
    t_lreg, ln = pool.lreg_steps(ln, ld, lsteps)
    # configs[3]: scipy's cdist of the whole 10 000 000 points takes ~45 s per iteration on 64 cores, outside the
    # bounded sample a default run may spend: the per-GPU tile of the 8-GPU run (1 250 000 points), 2 iterations
    kn, kd, kk, kiters = 1250000, 256, 1024, 2
    t_km, kn = pool.kmeans_iterations(kn, kd, kk, kiters)
  finally:
    pool.close()
  e = float(rows) * cols
  end_to_end = 2.0 * n ** 3 / t_dot / 1e12
  gemm_only = 2.0 * n ** 3 / t_dot_compute / 1e12
  return {'value': round(end_to_end, 4), 'unit': 'TFLOP/s', 'cores': W, 'kind': 'port',
          'value_is': 'dot end to end: the W per-worker GEMMs AND the W M x N partials travelling (through a shared-'
                      'memory ring, not pickled pipes) to the owner of the one target tile and being added there '
                      '(dot.py:277-278); the GEMMs alone: gemm_only_value',
          'gemm_only_value': round(gemm_only, 4),
          'workers': '%d processes pinned to %d physical cores, 1 BLAS thread each' % (W, W),
          'dot': {'shape': '%dx%dx%d fp32, K-split over %d workers, one target tile' % (n, n, n, W),
                  'seconds': round(t_dot, 3), 'gemm_seconds': round(t_dot_compute, 3),
                  'TFLOPs_end_to_end': round(end_to_end, 4), 'TFLOPs_gemm_only': round(gemm_only, 4)},
          'map_xx_plus_x_GBps': round(8.0 * e / t_map / 1e9, 2), 'sum_axis0_GBps': round(4.0 * e / t_sum / 1e9, 2),
          'map_sum_shape': '%dx%d fp32 in %d row tiles' % (rows, cols, W),
          'lreg': {'shape': '%dx%d fp32 (whole array of configs[4]) in %d row tiles' % (ln, ld, W), 'steps': lsteps,
                   'seconds_per_step': round(t_lreg, 4), 'GBps': round(2 * 4.0 * ln * ld / t_lreg / 1e9, 2)},
          'kmeans': {'shape': '%dx%d points, k=%d (one per-GPU tile of configs[3])' % (kn, kd, kk), 'iterations': kiters,
                     'seconds_per_iteration': round(t_km, 3), 'TFLOPs_of_2nkd': round(2.0 * kn * kk * kd / t_km / 1e12, 4)},
          'host_memory_GiB': {'available': round(avail, 1), 'total': round(total, 1)},
          'scaled_for_memory': scaled,
          'sample': 'oracle tile bodies (NumPy / BLAS / scipy cdist) on %d pinned one-thread workers: configs[1] dot '
                    '%d^3 K-split, one target tile; configs[2] x*x+x and sum(axis=0) on %dx%d; configs[4] %d lreg '
                    'steps on the WHOLE %dx%d array; configs[3] %d k-means iterations on one per-GPU tile %dx%d, '
                    'k=%d (the whole 10M points: ~45 s per iteration)%s' % (W, n, rows, cols, lsteps, ln, ld, kiters, kn, kd, kk,
                                      '' if not scaled else ' -- scaled for host memory: ' + '; '.join(scaled)),
          'wall_seconds': round(time.perf_counter() - t_all, 1)}


def one_gpu_8_tiles_section(line):
  """The WHOLE arrays of BASELINE configs[2] / [3] / [4] on this one GPU as 8 logical workers (16 GiB, 10 GB and
  16 GB: they fit 288 GB): the per-tile kernels PLUS the combine between tiles (update / merge with the reducer,
  the gather of glom) that the one-tile sections never run -- what the reference's runner does when it sweeps the
  worker count on one machine (tests/test_common.py:98-120).  Rates are whole-array algorithmic bytes / wall-clock;
  `vs_one_tile` divides them by the same program's one-tile rate from the sections above, so the cost of the
  combine is a number the driver sees."""
  sp.shutdown()
  D.trim_pool()
  ctx8 = sp.initialize('hip', num_workers=8)
  out = {'workers': 8}
  try:
    out['hbm'] = dist_section(ctx8, 8)
    D.trim_pool()
    out['lreg'] = lreg_dist_section(ctx8, 8, steps=20)
    D.trim_pool()
    out['kmeans'] = kmeans_dist_section(ctx8, 8)
    D.trim_pool()
    hbm1 = line.get('hbm') or {}
    rel = {}
    for key, one in (('sum_axisNone_GBps', 'sum_axisNone_GBps'), ('sum_axis0_GBps', 'sum_axis0_GBps'),
                     ('sum_axis1_GBps', 'sum_axis1_GBps'), ('argmax_axisNone_GBps', 'argmax_axisNone_GBps'),
                     ('argmax_axis0_GBps', 'argmax_axis0_GBps'), ('argmax_axis1_GBps', 'argmax_axis1_GBps'),
                     ('map_xx_plus_x_GBps', 'map_xx_plus_x_GBps')):
      if key in out['hbm'] and hbm1.get(one):
        rel[key[:-5]] = round(out['hbm'][key] / hbm1[one], 3)
    if (line.get('lreg') or {}).get('GBps'):
      rel['lreg_streamed'] = round(out['lreg']['streamed_GBps'] / line['lreg']['GBps'], 3)
    if (line.get('kmeans') or {}).get('iteration_ms'):
      # 8 tiles of the one-tile section's size: 8 x its iteration time is the no-overhead figure
      rel['kmeans_iteration'] = round(8 * line['kmeans']['iteration_ms'] / out['kmeans']['iteration_ms'], 3)
    out['vs_one_tile'] = rel
  finally:
    sp.shutdown()
    D.trim_pool()
  return out


def measured_traffic(n):
  """HBM bytes per GEMM launch from the PMC passes tools/profile_round.sh took (profiles/roofline_traffic.json),
  quoted only if those passes ran on THIS tree (the kernel sources' hash matches); else None + why."""
  path = os.path.join(ROOT, 'profiles', 'roofline_traffic.json')
  try:
    rec = json.load(open(path))
  except Exception:
    return None, 'no profiles/roofline_traffic.json'
  sha = _hip.source_sha()
  if rec.get('tree_sha') != sha:
    return None, ('profiles/roofline_traffic.json was measured on kernel sources %s, this tree is %s: not quoted'
                  % (rec.get('tree_sha'), sha))
  val = rec.get('gemm_%d' % n, {}).get('traffic_bytes')
  return val, ('profiles/roofline_traffic.json (tree %s): 2*FETCH_SIZE + WRITE_SIZE, separate rocprofv3 --pmc passes '
               'of this kernel and shape; algorithmic floor %d bytes' % (sha, 12 * n * n))


def _free_port():
  import socket
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  port = s.getsockname()[1]
  s.close()
  return port


def self_launch(n, argv, deadline_s):
  """`python3 bench.py --gpus N` as a plain process: start the N ranks ourselves (RANK / LOCAL_RANK / WORLD_SIZE /
  MASTER_ADDR / a free MASTER_PORT -- what torch.distributed.run would set; the reference's bench runner starts its
  own workers too, tests/test_common.py:86-125), hand rank 0's one JSON line through, and exit non-zero with the
  reason in a line of the same shape if any rank fails or the deadline passes.  With fewer visible GPUs than ranks
  the ranks share devices over the staged debug transport and the line says so (`rccl.ranks` = 0): a functional
  run, not a scaling measurement."""
  import subprocess
  import tempfile
  from spartan_amd import comm
  gpus = comm.gpu_count()
  port = _free_port()
  base = dict(os.environ)
  base.update({'WORLD_SIZE': str(n), 'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port),
               'SPARTAN_BENCH_LAUNCHER': 'self', 'SPARTAN_BENCH_VISIBLE_GPUS': str(gpus)})
  if gpus < n and 'SPARTAN_DIST_BACKEND' not in base:
    base['SPARTAN_DIST_BACKEND'] = 'socket'      # RCCL wants one device per rank
  out0 = tempfile.TemporaryFile()
  procs = []
  for rank in range(n):
    env = dict(base, RANK=str(rank), LOCAL_RANK=str(rank))
    procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                  stdout=out0 if rank == 0 else subprocess.DEVNULL))
  t_end = time.time() + deadline_s
  why = None
  while why is None:
    codes = [q.poll() for q in procs]
    bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
    if bad:
      why = 'rank %d exited with code %d' % bad[0]
    elif all(c == 0 for c in codes):
      break
    elif time.time() > t_end:
      why = 'deadline of %d s passed (ranks still running: %s)' % (deadline_s, [r for r, c in enumerate(codes) if c is None])
    else:
      time.sleep(0.05)
  if why is not None:
    time.sleep(1.0)                       # a rank that saw its peer die reports on its own
    for q in procs:
      if q.poll() is None:
        q.kill()                          # the exact children started above
    for q in procs:
      q.wait()
  out0.seek(0)
  text = out0.read().decode('utf-8', 'replace').strip()
  last = text.splitlines()[-1] if text else ''
  if why is None:
    try:
      json.loads(last)
    except ValueError:
      why = 'rank 0 printed no JSON line'
  if why is None:
    os.write(1, (last + '\n').encode())
    return 0
  line = {'metric': 'spartan.dot TFLOP/s (+ map/reduce HBM GB/s)', 'value': None, 'unit': 'TFLOP/s', 'n_gpus': n,
          'higher_is_better': True, 'error': 'FAILED: ' + why, 'launcher': 'bench.py self-launch, %d ranks on %d visible GPUs' % (n, gpus)}
  try:
    part = json.loads(last)               # a watchdog line from rank 0 (headline measured, an extra hung)
    part.setdefault('extras_error', line['error'])
    part['error'] = line['error']
    line = part
  except ValueError:
    pass
  os.write(1, (json.dumps(line) + '\n').encode())
  return 1


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


def _detail_path(p):
  """Where the detailed record of this run goes: gpurun_out/ when the tree has one (it is merged back from a GPU
  box), else the system temp directory.  tools/roofline.py and profiles/rNN_bench_n1.json are fed from this file."""
  if os.environ.get('SP_BENCH_DETAIL'):
    return os.environ['SP_BENCH_DETAIL']
  base = os.path.join(ROOT, 'gpurun_out')
  if not os.path.isdir(base):
    import tempfile
    base = tempfile.gettempdir()
  return os.path.join(base, 'bench_detail_n%d.json' % p)


def _emit(record, rank):
  """The ONE line of stdout, from rank 0: tools/bench_line.compact(record), bounded at bench_line.LIMIT bytes.
  The detailed record (launch table with clock readings, every emulation case, the raw sections) goes to a side
  file and to stderr -- a 21 KB stdout line was not parsed by the driver."""
  sys.stdout.flush()
  if rank != 0:
    return
  record = dict(record)
  record['profile_table'] = PROFILE_TABLE
  path = _detail_path(record.get('n_gpus') or 1)
  try:
    with open(path, 'w') as f:
      json.dump(record, f)
    record['detail'] = os.path.relpath(path, ROOT) if path.startswith(ROOT) else path
  except (IOError, OSError) as e:
    record['detail'] = 'not written: %s' % e
  sys.stderr.write('bench detail: ' + json.dumps(record) + '\n')
  sys.stderr.flush()
  line = bench_line.compact(record)
  text = json.dumps(line)
  assert len(text) < bench_line.LIMIT, len(text)
  os.write(_REAL_STDOUT if _REAL_STDOUT is not None else 1, (text + '\n').encode())


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=20)
  ap.add_argument('--warmup', type=int, default=3)
  ap.add_argument('--size', type=int, default=0, help='matrix order (default: 8192 on one GPU, 32768 on several)')
  ap.add_argument('--no-extras', action='store_true', help='headline only: skip the HBM / workload / emulation / CPU sections')
  ap.add_argument('--only', default='', help='comma-separated extras to run (northstar,hbm,dot_f64,gemm_shapes,host,lreg,kmeans,sparse,ksplit,tiles8,cpu; N > 1: hbm_dist,lreg_dist,kmeans_dist)')
  ap.add_argument('--deadline', type=int, default=1500, help='seconds the self-launched ranks of --gpus N > 1 may take')
  args = ap.parse_args()

  if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
    sys.exit(self_launch(args.gpus, sys.argv[1:], args.deadline))      # plain `python3 bench.py --gpus N`
  _claim_stdout()
  world = sp.World.from_env()
  if world.size != args.gpus:
    raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world.size))
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
    scaling = None                 # one GPU: neither weak nor strong
  else:
    if n % p:
      raise SystemExit('--size %d is not a multiple of --gpus %d' % (n, p))
    hint = (n // p, n)
    A = sp.from_tile_fn((n, n), np.float32, lambda ex: device_uniform(ex, -1.0, 1.0, SEED), tile_hint=hint)
    B = sp.from_tile_fn((n, n), np.float32, lambda ex: device_uniform(ex, -1.0, 1.0, SEED + 1), tile_hint=hint)
    workload = ('spartan.dot %dx%dx%d fp32 (north star), A, B and the result row-tiled %d x (%dx%d)'
                % (n, n, n, p, n // p, n))
    parallelism = ('K-split map2 join, 1 worker per GPU: all-to-all of A blocks into one slab, one GEMM per column '
                   'chunk, one asynchronous reduce-scatter per chunk behind the next chunk (transport: %s)'
                   % (world.note or getattr(world.transport, 'name', '?')))
    scaling = 'strong'
  A.force()
  B.force()
  collectives = None
  if p > 1 and not args.no_extras:
    # every collective once, on its real message size, BEFORE the pipeline (a slow or hanging exchange shows here)
    collectives = guarded(lambda: collectives_section(ctx), 240, world.rank,
                          {'metric': 'spartan.dot TFLOP/s (+ map/reduce HBM GB/s)', 'value': None, 'unit': 'TFLOP/s',
                           'n_gpus': p, 'higher_is_better': True, 'config': {'workload': workload},
                           'error': 'FAILED: the collectives self-timing before the pipeline did not complete'})

  keep = []

  def step():
    keep[:] = [sp.dot(A, B, tile_hint=hint).force()]

  # set-up launches (untimed, before the W warm-up steps): library load, allocator warm-up, and the
  # device's one-off dispatch stall (~30 ms, seen once per process about 50 ms into the first sustained
  # MFMA load on these boxes: profiles/r01_notes.md) -- so neither lands in the timed steps
  setup = SETUP_LAUNCHES if p == 1 else 2
  D.synchronize()
  t_head = clocks()
  for _ in range(setup):
    step()
  D.synchronize()
  ctx.backend.gemm_events = []
  stats0 = dict(world.stats)
  dt = time_steps(ctx, step, args.steps, args.warmup)
  D.synchronize()
  stats1 = dict(world.stats)
  all_events = ctx.backend.gemm_events
  ctx.backend.gemm_events = None
  # launches per step: 1 on one GPU; 3 + (chunks - 1) on several (dot.ksplit_pipeline)
  per_step = max(1, len(all_events) // (args.steps + args.warmup))
  events = all_events[-args.steps * per_step:]
  kernel_ms = [e0.elapsed_ms(e1) for (e0, e1, _, _, _) in events]
  flops_timed = sum(2.0 * m_ * n_ * k_ for (_, _, m_, n_, k_) in events)
  achieved = flops_timed / (sum(kernel_ms) * 1e-3) / 1e12
  avg_ms = sum(kernel_ms) / len(kernel_ms)
  flop_step = 2.0 * n * n * n
  if p == 1:
    note_section('dot %d^3 (headline)' % n, GEMM_KERNEL, setup + args.warmup + args.steps, flop_step, 'flop', 'mfma', t_head, clocks())
  value = flop_step * args.steps / dt / 1e12
  line = {
      'metric': 'spartan.dot TFLOP/s (+ map/reduce HBM GB/s)', 'value': round(value, 2), 'unit': 'TFLOP/s',
      'n_gpus': p, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3),
      'higher_is_better': True, 'scaling': scaling, 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
      'config': {'workload': workload, 'parallelism': parallelism, 'flop_per_step': flop_step,
                 'setup_launches': setup,
                 'inputs': 'uniform[-1,1) fp32 generated on device (sp_random_fill, Philox), resident in HBM',
                 'host': 'no torch in this process' if 'torch' not in sys.modules else 'torch is loaded in this process'},
      'roofline': {'bound': 'mfma', 'kernel': 'sp_gemm_glds_kernel<256x128x16, 4 waves> (v_mfma_f32_32x32x2_f32, k-tiles by global_load_lds)',
                   'achieved': round(achieved, 2), 'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                   'frac': round(achieved / MFMA_F32_PEAK_TFLOPS, 4),
                   'flop_per_launch': flops_timed / len(events), 'avg_launch_ms': round(avg_ms, 4),
                   'launches_per_step': per_step, 'traffic': None},
  }
  line['roofline']['traffic'], line['roofline']['traffic_source'] = measured_traffic(n)
  if collectives is not None:
    line['collectives'] = collectives
  if world.rank == 0:
    # the headline as soon as it is measured (stderr; stdout keeps its ONE line, printed at the end or by the
    # watchdog of a section that hangs)
    sys.stderr.write('bench headline: %s\n' % json.dumps({k: line[k] for k in ('value', 'unit', 'n_gpus', 'ms_per_step')}))
    sys.stderr.flush()
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
                'reduce-scatters run on a communication stream behind the GEMM of the next chunk',
    }
  only = set(x for x in args.only.split(',') if x)
  want = lambda name: not args.no_extras and (not only or name in only)   # noqa: E731
  if world.rank == 0 and p == 1:
    del keep[:]
    D.trim_pool()
    if want('northstar'):
      line['northstar_%d' % NORTH_STAR] = northstar_section(ctx)
      D.trim_pool()
    if want('hbm'):
      line['hbm'] = hbm_section(ctx)
      D.trim_pool()
    if want('dot_f64'):
      line['dot_f64'] = dot_f64_section(ctx)
      D.trim_pool()
    if want('gemm_shapes'):
      line['roofline']['gemm_shapes'] = gemm_shapes_section(ctx)
      D.trim_pool()
    if want('host'):
      line['host'] = host_section(ctx)
    if want('lreg'):
      line['lreg'] = lreg_section(ctx, line.get('hbm', {}).get('stream_copy_GBps', 6400.0))
      D.trim_pool()
    if want('kmeans'):
      line['kmeans'] = kmeans_section(ctx)
      D.trim_pool()
    if want('sparse'):
      line['sparse'] = sparse_section(ctx)
      D.trim_pool()
    if want('ksplit'):
      one = line.get('northstar_%d' % NORTH_STAR, {}).get('ms_per_call') or 2.0 * NORTH_STAR ** 3 / (value * 1e9)
      line['ksplit_rank_emulation'] = ksplit_rank_emulation(ctx, one)
      D.trim_pool()
    if want('cpu'):
      line['cpu_baseline'] = cpu_baseline()
    tiles8 = want('tiles8')
    # the second half of the metric and the north-star shape, inside `roofline` (the object the driver keeps whole):
    # every HBM-bound section as GB/s and as a fraction of the copy rate measured in this run and of the 8 TB/s spec
    ns = line.get('northstar_%d' % NORTH_STAR)
    if ns and 'ms_per_call' in ns:
      line['roofline']['northstar'] = {'workload': ns['workload'], 'ms': ns['ms_per_call'], 'TFLOPs': ns['TFLOPs'],
                                       'frac': ns['frac_of_mfma_peak']}
    hbm = line.get('hbm')
    if hbm and 'stream_copy_GBps' in hbm:
      copy = hbm['stream_copy_GBps']
      sections = {k[:-5]: {'GBps': v, 'frac_of_measured_copy': round(v / copy, 3), 'frac_of_spec': round(v / HBM_PEAK_GBPS, 3)}
                  for k, v in hbm.items() if k.endswith('_GBps') and k not in ('stream_copy_GBps', 'hbm_peak_GBps')}
      for name in ('lreg', 'kmeans', 'sparse'):
        sec = line.get(name) or {}
        for key, label in (('GBps', name + '_driver_loop'), ('step_kernels_GBps', name + '_step_kernels'),
                           ('accumulate_GBps', name + '_accumulate'), ('spmv_GBps', name + '_spmv')):
          if key in sec:
            sections[label] = {'GBps': sec[key], 'frac_of_measured_copy': round(sec[key] / copy, 3),
                               'frac_of_spec': round(sec[key] / HBM_PEAK_GBPS, 3)}
      line['roofline']['hbm_sections'] = {'measured_copy_GBps': copy, 'spec_GBps': HBM_PEAK_GBPS, 'sections': sections}
    km = line.get('kmeans')
    if km and 'assign_TFLOPs' in km:
      split = km.get('assign_split') or {}
      line['roofline']['kmeans_assign'] = {'bound': 'mfma_bf16', 'achieved': split.get('issued_TFLOPs'), 'peak': MFMA_BF16_PEAK_TFLOPS,
                                           'unit': 'TFLOP/s', 'frac': split.get('frac_of_bf16_peak'), 'ms': km['assign_ms'],
                                           'frac_is': 'bf16 flops the kernel ISSUES (3 x 2nkd: three exact-product MFMAs per 16 features) over the dense bf16 matrix peak',
                                           'useful_fp32_TFLOPs': km['assign_TFLOPs'],
                                           'useful_flops_over_fp32_mfma_peak': km['assign_useful_flops_over_fp32_mfma_peak'],
                                           'fp32_tier': km.get('assign_fp32_tier')}
    live, pooled = D.blob_stats()
    line['tile_store'] = {'live_blobs': live, 'pooled_bytes': pooled, 'kernel_sources': _hip.source_sha()}
    if tiles8:
      # (last: it replaces the context with one of 8 logical workers)
      del A, B
      line['one_gpu_8_tiles'] = one_gpu_8_tiles_section(line)
      ctx = None
  if world.distributed:
    line['comm'] = dict(world.stats)
    line['comm']['transport'] = world.note or getattr(world.transport, 'name', '?')
    line['rccl'] = rccl_report(world)
    line['launcher'] = ('bench.py self-launch' if os.environ.get('SPARTAN_BENCH_LAUNCHER') == 'self'
                        else 'torch.distributed.run (RANK / WORLD_SIZE from the environment)')
    if line['rccl']['ranks'] != p:
      line['valid_scaling_measurement'] = False      # ranks share devices over the staged debug transport
    del keep[:]
    del A, B
    D.trim_pool()
    for name, section in (('hbm_dist', dist_section), ('lreg_dist', lreg_dist_section), ('kmeans_dist', kmeans_dist_section)):
      if want(name):
        line[name] = guarded(lambda: section(ctx), 300, world.rank, dict(line))
        D.trim_pool()
  world.barrier()
  _emit(line, world.rank)
  sp.shutdown()
  distributed = world.distributed
  world.close()
  if distributed and 'torch.distributed' in sys.modules:      # only a caller that brought torch.distributed itself
    dist = sys.modules['torch.distributed']
    if dist.is_initialized():
      dist.destroy_process_group()


if __name__ == '__main__':
  main()
