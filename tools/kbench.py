"""Kernel micro-benchmarks (HIP-event timed on the launch stream).  Prints one
line per kernel/config; used to pick defaults and to fill profiles/."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spartan_amd import _hip, kernels  # noqa: E402
from spartan_amd.program import Program, dense_strides  # noqa: E402

DEV = 'cuda:0'


def timeit(fn, iters=10, warmup=3):
  for _ in range(warmup):
    fn()
  _hip.lib().sp_jit_wait()   # run-time specialised kernels requested by the warm-up are ready
  fn()
  torch.cuda.synchronize()
  e0, e1 = kernels.Event(), kernels.Event()
  e0.record()
  for _ in range(iters):
    fn()
  e1.record()
  e1.synchronize()
  return e0.elapsed_ms(e1) / iters


def bench_copy(nbytes):
  a = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
  b = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
  ms = timeit(lambda: kernels.stream_copy(b, a))
  print('stream_copy  %6.2f GiB  %8.3f ms  %8.1f GB/s (read+write)' % (nbytes / 2**30, ms, 2 * nbytes / ms / 1e6))


def map_prog(n, nops, shape=None):
  p = Program()
  shape = shape or (n,)
  p.add_input(np.float32, dense_strides(shape))
  if nops == 1:
    c = p.add_const(1.0)
    p.emit('CONST', 1, c)
    p.emit('ADD', 1, 0, 1)
  else:
    p.emit('MUL', 1, 0, 0)
    p.emit('ADD', 1, 1, 0)
    for _ in range(nops - 2):
      p.emit('ADD', 1, 1, 0)
  p.result_reg = 1
  return p.finish(_hip.SP_F32, shape, np.float32, True)


def bench_map(n):
  x = torch.rand(n, dtype=torch.float32, device=DEV)
  out = torch.empty_like(x)
  for nops, name in ((1, 'x+1'), (2, 'x*x+x'), (6, '6-op chain'), (12, '12-op chain')):
    prog = map_prog(n, nops)
    ms = timeit(lambda: kernels.map_fused(prog, [x], out))
    print('map %-12s n=%.2e  %8.3f ms  %8.1f GB/s (8 B/elem)' % (name, n, ms, 8.0 * n / ms / 1e6))


def bench_map_bcast(rows, cols):
  from spartan_amd.program import broadcast_strides, collapse
  x = torch.rand(rows, cols, dtype=torch.float32, device=DEV)
  out = torch.empty_like(x)
  for name, bshape in (('x - col(N,1)', (rows, 1)), ('x * row(D,)', (cols,))):
    b = torch.rand(*bshape, dtype=torch.float32, device=DEV)
    p = Program()
    cshape, cst = collapse((rows, cols), [broadcast_strides((rows, cols), (rows, cols)), broadcast_strides(bshape, (rows, cols))])
    for st in cst:
      p.add_input(np.float32, st)
    p.emit('SUB' if 'col' in name else 'MUL', 2, 0, 1)
    p.result_reg = 2
    prog = p.finish(_hip.SP_F32, cshape, np.float32, False)
    ms = timeit(lambda: kernels.map_fused(prog, [x, b], out))
    print('map %-14s %dx%d  %8.3f ms  %8.1f GB/s (8 B/elem)' % (name, rows, cols, ms, 8.0 * rows * cols / ms / 1e6))


def bench_reduce(rows, cols):
  x = torch.rand(rows, cols, dtype=torch.float32, device=DEV)
  n = rows * cols
  for axis in (None, 0, 1):
    if axis is None:
      O, A, I = 1, n, 1
    elif axis == 0:
      O, A, I = 1, rows, cols
    else:
      O, A, I = rows, cols, 1
    p = Program()
    p.add_input(np.float32, dense_strides((O, A, I)))
    prog = p.finish(_hip.SP_F32, (O, A, I), None, True)
    out = torch.empty(O * I, dtype=torch.float32, device=DEV)
    ms = timeit(lambda: kernels.reduce(prog, [x], 'SUM', O, A, I, out))
    print('sum axis=%-4s %dx%d  %8.3f ms  %8.1f GB/s (4 B/elem)' % (axis, rows, cols, ms, 4.0 * n / ms / 1e6))
    oi = torch.empty(O * I, dtype=torch.int64, device=DEV)
    ms = timeit(lambda: kernels.argreduce(prog, [x], 0, O, A, I, 0, n, oi, out))
    print('argmax axis=%-4s %dx%d  %8.3f ms  %8.1f GB/s (4 B/elem)' % (axis, rows, cols, ms, 4.0 * n / ms / 1e6))


def bench_gemm(M, N, K, variants=(0, 1, 2, 3)):
  a = torch.rand(M, K, dtype=torch.float32, device=DEV) * 2 - 1
  b = torch.rand(K, N, dtype=torch.float32, device=DEV) * 2 - 1
  c = torch.empty(M, N, dtype=torch.float32, device=DEV)
  for v in variants:
    os.environ['SP_GEMM_VARIANT'] = str(v)
    # the variant is latched on first use inside the library: one process per variant
    ms = timeit(lambda: kernels.gemm_f32(a, b, c), iters=5, warmup=2)
    print('gemm v%d %dx%dx%d  %8.3f ms  %7.1f TFLOP/s  (%.1f%% of 157.3)' % (
        v, M, N, K, ms, 2.0 * M * N * K / ms / 1e9, 2.0 * M * N * K / ms / 1e9 / 157.3 * 100))
    break


def bench_kmeans(n, k, d):
  """BASELINE configs[3] per-GPU tile: 1 250 000 x 256 fp32 points, k = 1024."""
  g = torch.Generator(device=DEV)
  g.manual_seed(20150708)
  x = torch.rand(n, d, dtype=torch.float32, device=DEV, generator=g)
  c = torch.rand(k, d, dtype=torch.float64, device=DEV, generator=g)
  labels = torch.empty(n, dtype=torch.int64, device=DEV)
  flop = 2.0 * n * k * d
  ms = timeit(lambda: kernels.nearest_center(x, c, labels, _hip.NEAREST_FUSED_UNCHECKED), iters=5, warmup=2)
  amb = int((labels < 0).sum().item())
  print('nearest fused kernel only  %dx%dx%d  %8.3f ms  %7.1f TFLOP/s (%.1f%% of 157.3)  undecided rows %d (%.3f%%)'
        % (n, k, d, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3 * 100, amb, 100.0 * amb / n))
  ms = timeit(lambda: kernels.nearest_center(x, c, labels, _hip.NEAREST_FUSED), iters=5, warmup=2)
  print('nearest fused + re-check   %dx%dx%d  %8.3f ms  %7.1f TFLOP/s (%.1f%% of 157.3)'
        % (n, k, d, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3 * 100))
  ne = min(n, 20000)
  le = torch.empty(ne, dtype=torch.int64, device=DEV)
  ms_e = timeit(lambda: kernels.nearest_center(x[:ne], c, le, _hip.NEAREST_EXACT), iters=2, warmup=1)
  print('nearest exact tier         %dx%dx%d  %8.3f ms  %7.2f TFLOP/s (fp64, 1 wave per point)'
        % (ne, k, d, ms_e, 2.0 * ne * k * d / ms_e / 1e9))
  assert bool((le == labels[:ne]).all().item()), 'fused tier disagrees with the exact tier'
  counts = torch.empty(k, dtype=torch.int64, device=DEV)
  ms = timeit(lambda: kernels.bincount(labels, k, counts))
  print('bincount                   n=%d k=%d  %8.3f ms  %8.1f GB/s (8 B/label)' % (n, k, ms, 8.0 * n / ms / 1e6))
  out = torch.empty(k, d, dtype=torch.float32, device=DEV)
  ms = timeit(lambda: kernels.segment_sum(x, labels, k, out))
  print('segment_sum                %dx%d k=%d  %8.3f ms  %8.1f GB/s (4*n*d bytes)' % (n, d, k, ms, 4.0 * n * d / ms / 1e6))
  bc = torch.bincount(labels, minlength=k)
  print('   cluster sizes: mean %.0f  max %d  min %d' % (n / k, int(bc.max()), int(bc.min())))
  ul = torch.randint(0, k, (n,), dtype=torch.int64, device=DEV, generator=g)
  ms = timeit(lambda: kernels.segment_sum(x, ul, k, out))
  print('segment_sum (uniform random labels)  %8.3f ms  %8.1f GB/s' % (ms, 4.0 * n * d / ms / 1e6))


def prewarm():
  """~150 ms of streaming load before anything is timed: these boxes stall dispatch once (~30 ms) about
  50 ms into the first sustained load of a process and ramp clocks afterwards (profiles/r01_notes.md)."""
  a = torch.empty(1 << 30, dtype=torch.uint8, device=DEV)
  b = torch.empty(1 << 30, dtype=torch.uint8, device=DEV)
  for _ in range(400):
    kernels.stream_copy(b, a)
  torch.cuda.synchronize()


if __name__ == '__main__':
  what = sys.argv[1] if len(sys.argv) > 1 else 'all'
  prewarm()
  if what in ('all', 'copy'):
    bench_copy(2 << 30)
  if what in ('all', 'map'):
    print('SP_NO_STATIC=%s SP_MAP_UNROLL=%s' % (os.environ.get('SP_NO_STATIC'), os.environ.get('SP_MAP_UNROLL')))
    bench_map(1 << 29)
    bench_map_bcast(125000, 4096)
  if what in ('all', 'reduce'):
    bench_reduce(8192, 65536)
    bench_reduce(125000, 4096)
  if what == 'splitk':
    for dt, peak in ((torch.float32, 157.3), (torch.float64, 78.6)):
      for (M, N, K) in ((512, 512, 1000000), (64, 64, 4000000), (4096, 4096, 125000)):
        a = torch.rand(M, K, dtype=dt, device=DEV) * 2 - 1
        b = torch.rand(K, N, dtype=dt, device=DEV) * 2 - 1
        c = torch.empty(M, N, dtype=dt, device=DEV)
        ms = timeit(lambda: kernels.gemm_f32(a, b, c), iters=5, warmup=2)
        nbytes = (M * K + K * N) * a.element_size()
        print('gemm %s %dx%dx%d  %8.3f ms  %7.1f TFLOP/s (%.0f%% of MFMA peak)  operands streamed at %6.1f GB/s  ws=%d'
              % (str(dt).split('.')[1], M, N, K, ms, 2.0 * M * N * K / ms / 1e9, 2.0 * M * N * K / ms / 1e9 / peak * 100,
                 nbytes / ms / 1e6, _hip.lib().sp_gemm_workspace_bytes(_hip.SP_F32 if dt == torch.float32 else _hip.SP_F64, M, N, K)))
        del a, b, c
  if what == 'dgemm':
    M, N, K = [int(v) for v in sys.argv[2:5]]
    a = torch.rand(M, K, dtype=torch.float64, device=DEV) * 2 - 1
    b = torch.rand(K, N, dtype=torch.float64, device=DEV) * 2 - 1
    c = torch.empty(M, N, dtype=torch.float64, device=DEV)
    ms = timeit(lambda: kernels.gemm_f32(a, b, c), iters=5, warmup=2)
    print('dgemm %dx%dx%d  %8.3f ms  %7.1f TFLOP/s  (%.1f%% of 78.6 fp64 MFMA peak)' % (
        M, N, K, ms, 2.0 * M * N * K / ms / 1e9, 2.0 * M * N * K / ms / 1e9 / 78.6 * 100))
  if what == 'kmeans':
    if len(sys.argv) > 4:
      bench_kmeans(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))
    else:
      bench_kmeans(1250000, 1024, 256)
  if what == 'gemm':
    M, N, K = [int(v) for v in sys.argv[2:5]]
    bench_gemm(M, N, K)
