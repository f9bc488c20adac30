"""Differential fuzz against the REFERENCE ITSELF: 300 random expression DAGs (the generator of tests/test_fuzz_gpu.py:
element-wise trees with NumPy broadcasting, scalar operands and dtype mixes, slices and transposes, reductions and
arg-reductions over every axis, fused or not) were built over the reference's builders and run by it at 1 / 3 / 4 / 8
workers (tests/golden/make_golden.py --fuzz -> fuzz_w*.npz, fuzz_meta.json); the ~210 it can run are fixtures.  Here
the same seeds are built over the product's builders and must give the reference's shapes, dtypes and values:
bit-exact on the NumPy tile backend (the same NumPy operations tile by tile, merged in the same order), bit-exact for
integer / boolean / index results and within 2e-5 relative for floating point on the HIP backend (its in-tile
reductions add in another order).  tests/test_fuzz_gpu.py compares the two BACKENDS under one host framework; this file
is what tells a mistake in the host framework itself."""
import json
import os

import numpy as np
import pytest

import spartan_amd as sp
from tests import test_fuzz_gpu as fz

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
META = json.load(open(os.path.join(HERE, 'fuzz_meta.json')))
KEEP = 1024
# A seed is a fixture if the reference runs it at EVERY worker count.  A dozen run with one worker only: mixed-dtype
# programs whose updates fail the reference's dtype assertion as soon as a target has a second tile -- and whose
# one-worker answers show the same confusion (a fused arg-reduction keeps its extreme values in the dtype of the first
# fused input; where that loses the value no element equals it and the sentinel comes back for every position).
DOTS = sorted(k for k in META['1'] if k.startswith('d'))
EVERYWHERE = set.intersection(*[{seed for seed, m in META[w].items() if 'skipped' not in m and not seed.startswith('d')} for w in META])


def _sample(val):
  flat = np.ascontiguousarray(val).ravel()
  if flat.size > KEEP:
    flat = flat[::-(-flat.size // KEEP)]
  return flat


def _run(workers, exact_floats):
  gold = np.load(os.path.join(HERE, 'fuzz_w%d.npz' % workers))
  meta = META[str(workers)]
  bad, ran, known, refused, truncated = [], 0, [0], [0], [0]
  for seed, m in sorted(((k, v) for k, v in meta.items() if not k.startswith('d')), key=lambda kv: int(kv[0])):
    if 'skipped' in m or seed not in EVERYWHERE:
      continue
    info = {}
    try:
      fz.SEEN_F32[0] = False
      with np.errstate(all='ignore'):
        got = np.asarray(fz._program(int(seed), sp, info))
    except Exception as e:   # noqa: BLE001
      # (HIP leg, as in tests/test_fuzz_gpu.py: arithmetic on two boolean operands makes NumPy produce int8 / float16
      #  tiles, which are outside the kernels' set -- the refusal is loud and such a program is not compared)
      if not exact_floats and ('unsupported dtype' in str(e) or 'is not supported by the HIP tile backend' in str(e)) and \
          any(t in str(e).split('(supported')[0] for t in ('float16', 'int8', 'int16')):
        refused[0] += 1
        continue
      bad.append((seed, 'raised %s: %s' % (type(e).__name__, str(e)[:120])))
      continue
    want = gold['s' + seed]
    ran += 1
    if list(got.shape) != m['shape']:
      bad.append((seed, 'shape', got.shape, m['shape']))
      continue
    g = _sample(got)
    if got.dtype.str != m['dtype']:
      # The two places where the product keeps NumPy's result type and the py3-run reference does not:
      #  * a FUSED arg-reduction: the reference's fused ReduceExpr takes its output dtype from the first input of the
      #    fused operator (reduce.py:102 `dtype_fn(children[0])`), so its indices come back as float32 / float64 --
      #    or as bool, every index but 0 collapsed to True; the product returns int64 indices: same positions;
      #  * mean(axis=None): sum / size with a Python int, float64 under NEP 50 there, the operand's float32 here.
      want_dt = np.dtype(m['dtype'])
      if info['tail'] == 'argmax' and info['optimized'] and got.dtype == np.int64 and want_dt.kind in 'fb':
        same = np.array_equal(g != 0, want) if want_dt.kind == 'b' else np.array_equal(g.astype(want_dt), want)
        if not same:
          bad.append((seed, 'fused argmax positions'))
        known[0] += 1
      elif info['tail'] == 'mean' and info['axis'] is None and got.dtype == np.float32 and want_dt == np.float64:
        if not np.allclose(g.astype(np.float64), want, rtol=2e-6, atol=0, equal_nan=True):
          bad.append((seed, 'mean(None) value'))
        known[0] += 1
      else:
        bad.append((seed, 'dtype', got.dtype.str, m['dtype'], info))
      continue
    if exact_floats and m['sum'] is not None and got.size > KEEP:
      # (beyond the sample: the float64 sum of ALL values, which identical values reproduce exactly)
      with np.errstate(all='ignore'):
        total = float(np.nansum(got.astype(np.float64)))
      if total != m['sum']:
        bad.append((seed, 'sum of all values', total, m['sum']))
        continue
    if got.dtype.kind in 'iub' or exact_floats:
      if not np.array_equal(g, want, equal_nan=got.dtype.kind == 'f'):
        diff = np.abs(g.astype(np.float64) - want.astype(np.float64))
        # (HIP leg: a float reduction whose result the reference's dtype rule stores as an integer truncates two
        #  summation orders of 10.999999999999998 to different integers -- off by one, in a few cells: counted)
        if not exact_floats and got.dtype.kind == 'i' and diff.max() == 1 and (diff > 0).mean() < 0.1:
          truncated[0] += 1
        else:
          bad.append((seed, 'values', float(diff.max())))
    else:
      f32 = got.dtype == np.float32 or fz.SEEN_F32[0]
      if not np.allclose(g, want, rtol=2e-5 if f32 else 1e-11, atol=5e-6 if f32 else 1e-6, equal_nan=True):
        bad.append((seed, 'values', float(np.nanmax(np.abs(g.astype(np.float64) - want.astype(np.float64))))))
  assert known[0] <= 0.1 * ran, 'too many results with the two known dtype differences: %d of %d' % (known[0], ran)
  assert refused[0] <= 0.1 * len(EVERYWHERE) and truncated[0] <= 2, (refused[0], truncated[0])
  return ran, bad


@pytest.mark.parametrize('workers', [1, 3, 4, 8])
def test_the_host_framework_computes_what_the_reference_recorded(workers):
  from oracle.np_backend import NumpyBackend
  sp.initialize(backend=NumpyBackend(), num_workers=workers)
  try:
    ran, bad = _run(workers, exact_floats=True)
  finally:
    sp.shutdown()
  assert ran >= 190 and not bad, bad[:10]


@pytest.mark.gpu
@pytest.mark.parametrize('workers', [1, 3, 4, 8])
def test_the_hip_backend_computes_what_the_reference_recorded(workers):
  sp.initialize('hip', num_workers=workers)
  try:
    ran, bad = _run(workers, exact_floats=False)
  finally:
    sp.shutdown()
  assert ran >= 175 and not bad, bad[:10]


def _run_dots(workers):
  """The 100 random dots (every dispatch branch of dot.py:243-299; integer-valued operands): exact on any backend."""
  gold = np.load(os.path.join(HERE, 'fuzz_w%d.npz' % workers))
  meta = META[str(workers)]
  bad = []
  for key in DOTS:
    m = meta[key]
    assert 'skipped' not in m, key
    got = np.asarray(fz._dot_case(int(key[1:]), sp))
    if list(got.shape) != m['shape'] or got.dtype.str != m['dtype'] or not np.array_equal(_sample(got), gold[key]) \
            or float(got.astype(np.float64).sum()) != m['sum']:      # (the sample, and the sum of ALL of it: exact)
      bad.append((key, got.shape, got.dtype.str, m['shape'], m['dtype']))
  return bad


@pytest.mark.parametrize('workers', [1, 3, 4, 8])
def test_random_dots_are_what_the_reference_recorded_cpu(workers):
  from oracle.np_backend import NumpyBackend
  sp.initialize(backend=NumpyBackend(), num_workers=workers)
  try:
    assert _run_dots(workers) == []
  finally:
    sp.shutdown()


@pytest.mark.gpu
@pytest.mark.parametrize('workers', [1, 3, 4, 8])
def test_random_dots_are_what_the_reference_recorded_hip(workers):
  sp.initialize('hip', num_workers=workers)
  try:
    assert _run_dots(workers) == []
  finally:
    sp.shutdown()
