"""CPU-side checks of the drop-in boundary: libspartan_hip.so loads without a GPU,
exports every entry point include/spartan_hip.h declares, the ctypes structs match
the header's layout, and the product path refuses to run without a device (there is
no CPU fallback).  No kernel is launched here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from spartan_amd import _hip
from spartan_amd.program import Program, dense_strides

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'spartan_hip.h')


EXTRAS_HEADER = os.path.join(ROOT, 'include', 'spartan_hip_extras.h')


def _declared_functions(header=HEADER):
  text = open(header).read()
  text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
  return sorted(set(re.findall(r'\b(sp_[a-z0-9_]+)\s*\(', text)))


def test_library_loads_and_exports_every_declared_symbol():
  names = _declared_functions()
  assert len(names) >= 20, names
  raw = C.CDLL(_hip.LIB_PATH)
  missing = [n for n in names if not hasattr(raw, n)]
  assert not missing, 'declared in include/spartan_hip.h but not exported: %s' % missing
  # and the Python binding declares the same set
  assert sorted(_hip.EXPORTS) == names
  assert _hip.lib().sp_abi_version() == 1
  # the operators outside the tile path live in a library of their own (`make extras`), with a header of their own
  extra = _declared_functions(EXTRAS_HEADER)
  assert sorted(_hip.EXPORTS_EXTRAS) == extra and not set(extra) & set(names)
  xraw = C.CDLL(_hip.EXTRAS_LIB_PATH)
  assert not [n for n in extra if not hasattr(xraw, n)]
  assert not [n for n in extra if hasattr(raw, n)], 'the default library must not carry the extras'


def test_struct_layout_matches_header():
  assert C.sizeof(_hip.sp_instr) == 8
  # cls,n_inputs,n_instr,result_reg,ndim,out_dtype,linear,pad (8 x i32) + shape[4] i64
  # + in_stride[8][4] i64 + in_dtype[8] i32 + consts[16] f64 + iconsts[16] i64 + instr[64]
  want = 8 * 4 + 4 * 8 + 8 * 4 * 8 + 8 * 4 + 16 * 8 + 16 * 8 + 64 * 8
  assert C.sizeof(_hip.sp_program) == want


def test_no_device_is_a_loud_error_not_a_fallback():
  from spartan_amd import comm, kernels
  if comm.gpu_count():
    pytest.skip('GPU present')
  p = Program()
  p.add_input(np.float32, dense_strides((4,)))
  prog = p.finish(_hip.SP_F32, (4,), np.float32, True)
  x = np.zeros(4, np.float32)
  with pytest.raises(Exception):
    kernels.map_fused(prog, [x], np.empty(4, np.float32))            # host memory is refused
  import spartan_amd
  with pytest.raises(Exception):
    spartan_amd.initialize('hip')                                     # no device: no backend, no fallback


def _chain_program(cls, dt):
  shape = (1024, 1024)
  p = Program()
  p.add_input(dt, dense_strides(shape))
  p.add_input(dt, dense_strides(shape))
  p.emit('MUL', 2, 0, 1)
  p.emit('ADD', 2, 2, 0)
  p.emit('NEG', 2, 2)
  p.result_reg = 2
  return p.finish(cls, shape, dt, True)


@pytest.mark.parametrize('cls,dt,t,v', [(_hip.SP_F32, np.float32, 'float', 4), (_hip.SP_F64, np.float64, 'double', 2),
                                        (_hip.SP_I64, np.int64, 'int64_t', 2)])
def test_runtime_specialisation_source_compiles(cls, dt, t, v):
  """The evaluator headers compile under hipRTC with a generated StaticProg (the
  run-time specialised tier, sp_jit.hip) -- cross-compiled here, no device needed."""
  lib = _hip.lib()
  if lib.sp_jit_configure(-1, -1) != 1:
    pytest.skip('libhiprtc not loadable')
  prog = _chain_program(cls, dt)
  exprs = [
      ('map_kernel.hpp', 'sp_map_kernel<%s, %d, 1, true, StaticProg<1000>, -1>' % (t, v)),
      ('map_kernel.hpp', 'sp_map_kernel<%s, %d, 1, false, StaticProg<1000>, 2>' % (t, v)),
      ('map_kernel.hpp', 'sp_map_kernel<%s, %d, 1, false, StaticProg<1000>, -1, true>' % (t, v)),
      ('map_kernel.hpp', 'sp_map_kernel<%s, %d, 1, false, StaticProg<1000>, 1, true>' % (t, v)),
      ('reduce_impl.hpp', 'sp_reduce_rows_kernel<%s, %d, true, PlainAcc, StaticProg<1000>, 0, -1>' % (t, v)),
      ('reduce_impl.hpp', 'sp_reduce_cols_kernel<%s, %d, false, ArgAcc, StaticProg<1000>, -1, -1>' % (t, v)),
  ]
  for header, expr in exprs:
    assert lib.sp_jit_compile_check(header.encode(), expr.encode(), C.byref(prog)) == 1, expr


def test_importing_the_package_does_not_import_torch():
  """torch is not part of the product's single-process path: it appears only as torch.distributed (gloo), the
  control plane of a multi-process job."""
  import subprocess
  import sys
  out = subprocess.run([sys.executable, '-c', "import sys, spartan_amd\nfrom spartan_amd import kernels, backend_hip, sparse, comm\n"
                        "assert 'torch' not in sys.modules\nprint('ok')"], cwd=ROOT, capture_output=True, text=True, timeout=120)
  assert out.returncode == 0 and 'ok' in out.stdout, out.stderr[-2000:]


def test_host_side_plans_need_no_device():
  """The planning entry points are host code: which GEMM schedule a shape gets (sp_gemm_workspace_bytes > 0: split-K
  for few output tiles and a long contraction, the balanced kernel when the 256 x 128 tiles do not fill the 512
  resident workgroups and the cost model predicts a win) and whether a sparse tile gets a column-blocked plan."""
  lib = _hip.lib()
  ws = lambda m, n, k, dt=_hip.SP_F32: int(lib.sp_gemm_workspace_bytes(dt, m, n, k))   # noqa: E731
  balanced = 2 * 256 * 2 * 256 * 128 * 4 + 256          # two 256 x 128 images per resident workgroup
  for shape in ((3072, 3072, 3072), (5000, 5000, 5000), (2304, 2304, 2304), (9216, 9216, 9216), (10000, 10000, 10000)):
    assert ws(*shape) == balanced, shape
  for shape in ((8192, 8192, 8192), (32768, 32768, 32768), (4096, 4096, 4096),     # tile count a multiple of 512
                (2048, 2048, 2048), (3072, 3072, 512), (100, 60, 30)):               # the model says: data-parallel
    assert ws(*shape) == 0, shape
  sk = ws(64, 64, 100000)                                                           # split-K: slices of 64 x 64 partials
  assert sk > 256 and (sk - 256) % (64 * 64 * 4) == 0 and 64 <= (sk - 256) // (64 * 64 * 4) <= 512
  assert ws(3072, 3072, 3072, _hip.SP_F64) == 0
  plan = lambda m, k, nnz, dt=_hip.SP_F32: int(lib.sp_csr_spmv_blockplan_bytes(dt, m, k, nnz))   # noqa: E731
  assert plan(900000, 900000, 9000000) > 10 * 9000000         # 10 bytes per entry + tables
  assert plan(900000, 900000, 9000000, _hip.SP_F64) == 0      # fp32 only
  assert plan(3000, 3000, 300000) == 0 and plan(900000, 900000, 100000) == 0      # too few rows / entries
  assert plan(5000, 5000, 5000 * 100) == 0                    # long rows: the lanes-per-row kernels


def test_rccl_is_found_in_the_documented_order():
  """libspartan_hip.so binds RCCL with dlopen: $SPARTAN_RCCL_LIB first, then librccl.so.1, librccl.so,
  /opt/rocm/lib/librccl.so.1 (README.md).  In fresh interpreters: a path given in the variable is the library that
  gets loaded; a path that does not exist falls through to the default names."""
  import subprocess
  import sys
  system = '/opt/rocm/lib/librccl.so.1'
  if not os.path.exists(system):
    pytest.skip('no RCCL in this image')
  prog = ("import ctypes, os, sys\n"
          "sys.path.insert(0, %r)\n"
          "from spartan_amd import _hip\n"
          "lib = _hip.lib()\n"
          "assert lib.sp_comm_available() == 1, lib.sp_last_error()\n"
          "v = ctypes.c_int(0)\n"
          "assert lib.sp_comm_version(ctypes.byref(v)) == 0 and v.value > 20000\n"
          "print('LOADED', sorted(set(l.split()[-1] for l in open('/proc/self/maps') if 'librccl' in l)))\n") % ROOT
  for value, expect in ((os.path.realpath(system), os.path.realpath(system)), ('/nonexistent/librccl.so', 'librccl')):
    env = dict(os.environ, SPARTAN_RCCL_LIB=value)
    out = subprocess.run([sys.executable, '-c', prog], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    text = out.stdout.decode('utf-8', 'replace')
    assert out.returncode == 0 and 'LOADED' in text, text[-2000:]
    assert expect in text.split('LOADED', 1)[1], text[-500:]


def test_bench_self_launch_reports_a_failed_rank_without_a_gpu():
  """`python bench.py --gpus 2` as a plain process starts its own ranks; on a machine without a GPU every rank fails
  loudly (there is no CPU fallback) and the launcher must say so: ONE JSON line of the contract's shape with the reason,
  and a non-zero exit code."""
  import json
  import subprocess
  import sys
  try:
    import torch
    if torch.cuda.is_available():
      pytest.skip('a GPU is visible: the ranks would run')
  except ImportError:
    pass
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  p = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0',
                      '--deadline', '240'], capture_output=True, text=True, timeout=300, cwd=root)
  assert p.returncode != 0
  lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
  assert len(lines) == 1
  line = json.loads(lines[0])
  assert line['n_gpus'] == 2 and line['value'] is None and 'FAILED' in line['error']
  assert 'self-launch' in line['launcher']
