"""The N>1 path on CPU: two processes, gloo backend, NumPy tile backend.
Covers the SPMD tile walk, grouped point-to-point fetch/update and the
collective fast paths (reduce / reduce-scatter / all-gather) of
spartan_amd/array/distarray.py."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  port = s.getsockname()[1]
  s.close()
  return port


def _run_two_ranks(workers, extra=()):
  _run_ranks(2, 'mp_worker.py', [str(workers)] + list(extra))


def _run_ranks(size, script, args=()):
  port = _free_port()
  procs = []
  for rank in range(size):
    env = dict(os.environ)
    env.update({'RANK': str(rank), 'WORLD_SIZE': str(size), 'LOCAL_RANK': str(rank),
                'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port),
                'OMP_NUM_THREADS': '1', 'GLOO_SOCKET_IFNAME': 'lo'})
    procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', script)] + list(args),
                                  env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, cwd=ROOT))
  outs = []
  for p in procs:
    try:
      out, _ = p.communicate(timeout=300)
    except subprocess.TimeoutExpired:
      for q in procs:
        q.kill()
      raise
    outs.append(out.decode('utf-8', 'replace'))
  for rank, (p, out) in enumerate(zip(procs, outs)):
    assert p.returncode == 0, 'rank %d failed:\n%s' % (rank, out[-4000:])
    assert 'RANK %d OK' % rank in out, out[-2000:]


@pytest.mark.parametrize('workers', [2, 4])
def test_two_ranks_gloo(workers):
  _run_two_ranks(workers)


@pytest.mark.parametrize('size', [3, 4, 8])
def test_ksplit_pipeline_more_ranks(size):
  """The K-split dot pipeline, the reductions' collectives and a k-means iteration at 3, 4 and 8 ranks."""
  _run_ranks(size, 'mp_ksplit_worker.py')


@pytest.mark.gpu
def test_two_ranks_hip_backend_shared_gpu():
  """The N>1 path with the HIP kernels: two ranks share GPU 0 and exchange HBM
  blobs through the staged debug transport (RCCL needs one GPU per rank, which
  the driver's 8-GPU run provides)."""
  _run_two_ranks(2, extra=('hip',))
