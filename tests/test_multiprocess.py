"""The N>1 path on CPU: several processes, NumPy tile backend, over both host transports -- the rendezvous hub
('socket': standard library only, the control plane of every job) and a torch.distributed gloo group ('gloo').
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


def _run_two_ranks(workers, extra=(), backend='socket', extra_env=None):
  _run_ranks(2, 'mp_worker.py', [str(workers)] + list(extra), backend=backend, extra_env=extra_env)


def _run_ranks(size, script, args=(), backend='socket', extra_env=None):
  port = _free_port()
  procs = []
  for rank in range(size):
    env = dict(os.environ)
    env.update({'RANK': str(rank), 'WORLD_SIZE': str(size), 'LOCAL_RANK': str(rank),
                'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port), 'SPARTAN_TEST_BACKEND': backend,
                'OMP_NUM_THREADS': '1', 'GLOO_SOCKET_IFNAME': 'lo'})
    env.update(extra_env or {})
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
def test_two_ranks_socket(workers):
  _run_two_ranks(workers, backend='socket')


def test_two_ranks_gloo():
  _run_two_ranks(2, backend='gloo')


def test_two_ranks_bring_their_own_torch_group():
  """`World.from_env()` with no backend argument in a process whose torch.distributed group is already initialised
  (CPU box): the default data plane is gloo over that group (it used to pick 'socket' and refuse)."""
  # (the case is a box WITHOUT GPUs; on a GPU box the two ranks would share one device, which RCCL refuses: hide it)
  _run_two_ranks(2, backend='own-torch', extra_env={'HIP_VISIBLE_DEVICES': '', 'ROCR_VISIBLE_DEVICES': '', 'CUDA_VISIBLE_DEVICES': ''})


@pytest.mark.parametrize('size,backend', [(3, 'socket'), (4, 'socket'), (8, 'socket'), (4, 'gloo')])
def test_ksplit_pipeline_more_ranks(size, backend):
  """The K-split dot pipeline, the reductions' collectives and a k-means iteration at 3, 4 and 8 ranks."""
  _run_ranks(size, 'mp_ksplit_worker.py', backend=backend)


@pytest.mark.gpu
def test_two_ranks_hip_backend_shared_gpu():
  """The N>1 path with the HIP kernels: two ranks share GPU 0 and exchange HBM
  blobs through the staged debug transport (RCCL needs one GPU per rank, which
  the driver's 8-GPU run provides)."""
  _run_two_ranks(2, extra=('hip',))


@pytest.mark.gpu
@pytest.mark.parametrize('size', [4, 8])
def test_ksplit_pipeline_hip_backend_many_ranks_one_gpu(size):
  """The 4- and 8-rank K-split pipeline, the reductions' collectives and a k-means iteration with the HIP kernels:
  every rank is a process of its own on GPU 0, blobs cross ranks through the staged transport; plan counters and
  bit-exact integer results are asserted inside the worker (tests/mp_ksplit_worker.py).  The reference's runner
  sweeps worker counts the same way (tests/test_common.py:98-120)."""
  _run_ranks(size, 'mp_ksplit_worker.py', ['hip'])


@pytest.mark.gpu
def test_bench_eight_ranks_on_one_gpu_line():
  """`python bench.py --gpus 8` as the driver would read it, on whatever GPUs are there: one JSON line with the
  distributed sections; with fewer than 8 GPUs the ranks share devices and the line says it is no scaling
  measurement."""
  import json
  env = dict(os.environ)
  for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
    env.pop(k, None)
  p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--size', '4096', '--steps', '2',
                      '--warmup', '1', '--deadline', '900'], env=env, cwd=ROOT, stdout=subprocess.PIPE,
                     stderr=subprocess.PIPE, timeout=1000)
  out = p.stdout.decode('utf-8', 'replace').strip()
  assert p.returncode == 0, (out[-2000:], p.stderr.decode('utf-8', 'replace')[-4000:])
  assert len(out.splitlines()) == 1, out[-2000:]
  line = json.loads(out)
  assert line['n_gpus'] == 8 and line['value'] > 0 and line['scaling'] == 'strong'
  for key in ('dot_breakdown', 'hbm_dist', 'kmeans_dist', 'lreg_dist', 'rccl', 'comm'):
    assert key in line, sorted(line)
  rccl = line['rccl']
  assert rccl['control_plane'] == 'socket' and rccl['torch_in_process'] is False, rccl
  assert all(n <= 1 for n in rccl['runtimes_mapped'].values()), rccl['runtimes_mapped']   # one HIP / HSA runtime (RCCL: 0 or 1)
  assert 'collectives' in line and line['collectives']['values_ok'] is True, line.get('collectives')
  assert len(out) < 7900
  if rccl['visible_gpus'] < 8:
    assert rccl['ranks'] == 0 and line['valid_scaling_measurement'] is False, line
  else:
    assert rccl['ranks'] == 8 and rccl['hip_runtime_path'] == rccl['own_hip_runtime_path'], rccl


def test_ranks_started_by_torch_distributed_run():
  """The driver's launcher: `python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1
  --master-port P script` -- its agent's TCPStore holds MASTER_PORT, the ranks' own rendezvous hub sits one port up
  (rendezvous.endpoint), and the rank processes themselves never import torch (asserted inside the worker)."""
  port = _free_port()
  env = dict(os.environ, SPARTAN_TEST_BACKEND='socket', OMP_NUM_THREADS='1')
  for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'SPARTAN_RDZV_PORT'):
    env.pop(k, None)
  p = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                      '--master-addr', '127.0.0.1', '--master-port', str(port),
                      os.path.join(ROOT, 'tests', 'mp_ksplit_worker.py')],
                     env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
  out = p.stdout.decode('utf-8', 'replace')
  assert p.returncode == 0, out[-4000:]
  assert 'RANK 0 OK' in out and 'RANK 1 OK' in out, out[-2000:]
