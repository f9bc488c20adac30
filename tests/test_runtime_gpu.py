"""The tile store and the collectives of the C-ABI (spartan_amd/csrc/runtime.hip) on the GPU box:
library-owned blobs driven by a host without torch, and the direct RCCL transport of spartan_amd/comm.py on a
one-rank communicator (RCCL refuses two ranks on one device, so N > 1 runs only on the multi-GPU node; every
primitive, the rendezvous, the stream ordering and the start-up self-test are exercised here)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def test_torch_free_host_runs_the_tile_path():
  out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'torch_free_host.py')], cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
  text = out.stdout.decode('utf-8', 'replace')
  assert out.returncode == 0, text[-3000:]
  assert 'torch-free host OK' in text and 'collectives OK' in text, text[-2000:]


def test_rccl_transport_one_rank():
  import torch
  from spartan_amd import comm
  t = comm.RcclTransport(1, 0, comm.RcclTransport.unique_id())
  try:
    ok, msg = t.self_test(60.0)
    assert ok, msg
    dev = torch.device('cuda', 0)
    x = torch.arange(1 << 20, dtype=torch.float32, device=dev)
    # asynchronous transfers: issued on the side stream behind the producer, consumed after wait()
    y = torch.empty_like(x)
    x.mul_(2.0)
    h = t.reduce_scatter(y, x, 'ADD', async_=True)
    z = torch.empty_like(x)
    h2 = t.all_gather_into(z, x, async_=True)
    h.wait()
    h2.wait()
    y.add_(1.0)
    torch.cuda.synchronize()
    want = np.arange(1 << 20, dtype=np.float32) * 2
    np.testing.assert_array_equal(y.cpu().numpy(), want + 1)
    np.testing.assert_array_equal(z.cpu().numpy(), want)
    b = torch.ones(1000, dtype=torch.bool, device=dev)
    t.all_reduce(b, 'ADD')                     # bool travels as bytes
    i = torch.arange(777, dtype=torch.int64, device=dev)
    t.reduce(i, 0, 'MAX')
    t.broadcast(i, 0)
    h3 = t.exchange([(0, i)], [(0, torch.empty_like(i))], async_=True)
    h3.wait()
    torch.cuda.synchronize()
    np.testing.assert_array_equal(i.cpu().numpy(), np.arange(777))
  finally:
    t.close()


def test_world_counts_and_one_rank_world():
  from spartan_amd import comm
  w = comm.World()
  assert not w.distributed and w.transport is None
  assert w.broadcast_object({'a': 1}, 0) == {'a': 1} and w.all_gather_object(3) == [3]
