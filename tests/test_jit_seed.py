"""`spartan_amd.jit_seed.seed()` -- the step of `__graft_entry__.build()` that pre-compiles the specialised kernels of
the known workloads -- on both kinds of machine: without a GPU (tiles are host stand-ins, the launches fail and are
ignored) and WITH one (the launches really run, so the tiles must be device tiles: a host pointer would be a GPU memory
fault that aborts the process, which is what build() did on a GPU box before this test existed)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROG = ("import sys\n"
        "from spartan_amd import jit_seed, devarray\n"
        "n = jit_seed.seed(sys.argv[1])\n"
        "assert devarray._storage_cls[0] is devarray.Storage      # seed mode is left again\n"
        "print('WRITTEN', n)\n")


def _seed_into(directory):
  p = subprocess.run([sys.executable, '-c', PROG, str(directory)], cwd=ROOT, capture_output=True, text=True, timeout=900)
  assert p.returncode == 0, (p.returncode, p.stderr[-2000:])
  written = int(p.stdout.strip().splitlines()[-1].split()[1])
  files = [f for f in os.listdir(str(directory)) if f.endswith('.spco')]
  assert written == len(files) and written >= 60, (written, len(files))
  return sorted(files)


def test_seeding_without_a_gpu_writes_the_code_objects(tmp_path):
  import torch
  if torch.cuda.is_available():
    pytest.skip('this is the no-GPU half; test_seeding_on_a_gpu_box covers the other')
  files = _seed_into(tmp_path)
  # the tree's own seeds (written by build()) are the same set: same programs, same source hash in the names
  tree = os.path.join(ROOT, 'spartan_amd', 'csrc', 'jit_seed')
  if os.path.isdir(tree) and os.listdir(tree):
    assert files == sorted(f for f in os.listdir(tree) if f.endswith('.spco'))


@pytest.mark.gpu
def test_seeding_on_a_gpu_box(tmp_path):
  files = _seed_into(tmp_path)
  tree = os.path.join(ROOT, 'spartan_amd', 'csrc', 'jit_seed')
  assert files == sorted(f for f in os.listdir(tree) if f.endswith('.spco'))
