"""One rank of the two-rank map2(update_region=...) test (launched by test_region_join.py): 4 logical workers over
2 processes, so every grid cell is fetched from, and written to, tiles of both ranks."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import spartan_amd as sp  # noqa: E402
from oracle.np_backend import NumpyBackend  # noqa: E402


def main():
  workers = int(sys.argv[1])
  use_hip = len(sys.argv) > 2 and sys.argv[2] == 'hip'
  world = sp.World.from_env(backend=os.environ.get('SPARTAN_TEST_BACKEND', 'socket'))
  assert world.size == 2
  if use_hip:
    world.staged = True
    sp.initialize('hip', num_workers=workers, world=world)
  else:
    sp.initialize(backend=NumpyBackend(), num_workers=workers, world=world)
  from tests import test_region_join
  before = dict(world.stats)
  n = test_region_join.check_all()
  assert world.stats['p2p_bytes'] > before['p2p_bytes'], 'no cell crossed the ranks?'
  world.barrier()
  print('RANK %d OK %d' % (world.rank, n))
  sys.stdout.flush()


if __name__ == '__main__':
  main()
