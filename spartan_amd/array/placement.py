"""Which worker gets which tile of a new array.

The reference picks among five policies with FLAGS.tile_assignment_strategy inside `distarray.create`
(spartan/array/distarray.py:441-476).  Here a policy is a function
    place(tiles, ctx) -> [worker for each tile]
over the tiles in creation order (`tiles` = [(extent, shard index)] as compute_extents yields them; policies count
tiles by their POSITION in that order -- the shard index is already reduced mod W, which makes the reference's
serpentine test `(i / num_workers) % 2` always false, distarray.py:455); every rank runs it on the same metadata
and must get the same answer, so nothing here may depend on rank-local state:

  round_robin  tile i -> worker i mod W (the default; it is what puts tile r on GPU r)
  serpentine   rows of W tiles alternate direction (0..W-1, W-1..0, ...): neighbouring tiles of consecutive
               tile rows share a worker
  performance  workers ranked by what they already hold, least loaded first (the reference ranks by the
               master's worker scores, master.py get_worker_scores; the bytes of live tiles per worker are known
               to every rank from array metadata, no exchange needed)
  static       worker ids read from a file, one per line, for the tiles in sorted order
               ($SPARTAN_TILES_MAP; the reference reads <user config dir>/spartan/tiles_map)
  random       a worker drawn per tile from a generator seeded identically on every rank

Select with `spartan_amd.array.placement.STRATEGY = name` or $SPARTAN_TILE_ASSIGNMENT.
"""
import os

import numpy as np

STRATEGY = os.environ.get('SPARTAN_TILE_ASSIGNMENT', 'round_robin')
_rng = np.random.RandomState(20150708)


def seed(value):
  """Re-seed the 'random' policy (call with the same value on every rank)."""
  _rng.seed(int(value) % 4294967295)


def _round_robin(tiles, ctx):
  return [i % ctx.num_workers for i in range(len(tiles))]


def _serpentine(tiles, ctx):
  w = ctx.num_workers
  return [(w - 1 - i % w) if (i // w) % 2 else i % w for i in range(len(tiles))]


def _performance(tiles, ctx):
  ranked = [worker for worker, _ in ctx.worker_scores()]
  return [ranked[i % len(ranked)] for i in range(len(tiles))]


def _static(tiles, ctx):
  path = os.environ.get('SPARTAN_TILES_MAP')
  if not path:
    raise ValueError("tile assignment 'static' needs $SPARTAN_TILES_MAP (one worker id per line)")
  with open(path) as f:
    ids = [int(line) for line in f.read().split()]
  order = sorted(range(len(tiles)), key=lambda j: tiles[j][0])
  if len(ids) < len(tiles):
    raise ValueError('%s lists %d workers for %d tiles' % (path, len(ids), len(tiles)))
  out = [0] * len(tiles)
  for rank_in_file, j in enumerate(order):
    out[j] = ids[rank_in_file] % ctx.num_workers
  return out


def _random(tiles, ctx):
  return [int(v) for v in _rng.randint(0, ctx.num_workers, size=len(tiles))]


POLICIES = {'round_robin': _round_robin, 'serpentine': _serpentine, 'performance': _performance,
            'static': _static, 'random': _random}


def place(tiles, ctx):
  try:
    policy = POLICIES[STRATEGY]
  except KeyError:
    raise ValueError('unknown tile assignment strategy %r (known: %s)' % (STRATEGY, ', '.join(sorted(POLICIES))))
  return policy(tiles, ctx)
