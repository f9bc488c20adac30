"""checkpoint: evaluate an expression and keep a copy of its tiles on disk (reference
spartan/expr/operator/checkpoint.py).  Mode 'disk' saves through fio.save under `<path>/<expr_id>/`; `load_data`
reads it back whole, or only the tiles listed as bad (`Context.mark_failed_worker`, the GPU-reset model of a lost
worker).  The reference's 'replica' mode and the heartbeat that DETECTS a dead worker (master.py:142-146) are control
plane and not built: a dead rank is detected by torch.distributed's watchdog and restarted by the launcher."""
import tempfile

from . import base
from .base import Expr, lazify
from .fio import load, partial_load, save

CHECKPOINT_PATH = tempfile.gettempdir() + '/spartan_amd_checkpoint'    # FLAGS.checkpoint_path


class CheckpointExpr(Expr):
  """checkpoint.py:11-48."""
  members = ('src', 'path', 'mode', 'ready')

  def dependencies(self):
    return {'src': self.src}

  def visit(self, visitor):
    return base.expr_like(self, src=visitor.visit(self.src), path=self.path, mode=self.mode, ready=self.ready)

  def pretty_str(self):
    return 'checkpoint(expr_id=%s, path=%s)' % (self.expr_id, self.path)

  def compute_shape(self):
    return self.src.shape

  def load_data(self, cached_result, workers_for_reload=None):
    """checkpoint.py:20-41: the array back from disk -- whole, or the tiles in cached_result.bad_tiles
    (`workers_for_reload`: {extent: worker}, what the reference's master computes)."""
    if not self.ready or self.mode != 'disk':
      return None
    if cached_result is not None:
      from .. import context
      extents = workers_for_reload or context.get().get_workers_for_reload(cached_result)
      for ex, tile_id in partial_load(extents, "%s" % self.expr_id, path=self.path, iszip=False).items():
        cached_result.tiles[ex] = tile_id
        cached_result.blob_to_ex[tile_id] = ex
        if ex in cached_result.bad_tiles:
          cached_result.bad_tiles.remove(ex)
      return cached_result
    return load("%s" % self.expr_id, path=self.path, iszip=False).evaluate()

  def _evaluate(self, ctx, deps):
    value = deps['src']
    if self.mode == 'disk':            # one file per tile, written by the worker that holds it (fio.save)
      save(value, str(self.expr_id), path=self.path, iszip=False)
    self.ready = True
    return value


def checkpoint(x, mode='disk'):
  """checkpoint.py:51-59."""
  return CheckpointExpr(src=lazify(x), path=CHECKPOINT_PATH, mode=mode, ready=False)
