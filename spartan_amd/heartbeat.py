"""Failure detection: who is still there.

The reference's workers report to the master every `heartbeat_interval` seconds (spartan/worker.py:347-368) and the
master declares a worker failed once its last report is older than `heartbeat_interval *
worker_failed_heartbeat_threshold` (spartan/master.py:142-146), recording every tile it held as bad
(master.py:134-140); cached values with bad tiles are then reloaded from a checkpoint or recomputed
(spartan/expr/operator/base.py:193-203).

Here a worker is a GPU driven by one process, and "alive" means that GPU still completes work: the beat of a rank
is a round trip through its device (an event recorded on a stream of its own and waited for -- a hung or reset GPU
stops the beats even if the process lives).  Beats are counters in a key-value store every rank can read: the
store of the job's control plane when there are several ranks (`World.store()`: the rendezvous hub of rank 0,
spartan_amd/rendezvous.py), a dictionary for one process.  Every rank
watches every other rank, so all of them reach the same verdict without a master; a verdict is QUEUED and applied by
the driver thread at its next safe point (`Context.apply_failures`, called when an expression starts to evaluate) --
tile tables are never touched from the watcher thread.  Applying it is `Context.mark_failed_worker` for each logical
worker of the silent rank.
"""
import threading
import time


class _LocalStore(object):
  def __init__(self):
    self._d = {}
    self._lock = threading.Lock()

  def set(self, key, value):
    with self._lock:
      self._d[key] = value

  def get(self, key):
    with self._lock:
      return self._d.get(key)

  def delete(self, key):
    with self._lock:
      self._d.pop(key, None)


class Heartbeat(object):
  """interval: seconds between beats; threshold: missed beats after which a rank is declared failed
  (reference flags heartbeat_interval = 3, worker_failed_heartbeat_threshold = 10, spartan/cluster.py:62-63)."""

  def __init__(self, ctx, interval=3.0, threshold=10, probe=None, store=None):
    self.ctx = ctx
    self.interval = float(interval)
    self.threshold = int(threshold)
    self.rank, self.size = ctx.world.rank, ctx.world.size
    self.store = store if store is not None else (ctx.world.store() if ctx.world.distributed else _LocalStore())
    self.probe = probe if probe is not None else getattr(ctx.backend, 'liveness_probe', lambda: True)
    self.failed_ranks = set()
    self._given_up = set()                    # ranks of earlier verdicts not heard from since (agree(); identical on every rank)
    self._carry = set()                       # a follower's own findings no verdict has taken up yet (agree())
    self.agree_floor_s = 15.0                 # least time the ranks wait for each other's verdicts at a safe point
    self._pending = []
    self._lock = threading.Lock()
    self._stop = threading.Event()
    self._paused = threading.Event()          # tests: a rank that stops reporting
    self._seen = {}                           # rank -> (last counter value, time it changed)
    self._threads = []

  # -- the two loops --------------------------------------------------------------------------------------
  def _beat_loop(self):
    beats = 0
    while not self._stop.is_set():
      if not self._paused.is_set():
        try:
          alive = self.probe()
        except Exception:
          alive = False
        if alive:
          beats += 1
          self.store.set('spartan_hb/%d' % self.rank, str(beats))
      self._stop.wait(self.interval)

  def _watch_loop(self):
    limit = self.interval * self.threshold
    while not self._stop.is_set():
      now = time.time()
      for r in range(self.size):
        try:
          value = self.store.get('spartan_hb/%d' % r)
        except Exception:
          value = None
        last = self._seen.get(r)
        if last is None or (value is not None and value != last[0]):
          self._seen[r] = (value, now)
          if last is not None:
            self.failed_ranks.discard(r)     # it beats again: a later silence is a new failure
        elif now - last[1] > limit and r not in self.failed_ranks:
          self.failed_ranks.add(r)
          with self._lock:
            self._pending.append(r)
      self._stop.wait(min(self.interval, 1.0))

  # -- control --------------------------------------------------------------------------------------------
  def start(self):
    for fn in (self._beat_loop, self._watch_loop):
      t = threading.Thread(target=fn, daemon=True)
      t.start()
      self._threads.append(t)
    return self

  def stop(self):
    self._stop.set()
    for t in self._threads:
      t.join(timeout=2 * self.interval + 1)

  def pause(self):
    self._paused.set()

  def resume(self):
    self._paused.clear()

  def take_failures(self):
    """Ranks declared failed since the last call (driver thread)."""
    with self._lock:
      out, self._pending = self._pending, []
    return out

  def agree(self, mine):
    """The verdict all ranks apply at this safe point, without a collective (a dead rank would never join one).  The
    driver programs are SPMD, so the safe points carry the same number on every rank; under that number

      phase 1  every rank posts the ranks ITS watcher declared silent and reads the post of EVERY other rank --
               whatever its own watcher thinks of them: the set of posts read must not depend on a local clock --
               until a deadline; a rank that posted nothing by then is added (dead or hung);
      phase 2  every rank posts the union it arrived at and takes the union of all unions posted.  A rank that
               reached the safe point late finds the others' posts still there, reads in them that it was given up
               on, and applies that verdict to ITSELF like everyone else (its workers' tiles become bad tiles on
               every rank including its own: the tile tables stay equal, the tiles are reloaded or recomputed).

    The posts of safe point N - 2 are deleted when N is posted (one small key per rank and phase is alive at a time).
    Deadlines are generous -- 3 x interval x threshold, at least `agree_floor_s`: a rank that is merely slow must
    not be declared dead; a really dead one costs the survivors this wait once: ranks of an earlier verdict are
    not waited for at later safe points, and are not reported again unless a watcher flags them anew.  A rank of an
    earlier verdict that posts again is counted on again from the safe point after one where a counted-on rank read
    its post (the `heard` lists of phase 2); until then it follows the others' verdict and decides nothing itself."""
    self._round = getattr(self, '_round', 0) + 1
    n = self._round
    key = lambda phase, rnd, rank: 'spartan_hb_agree/%d/%d/%d' % (rnd, phase, rank)     # noqa: E731
    if n > 2:
      for phase in (1, 2):
        self.store.delete(key(phase, n - 2, self.rank))
    wait_s = max(3 * self.interval * self.threshold, self.agree_floor_s)

    ints = lambda text: set(int(x) for x in text.split(',') if x not in ('-', ''))     # noqa: E731

    def post(phase, ranks, heard=()):
      text = ','.join(str(r) for r in sorted(ranks)) or '-'
      if phase == 2:
        text += '|' + (','.join(str(r) for r in sorted(heard)) or '-')
      self.store.set(key(phase, n, self.rank), text)

    def collect(phase, wait_for, also_read):
      """Posts of phase `phase`: waits (with the deadline) for the ranks in wait_for, takes what is there of the
      ranks in also_read; returns (union of what was read, ranks that never posted, the `heard` lists of the
      phase-2 posts read, the ranks of also_read whose post was there)."""
      union, heard, there, waiting, deadline = set(), set(), set(), list(wait_for), time.time() + wait_s

      def take(value):
        failed, _, back = value.partition('|')
        union.update(ints(failed))
        heard.update(ints(back))
      while waiting:
        for r in list(waiting):
          value = self._read(key(phase, n, r))
          if value is not None:
            take(value)
            waiting.remove(r)
        if not waiting or time.time() > deadline:
          break
        time.sleep(0.001)
      for r in also_read:
        value = self._read(key(phase, n, r))
        if value is not None:
          take(value)
          there.add(r)
      return union, waiting, heard, there

    # Ranks every rank gave up on at an earlier safe point (`_given_up` only changes here, by what all ranks read in
    # the same phase-2 posts, so it is the same set everywhere) are not waited for: a dead rank costs the survivors
    # the deadline ONCE.  A given-up rank that is still there -- it only stopped beating, or it was late -- keeps
    # running the same program and keeps posting.  Nobody waits for its posts, so nothing may depend on them alone:
    #   * it is a FOLLOWER: its verdict is what the ranks still counted on posted in phase 2, nothing of its own
    #     (what its watcher flagged is offered in phase 1 and carried to the next safe point until a verdict has it);
    #   * a counted-on rank that happened to read its phase-1 post takes the content into its own union (so it
    #     reaches everyone through phase 2) and names the rank in the `heard` list of its phase-2 post; the union of
    #     those lists is the same on every rank, and the ranks in it are counted on again from the next safe point.
    given_up = self._given_up
    others = [r for r in range(self.size) if r != self.rank]
    counted_on = [r for r in others if r not in given_up]
    follower = self.rank in given_up
    mine = set(mine) | self._carry
    post(1, mine)
    seen, absent, _, there = collect(1, counted_on, [r for r in others if r in given_up])
    if follower:
      post(2, (), ())                          # (nobody reads it; keeps the store's key pattern uniform)
      final, absent2, heard, _ = collect(2, counted_on, ())
      verdict = final | set(absent2)
      self._carry = mine - verdict - {self.rank}
    else:
      union = mine | seen | set(absent)
      post(2, union, there)
      # the unions of the ranks this one still counts on are waited for (ranks in `union` are in everybody's union)
      final, absent2, heard, _ = collect(2, [r for r in counted_on if r not in union], ())
      verdict = union | final | set(absent2)
      heard |= there
      self._carry = set()
    given_up.difference_update(heard)
    given_up.update(verdict)
    self.failed_ranks.update(r for r in verdict if r != self.rank)
    return sorted(verdict)

  def _read(self, key):
    try:
      return self.store.get(key)
    except Exception:
      return None
