"""One benchmark episode = full plan + N replans on one map instance, and its multi-rank form.

Independent map instances shard across ranks (one process per GPU, no data-path collective
for the maps themselves).  The patch stream is owned by rank 0 and reaches the other ranks by a
broadcast before every replan -- the only exchange the path has (RCCL over xGMI on GPUs, gloo in
the CPU tests).  The planner object only needs the reference surface (ReplannerBase.h:39-123)
plus `apply_patch(buffer, top, left)`; bench.py passes the HIP planner, the CPU tests the oracle.
"""
import time


class PatchStream:
    """Rank 0 holds all patches; `fetch(i)` returns patch i on every rank.  Per-patch views are made
    once (slicing a tensor costs microseconds, a replan a few hundred).

    `pipeline=True` (needs two receive buffers): the stream is a recording, so the broadcast of patch
    i+1 is issued (asynchronously, from the calling thread) when patch i is handed out and travels
    while the caller replans with patch i; the consumer waits for it on the device, not on the host.
    All broadcasts stay inside whatever region the caller times; only the collective's latency
    leaves the replans' critical path.  (A helper thread issuing the broadcasts was measured and
    dropped: Python's thread hand-offs cost more than the collective's enqueue they were to hide.)"""

    def __init__(self, patches, recv_buffer, dist=None, rank=0, sync=None, pipeline=False, count=None):
        self.patches = None if patches is None else [patches[i] for i in range(len(patches))]
        self.recv = list(recv_buffer) if isinstance(recv_buffer, (list, tuple)) else [recv_buffer]
        self.dist = dist
        self.rank = rank
        self.sync = sync                # callable making the broadcast result visible to the consumer
        self.broadcasts = 0
        self.count = count if count is not None else (len(self.patches) if self.patches is not None else None)
        self.pipeline = bool(pipeline) and dist is not None and len(self.recv) >= 2 and self.count is not None
        self._inflight = {}             # patch index -> (work handle, buffer)

    def _buffer(self, i):
        # rank 0 sends straight from its patch store (no staging copy); the others receive into a buffer
        return self.patches[i] if self.rank == 0 else self.recv[i % len(self.recv)]

    def _issue(self, i):
        buf = self._buffer(i)
        self._inflight[i] = (self.dist.broadcast(buf, src=0, async_op=True), buf)

    def close(self):
        for work, _ in self._inflight.values():
            work.wait()
        self._inflight.clear()

    def fetch(self, i):
        if self.dist is None:
            return self.patches[i]
        if not self.pipeline:
            buf = self._buffer(i)
            self.dist.broadcast(buf, src=0)
            self.broadcasts += 1
            if self.sync is not None:
                self.sync()
            return buf
        if i not in self._inflight:
            self._issue(i)
        work, buf = self._inflight.pop(i)
        work.wait()                     # RCCL: the current stream waits, the host does not; gloo: the host waits
        self.broadcasts += 1
        if self.sync is not None:
            self.sync()
        if i + 1 < self.count:
            self._issue(i + 1)          # travels while the caller replans with patch i
        return buf


def run_episode(planner, set_map, start, goal, script, stream, apply_patch, read_stats):
    """script: list of (k, start_xy, top, left).  Returns the summed statistics dict."""
    tot = {}

    def acc():
        for k, v in read_stats(planner).items():
            tot[k] = tot.get(k, 0) + v

    set_map(planner)
    planner.reset()
    planner.set_start(*start)
    planner.set_goal(*goal)
    rc = planner.step()
    if rc != 0:
        raise RuntimeError("plan step failed: %d" % rc)
    acc()
    for i, (k, s, top, left) in enumerate(script):
        apply_patch(planner, stream.fetch(i), top, left)
        planner.set_start(*s)
        rc = planner.step()
        if rc != 0:
            raise RuntimeError("replan %d failed: %d" % (k, rc))
        acc()
    return tot


def timed_episodes(run_one, steps, warmup, barrier):
    """The bench.py timing contract: W untimed warm-up steps, then exactly K steps bracketed by
    barrier(); returns (seconds, list of per-step statistics)."""
    for _ in range(warmup):
        run_one()
    barrier()
    t0 = time.perf_counter()
    out = [run_one() for _ in range(steps)]
    barrier()
    return time.perf_counter() - t0, out
