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


def run_episode(planner, set_map, start, goal, script, stream, apply_patch, read_stats, phases=None):
    """script: list of (k, start_xy, top, left).  Returns the summed statistics dict -- or, when read_stats hands out opaque
    per-step snapshots instead of dicts (bench.py: a byte copy of the statistics struct, half a microsecond instead of the ten a
    dict of thirteen ctypes fields costs per replan), the list of them, for the caller to evaluate outside its timed region.
    phases (a dict): host wall seconds of the episode's parts are added to its "set_map", "plan" and "replans" entries."""
    tot = {}
    raw = []
    t0 = time.perf_counter()

    def acc():
        r = read_stats(planner)
        if isinstance(r, dict):
            for k, v in r.items():
                tot[k] = tot.get(k, 0) + v
        else:
            raw.append(r)

    set_map(planner)
    t1 = time.perf_counter()
    planner.reset()
    planner.set_start(*start)
    planner.set_goal(*goal)
    rc = planner.step()
    if rc != 0:
        raise RuntimeError("plan step failed: %d" % rc)
    acc()
    t2 = time.perf_counter()
    for i, (k, s, top, left) in enumerate(script):
        apply_patch(planner, stream.fetch(i), top, left)
        planner.set_start(*s)
        rc = planner.step()
        if rc != 0:
            raise RuntimeError("replan %d failed: %d" % (k, rc))
        acc()
    if phases is not None:
        t3 = time.perf_counter()
        for key, dt in (("set_map", t1 - t0), ("plan", t2 - t1), ("replans", t3 - t2)):
            phases[key] = phases.get(key, 0.0) + dt
    return raw if raw else tot


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


# ---- batch form (BASELINE config 4): M independent maps per rank, every map with a patch stream of its own ----
REC_HDR = 16    # bytes: int32 map id (global), top, left, patch edge


def pack_round(entries, psz):
    """One replan round of the whole job as the byte records rank 0 broadcasts: per global map one record
    {int32 map_id, top, left, edge; uint8 patch[edge][edge]}, padded to a multiple of 16 bytes.
    entries: list of (map_id, top, left, patch ndarray).  Returns a uint8 ndarray [n][rec]."""
    import numpy as np
    rec = (REC_HDR + psz * psz + 15) // 16 * 16
    out = np.zeros((len(entries), rec), np.uint8)
    for i, (g, top, left, patch) in enumerate(entries):
        out[i, :REC_HDR] = np.array([g, top, left, psz], np.int32).view(np.uint8)
        out[i, REC_HDR:REC_HDR + psz * psz] = patch.reshape(-1)
    return out


class RoundStream:
    """Rank 0 owns the patch streams of ALL maps of the job; per replan round it broadcasts one packed buffer
    (pack_round) and every rank takes the records of its own maps out of it.  `rounds` (rank 0): tensor
    [n_rounds][n_global_maps][rec]; `recv`: one or two tensors [n_global_maps][rec] on the other ranks.
    With pipeline=True (two receive buffers) round i+1 travels while the maps replan round i."""

    def __init__(self, rounds, recv, n_rounds, dist=None, rank=0, pipeline=False):
        self.rounds = None if rounds is None else [rounds[i] for i in range(n_rounds)]
        self.recv = list(recv) if isinstance(recv, (list, tuple)) else [recv]
        self.dist, self.rank, self.count = dist, rank, n_rounds
        self.pipeline = bool(pipeline) and dist is not None and len(self.recv) >= 2
        self.broadcasts = 0
        self._inflight = {}

    def _buffer(self, i):
        return self.rounds[i] if self.rank == 0 else self.recv[i % len(self.recv)]

    def _issue(self, i):
        buf = self._buffer(i)
        self._inflight[i] = (self.dist.broadcast(buf, src=0, async_op=True), buf)

    def close(self):
        for work, _ in self._inflight.values():
            work.wait()
        self._inflight.clear()

    def fetch(self, i):
        if self.dist is None:
            return self.rounds[i]
        if not self.pipeline:
            buf = self._buffer(i)
            self.dist.broadcast(buf, src=0)
            self.broadcasts += 1
            return buf
        if i not in self._inflight:
            self._issue(i)
        work, buf = self._inflight.pop(i)
        work.wait()
        self.broadcasts += 1
        if i + 1 < self.count:
            self._issue(i + 1)
        return buf


def run_batch_episode(batch, n_maps, first_map, set_maps, start, goal, starts, stream, headers_of, apply_record, read_stats, phases=None):
    """Full plan of the rank's n_maps maps (global ids first_map ..), then one batch step per replan round.
    starts[i]: the start position of round i; headers_of(i, buf) -> int32 ndarray [n_global][4] (the record headers of
    round i on the host); apply_record(batch, local_map, buf, global_map, top, left, edge); phases: as run_episode."""
    tot = {}
    raw = []
    t0 = time.perf_counter()

    def acc():           # (dicts are summed here, opaque snapshots handed back: see run_episode)
        r = read_stats(batch)
        if isinstance(r, dict):
            for k, v in r.items():
                tot[k] = tot.get(k, 0) + v
        else:
            raw.append(r)

    set_maps(batch)
    t1 = time.perf_counter()
    for m in range(n_maps):
        batch.reset(m)
        batch.set_start(m, *start)
        batch.set_goal(m, *goal)
    rc = batch.step()
    if rc != 0:
        raise RuntimeError("batch plan step failed: %d" % rc)
    acc()
    t2 = time.perf_counter()
    for i in range(stream.count):
        buf = stream.fetch(i)
        hdr = headers_of(i, buf)
        for m in range(n_maps):
            g, top, left, edge = (int(v) for v in hdr[first_map + m])
            if g != first_map + m:
                raise RuntimeError("round %d: record %d carries map id %d" % (i, first_map + m, g))
            apply_record(batch, m, buf, first_map + m, top, left, edge)
            batch.set_start(m, *starts[i])
        rc = batch.step()
        if rc != 0:
            raise RuntimeError("batch replan round %d failed: %d" % (i, rc))
        acc()
    if phases is not None:
        t3 = time.perf_counter()
        for key, dt in (("set_map", t1 - t0), ("plan", t2 - t1), ("replans", t3 - t2)):
            phases[key] = phases.get(key, 0.0) + dt
    return raw if raw else tot
