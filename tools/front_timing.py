"""Diagnostic (GPU box, -DUFM_TIMING build, --lib build/exp/libufm_timing.so): when does the resident plan kernel reach a tile, when does
the tile get its last change, and how long do its activations wait for the owner.  FD-1 full plan."""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ufm_amd

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=4096)
ap.add_argument("--seed", type=int, default=7)
ap.add_argument("--param", action="append", default=[])
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--lib", default="build/exp/libufm_timing.so")
ap.add_argument("--ys", type=int, default=4, help="log2 of the owner pattern's second edge: 4 (256 owners, 16 waves per visit) or 5 (512, 8 waves)")
a = ap.parse_args()
ufm_amd.use_library(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), a.lib))
L = ufm_amd.load_library()
L.ufm_debug_tiles.argtypes = [C.c_void_p, C.c_int]
cost = ufm_amd.synth.cost_map(a.seed, a.size, a.size)
start, goal = ufm_amd.synth.start_goal(a.size, a.size)
p = ufm_amd.Planner(ufm_amd.ALGO_FD, 1, False)
p.set_occupancy_threshold(1)
for kv in a.param:
    n, v = kv.split("=")
    p.set_param(n, float(v))
T = L.ufm_tile_edge()
TX = TY = (a.size + 1 + T - 1) // T
NT = TX * TY
L.ufm_debug_sdiag.argtypes = [C.c_void_p, C.c_int]
for rep in range(a.reps):
    p.set_map(cost); p.reset(); p.set_start(*start); p.set_goal(*goal)
    L.ufm_debug_sdiag(None, 1)
    assert p.step() == 0
sd = (C.c_ulonglong * 16)()
L.ufm_debug_sdiag(sd, 0)
print("looks of idle workgroups %d, nothing to take %d | helping: attempts %d, victim had a free queued tile %d, inside the band %d, taken %d | failed takes: ahead %d, fresh %d" % tuple(sd[i] for i in range(8)))
buf = np.zeros((5, NT), np.uint32)
assert L.ufm_debug_tiles(buf.ctypes.data, NT) == NT
first, last, _, vis, wsum = [buf[i].astype(np.float64) for i in range(5)]
seen = buf[3] > 0
first, last, wsum = first / 100.0, last / 100.0, wsum / 100.0      # us
tx, ty = np.divmod(np.arange(NT), TY)
gx, gy = int(goal[0]) // T, int(goal[1]) // T
d = np.maximum(np.abs(tx - gx), np.abs(ty - gy))
print("tiles visited %d of %d, visits %d (%.2f per tile); first visit of the last tile at %.0f us, last change at %.0f us" % (
    seen.sum(), NT, vis[seen].sum(), vis[seen].mean(), first[seen].max(), last[seen].max()))
settle = (last - first)[seen & (last > 0)]
print("settling (last change - first visit start) us: median %.0f  mean %.0f  p90 %.0f  p99 %.0f  max %.0f" % (
    np.median(settle), settle.mean(), np.percentile(settle, 90), np.percentile(settle, 99), settle.max()))
print("activation -> visit wait: %.2f us per visit (sum %.0f us over %d visits)" % (wsum[seen].sum() / vis[seen].sum(), wsum[seen].sum(), vis[seen].sum()))
# the front: first arrival / last change against the distance from the goal tile
print("distance(tiles)  tiles  first: median  min  max | last change: median max | settle median")
for lo in range(0, int(d[seen].max()) + 1, 16):
    s = seen & (d >= lo) & (d < lo + 16) & (last > 0)
    if s.sum() == 0:
        continue
    print("  %3d..%3d  %6d   %8.0f %8.0f %8.0f | %8.0f %8.0f | %6.0f" % (lo, lo + 15, s.sum(), np.median(first[s]), first[s].min(), first[s].max(),
                                                                         np.median(last[s]), last[s].max(), np.median((last - first)[s])))
# how fast does the first arrival travel: regression of first on d over the rings
ds = np.arange(0, int(d[seen].max()) + 1)
mf = np.array([first[seen & (d == k)].min() if (seen & (d == k)).any() else np.nan for k in ds])
ml = np.array([last[seen & (d == k) & (last > 0)].max() if (seen & (d == k) & (last > 0)).any() else np.nan for k in ds])
ok = ~np.isnan(mf)
print("earliest first visit per ring: %.1f us per tile of distance; latest last change per ring: %.1f us per tile" % (
    np.polyfit(ds[ok], mf[ok], 1)[0], np.polyfit(ds[ok], ml[ok], 1)[0]))
print("stats: tile_visits %d kernel %.2f ms" % (p.stats.tile_visits, p.stats.resident_kernel_ms))

# ---- the critical path: from the visit that ended last back along "the activation this visit took was sent by ..." ----
L.ufm_debug_visits.argtypes = [C.c_void_p, C.c_int]
vbuf = np.zeros((1 << 20, 5), np.uint32)
nv = L.ufm_debug_visits(vbuf.ctypes.data, 1 << 20)
v = vbuf[:nv].astype(np.int64)
v_gt, v_s, v_e, v_pt, v_from = v[:, 0], v[:, 1], v[:, 2], v[:, 3], v[:, 4]
own = (((v_gt // TY) & 15) << a.ys) | ((v_gt % TY) & ((1 << a.ys) - 1))
by_tile, by_own = {}, {}
order = np.argsort(v_s)
for i in order:
    by_tile.setdefault(int(v_gt[i]), []).append(int(i))
    by_own.setdefault(int(own[i]), []).append(int(i))
cur = int(np.argmax(np.where(v_from != 0xFFFFFFFF, v_e, 0)))     # (the last visit that took an activation somebody sent)
links = []
inferred = 0
goal_tile = gx * TY + gy
while True:
    f, pt = int(v_from[cur]), int(v_pt[cur])
    cands = [i for i in by_tile.get(f, []) if v_s[i] <= pt] if f != 0xFFFFFFFF else []
    if not cands:
        # The record of who sent the activation is missing: own_push() writes the queue word and the diagnostic record with two atomics, and a
        # visitor that takes the word between them finds no record (round 3: the walk ended there, after ~30 links).  Inferred instead: among the
        # visits of the eight neighbouring tiles and of the tile itself that ended before this one began, the one that ended last -- a visit sends
        # its activations at its end, so that is the activation this visit was waiting for.
        t = int(v_gt[cur]); tx_, ty_ = divmod(t, TY)
        best = -1
        for dx in (-1, 0, 1):
            for dy in (-1, 0, 1):
                nx_, ny_ = tx_ + dx, ty_ + dy
                if not (0 <= nx_ < TX and 0 <= ny_ < TY):
                    continue
                for i in by_tile.get(nx_ * TY + ny_, []):
                    if i != cur and v_e[i] and v_e[i] <= v_s[cur] and (best < 0 or v_e[i] > v_e[best]):
                        best = i
        if best < 0:
            break
        cands, pt = [best], int(v_e[best])
        inferred += 1
    pred = cands[-1]
    # how much of the wait was the owner busy with other tiles
    busy = same = 0
    for i in by_own[int(own[cur])]:
        if i == cur:
            continue
        lo, hi = max(int(v_s[i]), pt), min(int(v_e[i]) if v_e[i] else int(v_s[i]), int(v_s[cur]))
        if hi > lo:
            busy += hi - lo
    for i in by_tile[int(v_gt[cur])]:           # ... or the tile itself was still being visited (by whoever) when the activation came
        if i == cur:
            continue
        lo, hi = max(int(v_s[i]), pt), min(int(v_e[i]) if v_e[i] else int(v_s[i]), int(v_s[cur]))
        if hi > lo:
            same += hi - lo
    links.append((cur, pred, (int(v_s[cur]) - pt) / 100.0, (pt - int(v_s[pred])) / 100.0, busy / 100.0, (int(v_e[pred]) - int(v_s[pred])) / 100.0, same / 100.0))
    if pred == cur:
        break
    cur = pred
    if len(links) > 100000:
        break
if not links:
    sys.exit(0)
la = np.array([(w, sp, b, d, sm) for (_c, _p, w, sp, b, d, sm) in links])
print("critical path: %d links back from the last visit (ends %.0f us) to a visit starting at %.0f us (tile %d; the goal's tile is %d); %d links inferred (sender record missing)" % (
    len(links), v_e.max() / 100.0, v_s[cur] / 100.0, int(v_gt[cur]), goal_tile, inferred))
print("  per link: activation sent %.1f us after the sender's visit began (its visit lasted %.1f us) + waited %.1f us for its own visit (owner busy with other tiles %.1f us of that)" % (
    la[:, 1].mean(), la[:, 3].mean(), la[:, 0].mean(), la[:, 2].mean()))
print("  sums: in sender visits %.0f us, waiting %.0f us (owner busy %.0f us; the tile itself still in an earlier visit %.0f us)" % (la[:, 1].sum(), la[:, 0].sum(), la[:, 2].sum(), la[:, 4].sum()))
print("  wait histogram (us) <2 <5 <10 <20 <40 <80 >=80: %s" % np.histogram(la[:, 0], bins=[-1e9, 2, 5, 10, 20, 40, 80, 1e9])[0])
self_links = sum(1 for (c, p_, *_r) in links if v_gt[c] == v_gt[p_])
print("  links where a tile re-queued itself: %d; distinct tiles on the path: %d" % (self_links, len(set(int(v_gt[c]) for (c, *_r) in links))))
