"""The queue as a caller of the reference can observe it between two steps (ReplannerBase::priority_queue, a public member:
ReplannerBase.h:154; PriorityQueue.h:47-63) -- ufm_read_queue and the C++ mirror's PriorityQueue view.

The reference's queue holds exactly the elements that are not consistent (enqueue_if_inconsistent, ReplannerBase.h:110-115), and when
step() returns none of them has a key below the start's (end_condition()).  The engine derives the same set from its field -- RHS by
min_rhs<level>() as the path code evaluates it, an implementation independent of the relaxation kernels -- so an empty view below the
start's key is also a whole-field check of the fixed point.  WHICH elements wait beyond the key depends on the order of the expansions
there (in the reference as here): membership is not compared, the invariants are."""
import numpy as np
import pytest

import ufm_amd
from helpers import ALGOS, make_pair, check_parity
from helpers import DFM_RTOL

pytestmark = pytest.mark.gpu


def keys_of(p_xy, g, rhs, start, hm, heuristic, cells):
    k2 = np.minimum(g, rhs)
    if not heuristic:
        return k2, k2
    sx, sy = (float(np.floor(start[0] + 0.5)), float(np.floor(start[1] + 0.5))) if cells else start
    d = np.hypot(np.float32(sx) - p_xy[:, 0].astype(np.float32), np.float32(sy) - p_xy[:, 1].astype(np.float32)).astype(np.float32)
    return (k2 + np.float32(hm) * d).astype(np.float32), k2


@pytest.mark.parametrize("algo,lvl,heur", [("FD", 1, True), ("FD", 0, False), ("SG", 2, True), ("SG", 1, False), ("DFM", 1, True), ("DFM", 0, False)])
def test_queue_view_invariants(algo, lvl, heur):
    width = length = 320
    seed = 21
    cost = ufm_amd.synth.cost_map(seed, width, length)
    start, goal = ufm_amd.synth.start_goal(width, length)
    start = (150.0, 170.0)                    # a start in the middle: a focused search leaves most of the far side unexpanded
    hm = float(cost.min())
    o, g = make_pair(ALGOS[algo], lvl, cost, start, goal, heuristic=heur, hm=hm)
    script = list(ufm_amd.synth.replan_script(seed, width, length, n_patches=6))
    cur = start
    for step in range(1 + len(script)):
        if step:
            _k, s, top, left, patch = script[step - 1]
            cur = (float(min(max(s[0], 20), length - 20)), float(min(max(s[1], 20), width - 20))) if step % 2 else cur
            for p in (o, g):
                p.patch_map(patch, top, left)
                p.set_start(*cur)
        assert o.step() == 0 and g.step() == 0
        what = "%s-%d heur %d step %d" % (algo, lvl, heur, step)
        check_parity(o, g, what, below_start_key=True)
        # the reference's invariant, on the restatement: the queue is the set of inconsistent elements, its top is not below the start's key
        og, orhs = o.g(), o.rhs()
        o_incons = ~((og == orhs) | (np.isinf(og) & np.isinf(orhs)))
        assert o.queue_size == int(o_incons.sum()), what
        skey = o.start_key()
        if o.queue_size:
            assert o.top_key()[0] >= skey or not np.isfinite(skey), what
        # the engine's view
        xy, qg, qrhs, total = g.read_queue()
        assert total == len(xy) == len(qg) == len(qrhs), what
        field = g.g()
        assert np.array_equal(field[xy[:, 0], xy[:, 1]], qg), what              # G of the entries is the field's
        assert not np.any((qg == qrhs) | (np.isinf(qg) & np.isinf(qrhs))), what  # every entry is inconsistent
        gx, gy = (int(np.floor(goal[0] + 0.5)), int(np.floor(goal[1] + 0.5)))
        assert not np.any((xy[:, 0] == gx) & (xy[:, 1] == gy)), what             # the goal (RHS = G = 0) never waits
        k1, _k2 = keys_of(xy, qg, qrhs, cur, hm, heur, algo == "DFM")
        if total:
            slack = DFM_RTOL * skey if algo == "DFM" else 0.0
            assert k1.min() >= skey - slack, "%s: an element with key %r waits below the start's key %r" % (what, float(k1.min()), skey)
        if algo == "DFM":
            # never-expanded frontier cells (G = +inf, RHS finite) are the bulk of the reference's queue after a focused step: they must be in the
            # view (round 3 dropped them: the tolerance test `|g - r| <= rtol * |g|` read inf <= inf), and every cell the oracle leaves unexpanded
            # between finalised neighbours is one of them wherever the engine has not given it a value either
            inf_entries = np.isinf(qg) & np.isfinite(qrhs)
            assert int(inf_entries.sum()) > 0, what
            tm = o.trusted_mask(below_start_key=True)
            allnb = np.zeros_like(tm)
            allnb[1:-1, 1:-1] = np.logical_and.reduce([tm[1 + dx:tm.shape[0] - 1 + dx, 1 + dy:tm.shape[1] - 1 + dy] for dx in (-1, 0, 1) for dy in (-1, 0, 1) if dx or dy])
            need = np.isinf(og) & np.isfinite(orhs) & allnb & np.isinf(field)
            inview = set(map(tuple, xy[np.isinf(qg)].tolist()))
            missing = [tuple(c) for c in np.argwhere(need).tolist() if tuple(c) not in inview]
            assert not missing, (what, missing[:5], int(need.sum()))
        # a capped read: the same count, the first `cap` entries
        if total > 3:
            xy3, g3, r3, t3 = g.read_queue(cap=3)
            assert t3 == total and len(xy3) == 3, what
    g.close()


def test_queue_view_full_plan_is_empty_below_the_key_everywhere():
    """an unfocused plan (no heuristic, start in the far corner): next to everything lies below the start's key, so the view is a
    whole-field check of RHS == G by code that shares nothing with the relaxation kernels"""
    size = 1024
    cost = ufm_amd.synth.cost_map(5, size, size)
    start, goal = ufm_amd.synth.start_goal(size, size)
    for algo, lvl in (("FD", 1), ("SG", 2)):
        p = ufm_amd.Planner(ALGOS[algo], lvl, False)
        p.set_occupancy_threshold(1); p.set_map(cost); p.set_start(*start); p.set_goal(*goal)
        assert p.step() == 0
        f = p.g()
        sx, sy = int(start[0]), int(start[1])
        el = f[sx:sx + 2, sy:sy + 2]
        skey = float(el[np.isfinite(el)].max())
        xy, qg, qrhs, total = p.read_queue()
        assert int(np.isfinite(f).sum()) > 0.9 * f.size
        assert total < 2000, total                                  # only the corner behind the start waits
        if total:
            assert np.minimum(qg, qrhs).min() >= skey, algo
        p.close()
