"""-m gpu: the planner's life cycle and degenerate inputs against the oracle's restatement of ReplannerBase::step
(ReplannerBase.h:43-75: a new goal or reset() re-initialises the search, set_map re-initialises the graph, a new start only moves the
end condition) -- the sequences a mission produces around the hot path: goals that change, maps that are replaced by another size,
starts and goals on obstacles, maps without a free cell, a start that a patch walls in and a later patch frees again."""
import numpy as np
import pytest

import oracle_py as orc
import ufm_amd
from helpers import ALGOS, check_parity

pytestmark = pytest.mark.gpu

PLANNERS = [("FD", 1), ("SG", 2), ("DFM", 1), ("FD", 0)]


def pair(algo, lvl, heuristic=False):
    return orc.OraclePlanner(ALGOS[algo], lvl, heuristic), ufm_amd.Planner(ALGOS[algo], lvl, heuristic)


def both(o, g, fn):
    for p in (o, g):
        fn(p)


def step_and_check(o, g, what, exact=True):
    ro, rg = o.step(), g.step()
    assert ro == rg, (what, ro, rg)
    if ro != 0:
        return
    n, nbad = check_parity(o, g, what, below_start_key=True)
    if exact and o.algo != 2:
        assert nbad == 0, what
    assert g.num_nodes_updated == o.num_updated, (what, g.num_nodes_updated, o.num_updated)


@pytest.mark.parametrize("algo,lvl", PLANNERS)
def test_goal_changes_and_reset_between_steps(algo, lvl):
    size = 192
    cost = ufm_amd.synth.cost_map(31, size, size)
    o, g = pair(algo, lvl)
    both(o, g, lambda p: (p.reset(), p.set_occupancy_threshold(1.0), p.set_map(cost), p.set_start(10.0, 12.0), p.set_goal(170.0, 160.0)))
    step_and_check(o, g, "%s-%d plan" % (algo, lvl))
    script = list(ufm_amd.synth.replan_script(31, size, size, n_patches=6))
    goals = [(170.0, 160.0), (30.0, 150.0), (30.0, 150.0), (120.0, 20.0), (120.0, 20.0), (170.0, 160.0)]
    for i, (k, s, top, left, patch) in enumerate(script):
        both(o, g, lambda p: (p.patch_map(patch, top, left), p.set_start(*s), p.set_goal(*goals[i])))       # a new goal re-initialises the search (:56-62)
        if i == 3:
            both(o, g, lambda p: p.reset())                                                                  # ... and so does reset() (:39-41)
        step_and_check(o, g, "%s-%d step %d goal %r" % (algo, lvl, i, goals[i]))
    # the same goal again, nothing pending: nothing to do
    both(o, g, lambda p: p.set_goal(*goals[-1]))
    before = g.g()
    step_and_check(o, g, "%s-%d idle step" % (algo, lvl))
    assert g.num_nodes_expanded == 0 and np.array_equal(before, g.g())
    g.close()


@pytest.mark.parametrize("algo,lvl", PLANNERS)
def test_map_replaced_by_another_size_on_the_same_handle(algo, lvl):
    o, g = pair(algo, lvl)
    both(o, g, lambda p: (p.reset(), p.set_occupancy_threshold(1.0)))
    for (width, length, seed) in [(96, 80, 3), (200, 136, 4), (33, 47, 5), (200, 136, 6)]:
        cost = ufm_amd.synth.cost_map(seed, width, length)
        start, goal = (6.0, 5.0), (float(length - 7), float(width - 6))
        both(o, g, lambda p: (p.set_map(cost), p.set_start(*start), p.set_goal(*goal)))
        step_and_check(o, g, "%s-%d map %dx%d" % (algo, lvl, width, length))
        patch = np.full((9, 9), 200, np.uint8)
        both(o, g, lambda p: (p.patch_map(patch, length // 2 - 4, width // 2 - 4), p.set_start(*start)))
        step_and_check(o, g, "%s-%d map %dx%d patched" % (algo, lvl, width, length))
        assert np.array_equal(g.read_map(width, length)[length // 2 - 4:length // 2 + 5, width // 2 - 4:width // 2 + 5], patch)
    g.close()


@pytest.mark.parametrize("algo,lvl", PLANNERS)
def test_start_and_goal_on_obstacles_and_a_map_without_a_free_cell(algo, lvl):
    size = 72
    base = np.full((size, size), 9, np.uint8)
    cases = {}
    c = base.copy(); c[8:12, 8:12] = 255; cases["start on an obstacle block"] = (c, (10.0, 10.0), (60.0, 60.0))
    c = base.copy(); c[58:63, 58:63] = 255; cases["goal on an obstacle block"] = (c, (10.0, 10.0), (60.0, 60.0))
    c = np.full((size, size), 255, np.uint8); cases["no free cell"] = (c, (10.0, 10.0), (60.0, 60.0))
    c = base.copy(); cases["start == goal"] = (c, (33.0, 35.0), (33.0, 35.0))
    c = base.copy(); c[:, 40] = 255; cases["goal behind a wall"] = (c, (10.0, 10.0), (60.0, 60.0))
    for name, (cost, start, goal) in cases.items():
        o, g = pair(algo, lvl)
        both(o, g, lambda p: (p.reset(), p.set_occupancy_threshold(1.0), p.set_map(cost), p.set_start(*start), p.set_goal(*goal)))
        ro, rg = o.step(), g.step()
        assert ro == rg == 0, (name, ro, rg)
        og, gg = o.g(), g.g()
        # what the oracle reached the engine reached with the same value (FD / SG bit for bit), and nothing else holds a value
        # below the start's key; with an unreachable start the key is +inf and both expand whatever the goal's side holds
        fin = np.isfinite(og) & (og == o.rhs())
        if np.isfinite(o.start_key()) and int(o.trusted_mask(below_start_key=True).sum()) == 0:
            # (MS-DFM, start == goal: the start's key is the goal's, 0 -- nothing lies below it)
            gx, gy = int(np.floor(goal[0] + 0.5)), int(np.floor(goal[1] + 0.5))
            assert gg[gx, gy] == 0.0 and og[gx, gy] == 0.0, name
        elif np.isfinite(o.start_key()):
            check_parity(o, g, "%s-%d %s" % (algo, lvl, name), below_start_key=True)
        else:
            assert np.isfinite(gg[fin]).all(), name
            if algo == "DFM":
                assert np.allclose(gg[fin], og[fin], rtol=2e-6, atol=0), name
            else:
                assert np.array_equal(gg[fin], og[fin]), name
            assert not np.isfinite(gg[~np.isfinite(og) & ~np.isfinite(o.rhs())]).any(), name
        # a path query must not hang or fault on these
        pts, costs, tc, td = g.extract_path(max_steps=200, allow_indirect=True)
        assert len(pts) <= 201
        g.close()


@pytest.mark.parametrize("algo,lvl", [("FD", 1), ("SG", 2), ("DFM", 1)])
def test_start_walled_in_by_a_patch_and_freed_by_the_next(algo, lvl):
    size = 128
    cost = ufm_amd.synth.cost_map(17, size, size, obstacles=False)
    o, g = pair(algo, lvl, heuristic=True)
    hm = float(cost.min())
    start, goal = (20.0, 22.0), (110.0, 100.0)
    both(o, g, lambda p: (p.reset(), p.set_occupancy_threshold(1.0), p.set_heuristic_multiplier(hm), p.set_map(cost), p.set_start(*start), p.set_goal(*goal)))
    step_and_check(o, g, "plan")
    ring = cost[10:33, 10:33].copy()
    wall = ring.copy(); wall[0, :] = wall[-1, :] = 255; wall[:, 0] = wall[:, -1] = 255
    both(o, g, lambda p: (p.patch_map(wall, 10, 10), p.set_heuristic_multiplier(hm), p.set_start(*start)))
    ro, rg = o.step(), g.step()
    assert ro == rg == 0
    assert not np.isfinite(o.start_key())
    gg, og = g.g(), o.g()
    assert np.isinf(gg[12:31, 12:31]).all() and np.isinf(og[12:31, 12:31]).all()           # nothing reaches the inside
    outside = np.isfinite(og) & (og == o.rhs())
    outside[8:35, 8:35] = False
    if algo == "DFM":
        assert np.allclose(gg[outside], og[outside], rtol=2e-6, atol=0)
    else:
        assert np.array_equal(gg[outside], og[outside])
    both(o, g, lambda p: (p.patch_map(ring, 10, 10), p.set_heuristic_multiplier(hm), p.set_start(*start)))
    step_and_check(o, g, "freed again")
    assert np.isfinite(g.g()[20, 22])
    g.close()


def test_host_patches_held_for_the_block_kernel():
    """ufm_patch_map of a small patch (single planner) is held in pinned memory and applied by the replan's block kernel itself (round 4:
    no staging copy, no patch kernel).  What a caller can observe must not change -- held against the same engine with "lazy_patches" 0
    (every patch uploaded and applied at the call), bit for bit: the caller's buffer is free when the call returns (overwritten here), a read
    of the raster between patch and step sees the patch, two patches before one step both count and keep their order where they overlap
    (the engine accumulates them; the reference keeps the last change list only, Graph.cpp:37), host and device patches mix, and a step
    that does not consume the patches (no new start) leaves them applied to the raster but not propagated (ReplannerBase.h:43-75).
    (Against the ORACLE the held patches are covered by every parity test of a single planner: they all hand their patches over from host memory.)"""
    from helpers import DeviceBytes
    size = 192
    cost = ufm_amd.synth.cost_map(31, size, size)
    start, goal = ufm_amd.synth.start_goal(size, size)
    ga, gb = ufm_amd.Planner(ufm_amd.ALGO_FD, 1, False), ufm_amd.Planner(ufm_amd.ALGO_FD, 1, False)
    ga.set_param("lazy_patches", 1); gb.set_param("lazy_patches", 0)
    for p in (ga, gb):
        p.reset(); p.set_occupancy_threshold(1); p.set_map(cost); p.set_start(*start); p.set_goal(*goal)
        assert p.step() == 0
    expect = cost.copy()
    rng = np.random.default_rng(5)
    keep = []
    for k in range(9):
        top, left = int(rng.integers(20, size - 60)), int(rng.integers(20, size - 60))
        patch = rng.integers(1, 201, (31, 31)).astype(np.uint8)
        for p in (ga, gb):
            buf = patch.copy()
            p.patch_map(buf, top, left)
            buf[:] = 7                                            # the caller's buffer is its own again
        expect[top:top + 31, left:left + 31] = patch
        if k % 3 == 1:                                            # a second, overlapping patch before the step
            p2 = rng.integers(1, 201, (20, 17)).astype(np.uint8)
            for p in (ga, gb):
                p.patch_map(p2.copy(), top + 5, left + 7)
            expect[top + 5:top + 25, left + 7:left + 24] = p2
        if k == 2:                                                # a device patch behind a held host patch
            p3 = rng.integers(1, 201, (9, 9)).astype(np.uint8)
            d3 = DeviceBytes(p3); keep.append(d3)
            for p in (ga, gb):
                p.patch_map_device(d3.data_ptr(), top + 1, left + 1, 9, 9)
            expect[top + 1:top + 10, left + 1:left + 10] = p3
        if k == 4:                                                # the raster is read before the step
            assert np.array_equal(ga.read_map(size, size), expect)
        if k == 5:                                                # a step without a new start: patches applied, not consumed
            for p in (ga, gb):
                assert p.step() == 0
            assert np.array_equal(ga.read_map(size, size), expect)
        cur = (float(8 + 3 * k), float(8 + 2 * k))
        for p in (ga, gb):
            p.set_start(*cur)
            assert p.step() == 0
        what = "step %d" % k
        assert ga.num_nodes_updated == gb.num_nodes_updated, (what, ga.num_nodes_updated, gb.num_nodes_updated)
        fa, fb = ga.g(), gb.g()
        key = min(float(fa[int(cur[0]) + i, int(cur[1]) + j]) for i in (0, 1) for j in (0, 1))
        m = np.isfinite(fb) & (fb < key)
        assert m.sum() > 1000 and np.array_equal(fa[m], fb[m]), what
        assert np.array_equal(ga.read_map(size, size), expect) and np.array_equal(gb.read_map(size, size), expect), what
        assert ga.check_layout() == (0, 0) and ga.check_info()[1:4] == (0, 0, 0), what
    if hasattr(ga.stats, "region_replans"):
        assert ga.stats.region_replans >= 6                   # the held patches did go through the block kernel
    ga.close(); gb.close()
