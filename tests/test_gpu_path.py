"""-m gpu: path extraction on the device (ufm_extract_path, through the C ABI) vs the oracle's
restatement of LinearInterpolationPathExtractor and vs the reference's own known answers."""
import numpy as np
import pytest

import oracle_py as orc
import ufm_amd
from helpers import ALGOS, make_pair

pytestmark = pytest.mark.gpu

# what the reference's extractor (max_steps = 800) returned on its own
# noise-trap bitmap: points, total_cost, total_dist; allow_indirect as the reference drivers set it
KNOWN_PATH = {"DFM": (154, 11808.9, 123.087, True), "SG": (146, 11721.3, 121.725, False),
              "FD": (146, 11721.3, 121.725, True)}
# reference drivers: Tests/Planners/{FDSTAR,DFM}/main.cpp:80-82 indirect, SGDFM/main.cpp:97 direct
INDIRECT = {"FD": True, "SG": False, "DFM": True}
PATH_RTOL = 1e-4        # path total_cost within 1e-4 relative


def same_path(a, b, what):
    """two extractions agree bit for bit"""
    pa, ca, tca, tda = a
    pb, cb, tcb, tdb = b
    assert pa.shape == pb.shape, "%s: %d vs %d points" % (what, len(pa), len(pb))
    assert np.array_equal(pa, pb), "%s: way points differ, first at %d" % (what, int(np.argmax((pa != pb).any(axis=1))))
    assert np.array_equal(ca, cb), "%s: step costs differ" % what
    assert tca == tcb and tda == tdb, "%s: totals differ (%r,%r) vs (%r,%r)" % (what, tca, tda, tcb, tdb)


def close_path(a, b, what):
    """device path on the device field vs oracle path on the oracle field"""
    pa, ca, tca, tda = a
    pb, cb, tcb, tdb = b
    assert len(pa) == len(pb), "%s: %d vs %d points" % (what, len(pa), len(pb))
    assert np.abs(pa - pb).max() <= 1e-3, "%s: way points differ by %g" % (what, np.abs(pa - pb).max())
    assert abs(tca - tcb) <= PATH_RTOL * abs(tcb), "%s: total_cost %r vs %r" % (what, tca, tcb)
    assert abs(tda - tdb) <= PATH_RTOL * abs(tdb) + 1e-5, "%s: total_dist %r vs %r" % (what, tda, tdb)


def close_path_while_final(a, b, o, what):
    """DFM under the reference's end condition: node values are averages of four cells, and next to
    the start some of those cells lie beyond the start's key -- never expanded (+inf) or stale in
    the reference, merely not final here.  The reference's own path through such cells is an
    artefact of its expansion order (it oscillates until max_steps), so the
    two chains are held to each other as far as the way points only touch final cells."""
    pa, pb = a[0], b[0]
    n = min(len(pa), len(pb))
    d = np.abs(pa[:n] - pb[:n]).max(axis=1) > 1e-3
    if not d.any() and len(pa) == len(pb):
        return close_path(a, b, what)
    i = int(np.argmax(d)) if d.any() else n
    assert i >= 1, what
    mask = o.trusted_mask(below_start_key=True)
    x, y = int(np.floor(pa[i - 1][0])), int(np.floor(pa[i - 1][1]))
    win = mask[max(x - 2, 0):x + 3, max(y - 2, 0):y + 3]
    assert not win.all(), "%s: paths part at %r although every cell around is final" % (what, tuple(pa[i - 1]))


def oracle_on_device_field(g, algo, cost, thr_uchar, start, goal, **kw):
    field = g.read_field()[1]
    return orc.extract_path_field(field, algo == "DFM", cost, thr_uchar, start, goal, **kw)


@pytest.mark.parametrize("algo", ["FD", "SG", "DFM"])
def test_noise_trap_path_known_answers(ref_bitmaps, algo):
    """the product against the numbers the reference itself produced"""
    cost, sg = ref_bitmaps["noise-trap"]
    o, g = make_pair(ALGOS[algo], 0, cost, sg[:2], sg[2:])
    assert g.step() == 0
    npts, tcost, tdist, indirect = KNOWN_PATH[algo]
    pts, costs, total_cost, total_dist = g.extract_path(max_steps=800, allow_indirect=indirect)
    assert len(pts) == npts
    assert abs(total_cost - tcost) < 0.06
    assert abs(total_dist - tdist) < 6e-4
    assert tuple(pts[0]) == (90.0, 90.0) and tuple(pts[-1]) == (25.0, 25.0)
    assert g.path_info.steps <= 800 and g.path_info.n_costs == len(costs)
    g.close()


@pytest.mark.parametrize("algo", ["FD", "SG", "DFM"])
@pytest.mark.parametrize("bitmap", ["noise-trap", "square", "wall-a", "wall-b"])
@pytest.mark.parametrize("indirect", [True, False])
def test_path_reference_bitmaps(ref_bitmaps, algo, bitmap, indirect):
    cost, sg = ref_bitmaps[bitmap]
    o, g = make_pair(ALGOS[algo], 0, cost, sg[:2], sg[2:])
    assert o.step() == 0 and g.step() == 0
    for max_steps, lookahead in ((20, True), (600, True), (600, False)):
        kw = dict(max_steps=max_steps, lookahead=lookahead, allow_indirect=indirect)
        dev = g.extract_path(**kw)
        what = "%s/%s steps=%d la=%d ind=%d" % (algo, bitmap, max_steps, lookahead, indirect)
        # same field, two extractors: bit for bit
        same_path(dev, oracle_on_device_field(g, algo, cost, 255, sg[:2], sg[2:], **kw), what + " [device field]")
        # the whole chain: planner + extractor on either side
        close_path(dev, o.extract_path(**kw), what)
    g.close()


@pytest.mark.parametrize("algo", ["FD", "SG", "DFM"])
@pytest.mark.parametrize("size", [(96, 96), (200, 136)])
def test_path_synthetic(algo, size):
    width, length = size
    cost = ufm_amd.synth.cost_map(77, width, length)
    start, goal = ufm_amd.synth.start_goal(width, length)
    o, g = make_pair(ALGOS[algo], 0, cost, start, goal)
    assert o.step() == 0 and g.step() == 0
    for max_steps in (20, 1000):
        kw = dict(max_steps=max_steps, lookahead=True, allow_indirect=INDIRECT[algo])
        dev = g.extract_path(**kw)
        what = "%s/%dx%d steps=%d" % (algo, width, length, max_steps)
        same_path(dev, oracle_on_device_field(g, algo, cost, 255, start, goal, **kw), what + " [device field]")
        close_path(dev, o.extract_path(**kw), what)
    g.close()


@pytest.mark.parametrize("algo,lvl,heur", [("FD", 1, False), ("FD", 1, True), ("SG", 2, False), ("DFM", 1, False)])
def test_driver_loop_paths(algo, lvl, heur):
    """The reference driver's loop (Tests/Planners/FDSTAR/main.cpp:90-167): patch, step, extract
    (max_steps 20), advance the start along the extracted path until more than 5 cells away."""
    width = length = 160
    seed = 11
    cost0 = ufm_amd.synth.cost_map(seed, width, length)
    start, goal = ufm_amd.synth.start_goal(width, length)
    o, g = make_pair(ALGOS[algo], lvl, cost0, start, goal, heuristic=heur, hm=1.0)
    cost = cost0.copy()
    script = list(ufm_amd.synth.replan_script(seed, width, length, n_patches=30))
    nxt = start
    moves = 0
    for k, _s, top, left, patch in script:
        # the patch follows the robot, as the simulator's field of view does
        top = int(min(max(round(nxt[0]) - patch.shape[0] // 2, 0), length - patch.shape[0]))
        left = int(min(max(round(nxt[1]) - patch.shape[1] // 2, 0), width - patch.shape[1]))
        cost[top:top + patch.shape[0], left:left + patch.shape[1]] = patch
        for p in (o, g):
            p.patch_map(patch, top, left)
            p.set_start(*nxt)
        assert o.step() == 0 and g.step() == 0
        kw = dict(max_steps=20, lookahead=True, allow_indirect=INDIRECT[algo])
        dev = g.extract_path(**kw)
        what = "%s-%d replan %d from %r" % (algo, lvl, k, nxt)
        same_path(dev, oracle_on_device_field(g, algo, cost, 255, nxt, goal, **kw), what + " [device field]")
        ref = o.extract_path(**kw)
        if algo == "DFM":
            close_path_while_final(dev, ref, o, what)
        else:
            close_path(dev, ref, what)
        pts = dev[0]
        assert len(pts) >= 2, what
        # main.cpp:157-166 (Cell(Position) rounds, Cell.cpp:20-21)
        prev = nxt
        for i in range(1, len(pts)):
            nxt = (float(pts[i][0]), float(pts[i][1]))
            d = np.hypot(orc._roundf(nxt[0]) - orc._roundf(prev[0]), orc._roundf(nxt[1]) - orc._roundf(prev[1]))
            if d > 5:
                break
        moves += 1
        if nxt == goal:
            break
    assert moves >= 10
    g.close()


def test_path_capacity_and_counts(ref_bitmaps):
    """way points / costs beyond the caller's capacity are counted, not stored"""
    import ctypes as C
    cost, sg = ref_bitmaps["noise-trap"]
    o, g = make_pair(ALGOS["FD"], 0, cost, sg[:2], sg[2:])
    assert g.step() == 0
    full = g.extract_path(max_steps=800)
    pts = np.full((10, 2), -1, np.float32)
    costs = np.full(4, -1, np.float32)
    info = ufm_amd.capi.PathInfo()
    rc = g.L.ufm_extract_path(g.h, 800, 1, 1, pts.ctypes.data, 10, costs.ctypes.data, 4, C.byref(info))
    assert rc == 0
    assert info.n_points == len(full[0]) and info.n_costs == len(full[1])
    assert np.array_equal(pts, full[0][:10]) and np.array_equal(costs, full[1][:4])
    assert info.total_cost == full[2] and info.total_dist == full[3]
    # bad arguments
    assert g.L.ufm_extract_path(g.h, 0, 1, 1, pts.ctypes.data, 10, costs.ctypes.data, 4, C.byref(info)) == -22
    assert g.L.ufm_extract_path(g.h, 20, 1, 1, None, 10, costs.ctypes.data, 4, C.byref(info)) == -22
    g.close()


def test_path_unreachable_goal():
    """goal walled in: every traversal is empty, the reference's extractor stays at the start for
    max_steps moves (value-initialised additions, cost_to_goal 0) -- same here"""
    width = length = 64
    cost = np.full((length, width), 10, np.uint8)
    cost[40:50, 38] = 255
    cost[40:50, 50] = 255
    cost[40, 38:51] = 255
    cost[49, 38:51] = 255
    o, g = make_pair(ALGOS["FD"], 0, cost, (5.0, 5.0), (45.0, 45.0), thr=1.0)
    assert o.step() == 0 and g.step() == 0
    dev = g.extract_path(max_steps=20)
    ref = o.extract_path(max_steps=20)
    assert len(dev[0]) == len(ref[0]) == 1
    assert dev[2] == ref[2] == 0.0 and dev[3] == ref[3] == 0.0
    assert g.path_info.steps == 20
    g.close()


@pytest.mark.parametrize("algo", ["FD", "DFM"])
def test_batch_paths(algo):
    """all maps of a batch in one launch == one planner per map"""
    n, width, length = 4, 128, 128
    b = ufm_amd.BatchPlanner(n, ALGOS[algo], 1)
    b.set_occupancy_threshold(1.0)
    singles = []
    for m in range(n):
        cost = ufm_amd.synth.cost_map(1000 + m, width, length)
        start, goal = ufm_amd.synth.start_goal(width, length)
        start = (start[0] + 3 * m, start[1] + m)
        b.set_map(m, cost)
        b.set_start(m, *start)
        b.set_goal(m, *goal)
        g = ufm_amd.Planner(ALGOS[algo], 1)
        g.set_occupancy_threshold(1.0)
        g.set_map(cost)
        g.set_start(*start)
        g.set_goal(*goal)
        assert g.step() == 0
        singles.append(g)
    assert b.step() == 0
    paths = b.extract_paths(max_steps=400, allow_indirect=INDIRECT[algo])
    for m in range(n):
        one = singles[m].extract_path(max_steps=400, allow_indirect=INDIRECT[algo])
        if algo == "FD":
            same_path(paths[m], one, "batch map %d" % m)
        else:
            close_path(paths[m], one, "batch map %d" % m)
        if algo == "FD":   # (DFM + this extractor oscillates short of the goal on smooth maps)
            assert tuple(paths[m][0][-1]) == tuple(float(v) for v in ufm_amd.synth.start_goal(width, length)[1])
        singles[m].close()
    b.close()
