"""The place where parity is PINNED to the reference's own output: the two mission logs it holds under Tests/Results/.

Tests/Results/{noise-trap,wall-b}/planner_opt0.log are the console output of the reference's Field D* planner process (level 0, heuristic
keys) driven through whole missions by Simulator/simulator/run_simulator.py on Tests/Tests/noise-trap_90_90_25_25_.bmp (134 closed-loop
steps) and Tests/Tests/wall-b_27_10_2_10_.bmp (89 steps): the simulator reveals a disc of radius 15 around the robot in a blurred,
penalised copy of the bitmap and sends the bounding patch and the map's smallest cost (the heuristic multiplier); the planner replans,
extracts its path and moves to the path's next way point -- and for every step the planner printed its position, the patch rectangle,
"nodes updated", "nodes expanded", and the cost and length of the extracted path.  tests/golden/ref_missions.npz holds those numbers and
the bitmaps' pixels (make_mission_fixture.py).

The simulator's side is regenerated without cv2 (ufm_amd harness: OpenCV's fixed-point Gaussian 13 x 13, cost = ~pixel, penalty 15, disc
15, C-space 1); the planner's side is the oracle (CPU tests) or the engine through the C ABI (GPU tests).  Nothing of a log is fed back:
every position is the replay's own (path point 1 of its own extraction), so 134 / 89 chained replans, extractions and moves have to agree
with the reference to the last printed digit -- position (the simulator prints it with six decimals: eight significant digits), path cost
and path length of every step.

ROUND 4: BOTH logs are reproduced in full, counts included, once the restatement follows the REVISION that wrote them (the logs print lines
the current sources have commented out, FieldDPlanner_impl.h:65,139).  Two differences from the current sources were identified by search
against the logs (tools/mission_revision_probe.py), neither in an update operator:
  (1) start_cell_ was the cell that CONTAINS the start position (floor), where Cell(const Position&) now rounds (Cell.cpp:20-21): the four
      start nodes of end_condition() differ whenever a coordinate's fraction is >= 0.5.  "nodes expanded" then equals the log's in 104 of 105
      (noise-trap) and 73 of 73 (wall-b) replans -- with roundf: 84 and 8 -- and wall-b, a binary bitmap whose free cells all cost 1 (paths
      tie, so which nodes beyond the start's key hold stale values decides between equal-cost way points), is replayed to its last step
      instead of parting in step 1;
  (2) update() left out the corner nodes on the map's far borders (x == length, y == width): "nodes updated" is 2 smaller in exactly the
      steps whose changed cells touch the bottom row / right column; with it 133 of 133 and 88 of 88 agree.
ORC_REV_LOG (oracle) / ufm_set_param("start_cell_floor", 1) (engine) select that revision; the defaults follow the current sources.

WHAT THE LOGS PIN, measured by ablation (test_which_branches_the_logs_pin): the machinery all planners share -- Graph, keys with heuristic,
re-keying on a moved start, end_condition, update, the queue's order up to ties (177 of 178 expansion counts), the extractor with lookahead
and indirect traversals, the simulator restatement -- and, of compute_optimal_cost (FD impl:292-319): cases B, II, A and **Type I**
(wall-b steps 29-31: without Type I, or with the shifted-grid operator, path cost 769.699 instead of the log's 769.831).  NOT pinned:
FD's Type III in both its forms (`f <= 0` and the `f^2 <= CATH(c,b)` clause): they are evaluated (370 k times) and decide 14 k RHS values
in the noise-trap mission, yet the logs come out the same to the last digit without them -- and, on noise-trap alone, with the whole
shifted-grid operator in FD's place.  MS-DFM: no log, unpinned."""
import json
import os

import numpy as np
import pytest

import ufm_amd
from ufm_amd_pkg import harness

HERE = os.path.dirname(os.path.abspath(__file__))


def g6(v):
    """how the reference's std::cout prints a float: six significant digits"""
    return "%g" % np.float32(v)


def load(name):
    z = np.load(os.path.join(HERE, "golden", "ref_missions.npz"))
    steps = json.loads(bytes(z[name + "_steps"]).decode())
    sg = z[name + "_startgoal"]
    return z[name + "_pixels"], (float(sg[0]), float(sg[1])), (float(sg[2]), float(sg[3])), steps


def replay(name, planner, read_counts, n_steps=None):
    """the mission of run_simulator.py:123-230 against `planner` (oracle or engine: same surface); yields per step what the planner process printed"""
    pixels, start, goal, steps = load(name)
    data_l, data_h = harness.simulation_data(pixels, low_res_penalty=15, filter_size=13)     # run_simulator.py:147
    planner.reset()
    planner.set_occupancy_threshold(1.0)
    planner.set_heuristic_multiplier(float(int(data_l.min())))                                # :151-152: the hint is sent as an int
    planner.set_map(data_l)
    planner.set_start(*start)
    planner.set_goal(*goal)
    pos = start
    for k, st in enumerate(steps[:n_steps]):
        center = (int(round(pos[1])), int(round(pos[0])))                                     # :170
        data_l, (top, left), rng = harness.round_patch_update(data_l, data_h, center, 15)     # :171
        patch = np.ascontiguousarray(data_l[rng[0], rng[1]])
        planner.patch_map(patch, top, left)
        planner.set_heuristic_multiplier(float(int(data_l.min())))
        planner.set_start(*pos)
        assert planner.step() == 0
        pts, costs, total_cost, total_dist = planner.extract_path(max_steps=4000, lookahead=True, allow_indirect=True)
        updated, expanded = read_counts(planner)
        yield k, st, {"pos": [g6(pos[0]), g6(pos[1])], "sim_pos": ["%f" % np.float32(pos[0]), "%f" % np.float32(pos[1])], "patch": [top, left, patch.shape[1], patch.shape[0]], "updated": updated, "expanded": expanded,
                      "cost": g6(total_cost), "dist": g6(total_dist)}
        if len(pts) < 2:
            return
        pos = (float(pts[1][0]), float(pts[1][1]))
        if pos == goal:
            return


def check_mission(name, planner, read_counts, count_expanded, counts=True, exact_counts=False):
    """all steps of a log against `planner`: position, patch rectangle, path cost, path length to the printed digit; counts as asked for.
    Returns (steps, steps whose "nodes updated" is the log's, whose "nodes expanded" is, steps that print "nodes expanded")."""
    _, _, _, steps = load(name)
    n = upd_same = exp_same = exp_n = 0
    for k, st, got in replay(name, planner, read_counts):
        what = "%s step %d" % (name, k)
        assert got["pos"] == st["pos"], (what, got, st)                    # where the planner stands: the previous step's own path point
        assert got["sim_pos"] == st["sim_pos"], (what, got, st)            # ... as the simulator printed it: six decimals
        assert got["patch"] == st["sim_patch"] == st.get("patch", st["sim_patch"]), (what, got, st)   # the simulator's rectangle for it
        assert (got["cost"], got["dist"]) == (st["cost"], st["dist"]), (what, got, st)    # cost and length of the extracted path
        if counts and "updated" in st:
            assert abs(got["updated"] - st["updated"]) <= (0 if exact_counts else 2), (what, got, st)
            upd_same += got["updated"] == st["updated"]
        if count_expanded and "expanded" in st:
            exp_n += 1
            exp_same += got["expanded"] == st["expanded"]
            if k == 0:
                assert got["expanded"] == st["expanded"], (what, got, st)  # the first plan's expansions: 8760 / 2258
            if exact_counts:
                assert abs(got["expanded"] - st["expanded"]) <= 1, (what, got, st)
        n += 1
    assert n == len(steps) == STEPS[name], (n, len(steps))                   # the replay reaches the goal in the step the reference did
    return n, upd_same, exp_same, exp_n


STEPS = {"noise-trap": 134, "wall-b": 89}
o_counts = lambda p: (p.num_updated, p.num_expanded)
g_counts = lambda p: (p.num_nodes_updated, p.num_nodes_expanded)


def first_mismatch(name, planner, read_counts):
    """the steps at which a replay's printed path cost / length / position differ from the log's (the replay is closed-loop: after the first one it is on its own)"""
    bad = []
    for k, st, got in replay(name, planner, read_counts):
        if (got["pos"], got["cost"], got["dist"]) != (st["pos"], st["cost"], st["dist"]):
            bad.append(k)
    return bad


# ---- the oracle, in the logs' revision: everything the logs print -----------------------------------------------------------------
@pytest.mark.parametrize("name,exp_expected", [("noise-trap", (104, 105)), ("wall-b", (73, 73))])
def test_oracle_replays_the_reference_mission_logs_counts_included(name, exp_expected):
    import oracle_py as orc
    o = orc.OraclePlanner(ufm_amd.ALGO_FD, 0, True, revision=orc.REV_LOG)
    n, upd_same, exp_same, exp_n = check_mission(name, o, o_counts, True, exact_counts=True)
    assert upd_same == n - 1                                    # every step but the first plan prints "nodes updated": all equal
    assert (exp_same, exp_n) == exp_expected                    # "nodes expanded": all but one step of one log (noise-trap step 119: 272 against 273)


def test_oracle_revision_differences_one_at_a_time():
    """what each of the two identified differences explains (the module text): the start cell the expansion counts and wall-b's paths, the
    far-border rule the "nodes updated" of the steps at the bottom / right border"""
    import oracle_py as orc
    res = {}
    for rev in (orc.REV_CURRENT, orc.REV_START_CELL_FLOOR, orc.REV_UPDATE_SKIPS_FAR_BORDER):
        o = orc.OraclePlanner(ufm_amd.ALGO_FD, 0, True, revision=rev)
        n, upd_same, exp_same, exp_n = check_mission("noise-trap", o, o_counts, True)     # (the paths hold in every revision on this map)
        res[rev] = (upd_same, exp_same)
    assert res[orc.REV_CURRENT] == (124, 84)
    assert res[orc.REV_START_CELL_FLOOR] == (124, 104)
    assert res[orc.REV_UPDATE_SKIPS_FAR_BORDER] == (133, 84)
    # wall-b: with the current sources' start cell the closed loop parts from the log in its second step (same path cost, another way point)
    o = orc.OraclePlanner(ufm_amd.ALGO_FD, 0, True, revision=orc.REV_UPDATE_SKIPS_FAR_BORDER)
    assert first_mismatch("wall-b", o, o_counts)[0] == 1


# ---- the oracle as the current sources stand (what the engine is held to everywhere else) ---------------------------------------------
def test_oracle_replays_the_reference_mission_log():
    import oracle_py as orc
    o = orc.OraclePlanner(ufm_amd.ALGO_FD, 0, True)
    n, upd_same, exp_same, exp_n = check_mission("noise-trap", o, o_counts, True)
    assert (n, upd_same) == (134, 124) and (exp_same, exp_n) == (84, 105)


def test_oracle_first_plan_of_the_second_log():
    import oracle_py as orc
    o = orc.OraclePlanner(ufm_amd.ALGO_FD, 0, True)
    k, st, got = next(replay("wall-b", o, o_counts, 1))
    assert (got["pos"], got["patch"], got["expanded"], got["cost"], got["dist"]) == (st["pos"], st["sim_patch"], st["expanded"], st["cost"], st["dist"])
    assert (st["expanded"], st["cost"], st["dist"]) == (2258, "1203.34", "89.0422")


# ---- what the logs pin and what they do not -----------------------------------------------------------------------------------------
def test_which_branches_the_logs_pin():
    """Ablation: FD's compute_optimal_cost without one of its branches (oracle test hook) against both logs.  A branch whose removal changes a
    printed digit is pinned; one whose removal changes nothing is not, however often it is taken."""
    import oracle_py as orc
    L = orc.lib()
    try:
        outcome = {}
        for mask in (1, 2, 4, 8):            # no f^2 <= CATH clause / no Type I / no c > b chain at all / Type III pays c instead of b
            L.orc_set_fd_ablation(mask)
            outcome[mask] = tuple(first_mismatch(nm, orc.OraclePlanner(ufm_amd.ALGO_FD, 0, True, revision=orc.REV_LOG), o_counts) for nm in ("noise-trap", "wall-b"))
    finally:
        L.orc_set_fd_ablation(0)
    assert outcome[2] == ([], [29, 30, 31])      # Type I: pinned, by three steps of wall-b
    assert outcome[4] == ([], [29, 30, 31])
    assert outcome[1] == ([], [])                # Type III, either form: NOT pinned
    assert outcome[8] == ([], [])
    # ... although the noise-trap mission takes every branch, and every one of them decides RHS values
    orc.case_counts_reset()
    check_mission("noise-trap", orc.OraclePlanner(ufm_amd.ALGO_FD, 0, True, revision=orc.REV_LOG), o_counts, False, counts=False)
    ev, won = orc.case_counts()
    for c in orc.CASES[:8]:
        assert ev[c] > 500 and won[c] > 50, (c, ev[c], won[c])
    assert won["FD III (f <= 0)"] + won["FD III (f^2 <= CATH(c,b) [sic])"] > 10000


@pytest.mark.parametrize("lvl", [0, 2])
def test_shifted_grid_planner_against_the_logs(lvl):
    """The first log cannot tell Field D* from the shifted-grid planner (SG reproduces all 134 steps: what that log pins is what the two
    share); the second can: SG parts from it where FD's Type I decides (steps 29-31)."""
    import oracle_py as orc
    n, _, _, _ = check_mission("noise-trap", orc.OraclePlanner(ufm_amd.ALGO_SG, lvl, True, revision=orc.REV_LOG), o_counts, False, counts=False)
    assert n == 134
    assert first_mismatch("wall-b", orc.OraclePlanner(ufm_amd.ALGO_SG, lvl, True, revision=orc.REV_LOG), o_counts) == [29, 30, 31]


# ---- the engine ------------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_engine_replays_the_reference_mission_log():
    """the product -- field on the GPU, path extraction on the GPU, through the C ABI -- in the reference's closed loop: 134 steps, every
    printed digit of position, path cost and path length; "nodes updated" as the oracle has it.  ("nodes expanded" is not compared: the
    engine counts elements whose value changed, not queue pops.)"""
    g = ufm_amd.Planner(ufm_amd.ALGO_FD, 0, True)
    n, upd_same, _, _ = check_mission("noise-trap", g, g_counts, False)
    assert (n, upd_same) == (134, 124)
    g.close()


@pytest.mark.gpu
def test_engine_first_plan_of_the_second_log():
    g = ufm_amd.Planner(ufm_amd.ALGO_FD, 0, True)
    k, st, got = next(replay("wall-b", g, g_counts, 1))
    assert (got["pos"], got["patch"], got["cost"], got["dist"]) == (st["pos"], st["sim_patch"], st["cost"], st["dist"])
    g.close()


def open_loop(name, planner):
    """a log OPEN-LOOP: every step starts from the position the log printed (the simulator's six decimals); yields (k, log step, path cost, path length)"""
    pixels, start, goal, steps = load(name)
    data_l, data_h = harness.simulation_data(pixels, low_res_penalty=15, filter_size=13)
    planner.reset()
    planner.set_occupancy_threshold(1.0)
    planner.set_heuristic_multiplier(float(int(data_l.min())))
    planner.set_map(data_l)
    planner.set_start(*start)
    planner.set_goal(*goal)
    for k, st in enumerate(steps):
        pos = (float(np.float32(st["sim_pos"][0])), float(np.float32(st["sim_pos"][1])))
        center = (int(round(pos[1])), int(round(pos[0])))
        data_l, (top, left), rng = harness.round_patch_update(data_l, data_h, center, 15)
        planner.patch_map(np.ascontiguousarray(data_l[rng[0], rng[1]]), top, left)
        planner.set_heuristic_multiplier(float(int(data_l.min())))
        planner.set_start(*pos)
        assert planner.step() == 0
        pts, costs, total_cost, total_dist = planner.extract_path(max_steps=4000, lookahead=True, allow_indirect=True)
        yield k, st, total_cost, total_dist


@pytest.mark.gpu
def test_engine_replays_the_first_log_with_its_start_cell():
    """the engine with the logs' revision of the start cell (ufm_set_param "start_cell_floor"): the closed loop of the first log, all 134 steps"""
    g = ufm_amd.Planner(ufm_amd.ALGO_FD, 0, True)
    g.set_param("start_cell_floor", 1)
    n, _, _, _ = check_mission("noise-trap", g, g_counts, False)
    assert n == 134
    g.close()


@pytest.mark.gpu
def test_engine_second_log_open_loop():
    """wall-b through the engine.  Closed-loop the engine leaves this log in step 1 -- same path cost, another of several equal-cost way points:
    on this binary bitmap paths tie, and which way point the reference takes depends on the stale values its queue order leaves on the nodes
    beyond the start's key, while the engine finalises everything below the key plus one move (a superset).  So the log is fed OPEN-LOOP: from
    each of the 89 positions the log printed, the engine's replan + extraction must give the log's path cost (1e-4 relative; measured:
    the printed digit in 87 steps, 4.8e-6 at worst) and its path length within 1e-4 (measured 1.2e-5: flat minima move way points, not costs)."""
    g = ufm_amd.Planner(ufm_amd.ALGO_FD, 0, True)
    g.set_param("start_cell_floor", 1)
    n = same = 0
    worst_c = worst_d = 0.0
    for k, st, tc, td in open_loop("wall-b", g):
        worst_c = max(worst_c, abs(tc - float(st["cost"])) / float(st["cost"]))
        worst_d = max(worst_d, abs(td - float(st["dist"])) / float(st["dist"]))
        same += g6(tc) == st["cost"]
        n += 1
    print("wall-b open-loop: %d steps, path cost to the printed digit in %d, worst relative difference cost %.2e length %.2e" % (n, same, worst_c, worst_d))
    assert n == 89 and same >= 80 and worst_c <= 1e-5 and worst_d <= 1e-4
    g.close()


# The same missions through the planner's other forms.  Level 1 (the back-pointer variant) is the reference's own alternative for the same
# search: same field below the start's key.  On noise-trap the builds without heuristic keys give the same paths too; on wall-b they do not
# have to (another key order leaves other elements beyond the start's key with stale values, and there ties decide).
OTHER_FORMS = [(1, True), (0, False), (1, False)]


@pytest.mark.parametrize("lvl,heur", OTHER_FORMS)
def test_oracle_other_forms_replay_the_reference_mission_log(lvl, heur):
    import oracle_py as orc
    o = orc.OraclePlanner(ufm_amd.ALGO_FD, lvl, heur)
    n, _, _, _ = check_mission("noise-trap", o, o_counts, False, counts=False)
    assert n == 134


def test_oracle_level_1_replays_the_second_log_too():
    import oracle_py as orc
    n, _, _, _ = check_mission("wall-b", orc.OraclePlanner(ufm_amd.ALGO_FD, 1, True, revision=orc.REV_LOG), o_counts, False, counts=False)
    assert n == 89


@pytest.mark.gpu
@pytest.mark.parametrize("lvl,heur", OTHER_FORMS)
def test_engine_other_forms_replay_the_reference_mission_log(lvl, heur):
    g = ufm_amd.Planner(ufm_amd.ALGO_FD, lvl, heur)
    n, _, _, _ = check_mission("noise-trap", g, g_counts, False, counts=False)
    assert n == 134
    g.close()


@pytest.mark.gpu
@pytest.mark.parametrize("lvl", [0, 2])
def test_engine_shifted_grid_planner_on_the_first_log(lvl):
    """the engine's SG operator through the same closed loop (see test_shifted_grid_planner_against_the_logs)"""
    g = ufm_amd.Planner(ufm_amd.ALGO_SG, lvl, True)
    n, _, _, _ = check_mission("noise-trap", g, g_counts, False, counts=False)
    assert n == 134
    g.close()
