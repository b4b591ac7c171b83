"""The one place where parity is PINNED to the reference's own output: the mission logs it holds under Tests/Results/.

Tests/Results/noise-trap/planner_opt0.log is the console output of the reference's Field D* planner process (level 0, heuristic keys)
driven through a whole mission by Simulator/simulator/run_simulator.py on Tests/Tests/noise-trap_90_90_25_25_.bmp: 134 closed-loop steps --
the simulator reveals a disc of radius 15 around the robot in a blurred, penalised copy of the bitmap and sends the bounding patch and the
map's smallest cost (the heuristic multiplier); the planner replans, extracts its path and moves to the path's next way point -- and for
every step the planner printed its position, the patch rectangle, "nodes updated", "nodes expanded", and the cost and length of the
extracted path.  tests/golden/ref_missions.npz holds those numbers and the bitmap's pixels (make_mission_fixture.py).

The simulator's side is regenerated without cv2 (ufm_amd harness: OpenCV's fixed-point Gaussian 13 x 13, cost = ~pixel, penalty 15, disc
15, C-space 1); the planner's side is the oracle (CPU test) or the engine through the C ABI (GPU test).  Nothing of the log is fed
back: every position is the replay's own (path point 1 of its own extraction), so 134 chained replans, extractions and moves have to
agree with the reference to the last printed digit -- position (2 numbers; the simulator prints them with six decimals, i.e. eight
significant digits), path cost and path length of every step.

What the log does NOT pin, said plainly: it comes from an older revision of the reference (it prints lines the current sources have
commented out, FieldDPlanner_impl.h:65,139) -- its "nodes expanded" agrees with the oracle's num_nodes_expanded in the first plan (8760)
and in 84 of the 105 replans that print it, its "nodes updated" in all but 9 steps (2 fewer there: the nine steps whose patch reaches the map's bottom border -- that
revision evidently did not count two nodes on the border row): counts of queue operations, which depend on the revision, not on the field.  The second log (wall-b, a binary bitmap: free cells all cost 1, so paths tie) agrees in its
first plan (2258 nodes, cost 1203.34, length 89.0422) and parts one step later, in the sixth digit of a path length.  One planner
(FD level 0 with heuristic keys), its extractor and the simulator's map preparation are pinned this way; SG / MS-DFM, the other levels and
the keys without heuristic remain cross-checked only."""
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


def check_mission(name, planner, read_counts, count_expanded, counts=True):
    _, _, _, steps = load(name)
    n = upd_same = exp_same = exp_n = 0
    for k, st, got in replay(name, planner, read_counts):
        what = "%s step %d" % (name, k)
        assert got["pos"] == st["pos"], (what, got, st)                    # where the planner stands: the previous step's own path point
        assert got["sim_pos"] == st["sim_pos"], (what, got, st)            # ... as the simulator printed it: six decimals
        assert got["patch"] == st["sim_patch"] == st.get("patch", st["sim_patch"]), (what, got, st)   # the simulator's rectangle for it
        assert (got["cost"], got["dist"]) == (st["cost"], st["dist"]), (what, got, st)    # cost and length of the extracted path
        if counts and "updated" in st:
            assert abs(got["updated"] - st["updated"]) <= 2, (what, got, st)
            upd_same += got["updated"] == st["updated"]
        if count_expanded and "expanded" in st:
            exp_n += 1
            exp_same += got["expanded"] == st["expanded"]
            if k == 0:
                assert got["expanded"] == st["expanded"], (what, got, st)  # the first plan's expansions: 8760
        n += 1
    assert n == len(steps) == 134, (n, len(steps))                          # the replay reaches the goal in the step the reference did
    assert not counts or upd_same >= 124, upd_same
    if count_expanded:
        assert exp_same >= 80, (exp_same, exp_n)
    return n, upd_same, exp_same, exp_n


def test_oracle_replays_the_reference_mission_log():
    import oracle_py as orc
    o = orc.OraclePlanner(ufm_amd.ALGO_FD, 0, True)
    n, upd_same, exp_same, exp_n = check_mission("noise-trap", o, lambda p: (p.num_updated, p.num_expanded), True)
    assert (n, upd_same) == (134, 124) and (exp_same, exp_n) == (84, 105)     # exactly what the module text says (133 steps print "nodes updated")


def test_oracle_first_plan_of_the_second_log():
    import oracle_py as orc
    o = orc.OraclePlanner(ufm_amd.ALGO_FD, 0, True)
    k, st, got = next(replay("wall-b", o, lambda p: (p.num_updated, p.num_expanded), 1))
    assert (got["pos"], got["patch"], got["expanded"], got["cost"], got["dist"]) == (st["pos"], st["sim_patch"], st["expanded"], st["cost"], st["dist"])
    assert (st["expanded"], st["cost"], st["dist"]) == (2258, "1203.34", "89.0422")


@pytest.mark.gpu
def test_engine_replays_the_reference_mission_log():
    """the product -- field on the GPU, path extraction on the GPU, through the C ABI -- in the reference's closed loop: 134 steps, every
    printed digit of position, path cost and path length; "nodes updated" as the oracle has it.  ("nodes expanded" is not compared: the
    engine counts elements whose value changed, not queue pops.)"""
    g = ufm_amd.Planner(ufm_amd.ALGO_FD, 0, True)
    n, upd_same, _, _ = check_mission("noise-trap", g, lambda p: (p.num_nodes_updated, p.num_nodes_expanded), False)
    assert (n, upd_same) == (134, 124)
    g.close()


@pytest.mark.gpu
def test_engine_first_plan_of_the_second_log():
    g = ufm_amd.Planner(ufm_amd.ALGO_FD, 0, True)
    k, st, got = next(replay("wall-b", g, lambda p: (p.num_nodes_updated, p.num_nodes_expanded), 1))
    assert (got["pos"], got["patch"], got["cost"], got["dist"]) == (st["pos"], st["sim_patch"], st["cost"], st["dist"])
    g.close()


# The same mission through the planner's other forms.  The log was written by level 0 with heuristic keys; level 1 (the back-pointer variant)
# and the builds without heuristic keys are the reference's own alternatives for the same search -- same field below the start's key, hence the
# same paths -- so they have to reproduce the log's positions, path costs and path lengths too (their queue-operation counts are their own).
OTHER_FORMS = [(1, True), (0, False), (1, False)]


@pytest.mark.parametrize("lvl,heur", OTHER_FORMS)
def test_oracle_other_forms_replay_the_reference_mission_log(lvl, heur):
    import oracle_py as orc
    o = orc.OraclePlanner(ufm_amd.ALGO_FD, lvl, heur)
    n, _, _, _ = check_mission("noise-trap", o, lambda p: (p.num_updated, p.num_expanded), False, counts=False)
    assert n == 134


@pytest.mark.gpu
@pytest.mark.parametrize("lvl,heur", OTHER_FORMS)
def test_engine_other_forms_replay_the_reference_mission_log(lvl, heur):
    g = ufm_amd.Planner(ufm_amd.ALGO_FD, lvl, heur)
    n, _, _, _ = check_mission("noise-trap", g, lambda p: (p.num_nodes_updated, p.num_nodes_expanded), False, counts=False)
    assert n == 134
    g.close()
