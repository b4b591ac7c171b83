"""The planner process (apps/ufm_planner.cpp) driven over two FIFOs with the reference's wire
protocol (Simulator/simulator/run_simulator.py:38-103, Tests/run_test.py:85-177).
This test plays the simulator's side with numpy only and keeps the CPU oracle in lockstep."""
import os
import struct
import subprocess

import numpy as np
import pytest

import oracle_py as orc
import ufm_amd
from helpers import ALGOS
from test_gpu_path import INDIRECT, close_path, close_path_while_final

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "unige-tasi-path-planners_amd")


def _exe(heur):
    exe = os.path.join(PKG, "ufm_planner" if heur else "ufm_planner_no_heur")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", PKG, "apps"])
    return exe


def test_planner_process_builds_and_prints_usage():
    for heur in (True, False):
        r = subprocess.run([_exe(heur)], capture_output=True, text=True)
        assert r.returncode == 1 and "fifo_in" in r.stderr


class Sim:
    """simulator end of the pipes (run_simulator.py:38-103)"""

    def __init__(self, to_planner, from_planner, alive):
        pipes = ufm_amd.harness.Pipes(to_planner, from_planner, alive=alive)   # does not hang if the planner died
        self.o, self.i = pipes.o, pipes.i

    def send(self, fmt, *v):
        self.o.write(struct.pack("<" + fmt, *v))

    def recv(self, fmt):
        n = struct.calcsize("<" + fmt)
        b = self.i.read(n)
        assert len(b) == n, "planner closed the pipe"
        return struct.unpack("<" + fmt, b)

    def close(self):
        self.o.close()
        self.i.close()


@pytest.mark.gpu
@pytest.mark.parametrize("algo,lvl,heur,tof,argstyle", [
    ("FD", 1, False, 1, "short"), ("SG", 2, False, 0, "long"), ("DFM", 1, True, 0, "short"), ("FD", 1, True, 0, "name"),
])
def test_simulator_drives_the_planner_process(tmp_path, algo, lvl, heur, tof, argstyle):
    width, length, seed, moves = 144, 128, 21, 12
    cost = ufm_amd.synth.cost_map(seed, width, length)
    start, goal = (8.0, 8.0), (float(length - 8), float(width - 8))
    script = list(ufm_amd.synth.replan_script(seed, width, length, n_patches=moves))
    a, b = str(tmp_path / "pipe_1"), str(tmp_path / "pipe_2")
    os.mkfifo(a)
    os.mkfifo(b)
    exe = _exe(heur)
    if argstyle == "short":        # DFM/main.cpp: <fifo_in> <fifo_out>, start / goal in-band
        cmd = [exe, "--planner", algo, "--level", str(lvl), "--max-moves", str(moves), a, b]
    elif argstyle == "long":       # FDSTAR/main.cpp:16-31
        cmd = [exe, "--planner", algo, "--level", str(lvl), "--max-moves", str(moves), "map.bmp",
               str(start[0]), str(start[1]), str(goal[0]), str(goal[1]), "1", a, b, "0", str(tof), "out"]
    else:                          # planner and level from a reference-style binary name
        link = str(tmp_path / ("field_d_planner_%d" % lvl))
        os.symlink(exe, link)
        cmd = [link, "--max-moves", str(moves), a, b]
    proc = subprocess.Popen(cmd, stdout=subprocess.DEVNULL)
    try:
        sim = Sim(a, b, alive=lambda: proc.poll() is None)
        assert sim.recv("b") == (0,)
        sim.send("b", 0)
        sim.send("ii", width, length)
        sim.o.write(cost.tobytes())
        if argstyle != "long":
            sim.send("ffffB", start[0], start[1], goal[0], goal[1], tof)
        hm = int(cost.min())
        sim.send("i", hm)
        sim.o.flush()

        o = orc.OraclePlanner(ALGOS[algo], lvl, heur)
        o.reset()
        o.set_occupancy_threshold(1)
        o.set_heuristic_multiplier(hm)
        o.set_map(cost)
        o.set_start(*start)
        o.set_goal(*goal)
        shift = 0.5 if algo == "DFM" else 0.0
        n_moves = 0
        here = start
        while True:
            (code,) = sim.recv("b")
            if code == 2:
                break
            assert code == 1
            x, y, step_cost = sim.recv("fff")
            assert (x - shift, y - shift) == here
            k, _s, _t, _l, patch = script[n_moves]
            ph, pw = patch.shape
            top = int(min(max(round(here[0]) - ph // 2, 0), length - ph))
            left = int(min(max(round(here[1]) - pw // 2, 0), width - pw))
            sim.send("b", 1)
            sim.send("iiii", top, left, ph, pw)
            sim.o.write(patch.tobytes())
            sim.send("i", hm)
            sim.o.flush()
            o.patch_map(patch, top, left)
            o.set_start(*here)
            o.set_heuristic_multiplier(hm)
            assert o.step() == 0
            ref = o.extract_path(max_steps=20, allow_indirect=INDIRECT[algo])

            assert sim.recv("b") == (3,)
            (n,) = sim.recv("i")
            pts = np.array(sim.recv("%df" % (2 * n)), np.float32).reshape(n, 2)
            # the reference's reader assumes one cost per segment (run_simulator.py:82-83)
            assert len(ref[1]) == len(ref[0]) - 1
            costs = np.array(sim.recv("%df" % (n - 1)), np.float32)
            dist, total = sim.recv("ff")
            u_ms, p_ms, e_ms = sim.recv("fff")
            assert u_ms >= 0 and p_ms > 0 and e_ms > 0
            got = (pts, costs, total, dist)
            what = "%s-%d move %d from %r" % (algo, lvl, n_moves, here)
            if algo == "DFM":
                close_path_while_final(got, ref, o, what)
            else:
                close_path(got, ref, what)
                assert np.allclose(costs, ref[1], rtol=1e-5)
            if tof:
                assert sim.recv("b") == (4,)
                (count,) = sim.recv("q")
                rec = np.frombuffer(sim.i.read(16 * count), dtype=[("x", "<i4"), ("y", "<i4"), ("g", "<f4"), ("rhs", "<f4")])
                assert len(rec) == count
                field = np.full(o.g().shape, np.inf, np.float32)
                field[rec["x"], rec["y"]] = rec["g"]
                m = o.trusted_mask(below_start_key=True)
                assert np.array_equal(field[m], o.g()[m]), what + ": expanded-element dump differs from the oracle"
                assert np.array_equal(rec["g"], rec["rhs"])
            # FDSTAR/main.cpp:157-166
            nxt = here
            for i in range(1, n):
                nxt = (float(pts[i][0]), float(pts[i][1]))
                if np.hypot(orc._roundf(nxt[0]) - orc._roundf(here[0]), orc._roundf(nxt[1]) - orc._roundf(here[1])) > 5:
                    break
            here = nxt
            n_moves += 1
            if here == goal:
                break
        sim.send("b", 2)
        sim.o.flush()
        if here == goal:
            assert sim.recv("b") == (2,)
        assert proc.wait(timeout=60) == 0
        assert n_moves == moves
        sim.close()
    finally:
        if proc.poll() is None:
            proc.kill()
