"""Generates tests/golden/ref_missions.npz from the two mission logs the reference holds under Tests/Results/ and the
two bitmaps they were recorded on (Tests/Tests/*.bmp).  Run ONCE in the build container (the reference tree does not
exist on the GPU box):

    python tests/golden/make_mission_fixture.py

The npz holds DATA only: the grey-scale pixels of the two bitmaps, their start / goal (from the file names) and, per
step of each recorded mission, what the reference's planner process printed -- position, patch rectangle, "nodes
updated", "nodes expanded", path cost and distance, the numbers as the strings the log holds (six significant digits) --
and what the simulator printed for the same step (position with six decimals, patch rectangle).
The logs are console output of the reference's Field D* planner driven by Simulator/simulator/run_simulator.py (an
older revision: it prints lines the current sources have commented out, FieldDPlanner_impl.h:65,139)."""
import json
import os
import re

import numpy as np
from PIL import Image

REF = "/root/reference/Tests"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_missions.npz")
MISSIONS = {"noise-trap": "noise-trap_90_90_25_25_.bmp", "wall-b": "wall-b_27_10_2_10_.bmp"}

out = {}
for name, bmp in MISSIONS.items():
    fx, fy, tx, ty = (float(v) for v in bmp[:-5].split("_")[1:5])
    out[name + "_pixels"] = np.array(Image.open(os.path.join(REF, "Tests", bmp)).convert("L"), dtype=np.uint8)
    out[name + "_startgoal"] = np.array([fx, fy, tx, ty], np.float32)
    steps, cur = [], None
    sim_pos, sim_patch = [], []
    for ln in open(os.path.join(REF, "Results", name, "planner_opt0.log")).read().splitlines():
        m = re.match(r"\[SIMULATOR\] New position: \[([-0-9.]+), ([-0-9.]+)\]", ln)       # the simulator's own print of the same step: %f
        if m:
            sim_pos.append([m.group(1), m.group(2)])
            continue
        m = re.match(r"\[SIMULATOR\] New patch: position \[(\d+), (\d+)\], shape \[(\d+), (\d+)\]", ln)
        if m:
            sim_patch.append([int(v) for v in m.groups()])
            continue
        m = re.match(r"\[PLANNER\]\s+New position: \[([-0-9.e+]+), ([-0-9.e+]+)\]", ln)
        if m:
            cur = {"pos": [m.group(1), m.group(2)]}
            steps.append(cur)
            continue
        m = re.match(r"\[PLANNER\]\s+New patch: position \[(\d+), (\d+)\], shape \[(\d+), (\d+)\]", ln)
        if m:
            cur["patch"] = [int(v) for v in m.groups()]      # top, left, width, height
            continue
        m = re.match(r"(\d+) nodes (updated|expanded)", ln)
        if m:
            cur[m.group(2)] = int(m.group(1))
            continue
        m = re.match(r"Found path. Cost: ([-0-9.e+]+) Distance: ([-0-9.e+]+)", ln)
        if m:
            cur["cost"], cur["dist"] = m.group(1), m.group(2)
    assert len(sim_pos) == len(sim_patch) == len(steps), (len(sim_pos), len(sim_patch), len(steps))
    for st, a, b in zip(steps, sim_pos, sim_patch):
        st["sim_pos"], st["sim_patch"] = a, b
        assert "patch" not in st or st["patch"] == b, (st, b)          # (one planner line per log is torn by the two processes' interleaved output)
    out[name + "_steps"] = np.frombuffer(json.dumps(steps).encode(), dtype=np.uint8)
    print(name, out[name + "_pixels"].shape, len(steps), "steps; first", steps[0], "last", steps[-1])
np.savez_compressed(OUT, **out)
print("wrote", OUT, os.path.getsize(OUT), "bytes")
