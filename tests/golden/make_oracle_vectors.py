"""Generates tests/golden/oracle_vectors.npz: outputs of the CPU oracle (oracle/) on small seeded
cases, committed as data.  They do two jobs: the GPU tests compare the engine with them without
trusting the oracle library present at test time, and a CPU test holds the oracle itself to them
(a change of the oracle's results shows up as a diff of this file).

    python tests/golden/make_oracle_vectors.py

Per case (planner variant x map): the map, start / goal, the patch script; after the first plan and
after each replan the consistent-and-below-the-start-key mask, the G values on it, num_nodes_updated,
and the extracted path (max_steps 30)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle_py as orc   # noqa: E402
import ufm_amd            # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_vectors.npz")
CASES = [("FD", 0, 0), ("FD", 1, 0), ("FD", 1, 1), ("SG", 0, 0), ("SG", 1, 0), ("SG", 2, 1), ("DFM", 0, 0), ("DFM", 1, 0)]
ALGOS = {"FD": orc.ALGO_FD, "SG": orc.ALGO_SG, "DFM": orc.ALGO_DFM}


def run_case(algo, lvl, heur, seed=5, width=56, length=40, n_patches=2):
    cost = ufm_amd.synth.cost_map(seed, width, length)
    start, goal = ufm_amd.synth.start_goal(width, length)
    script = list(ufm_amd.synth.replan_script(seed, width, length, n_patches=n_patches, size=15))
    o = orc.OraclePlanner(ALGOS[algo], lvl, bool(heur))
    o.reset(); o.set_occupancy_threshold(1); o.set_heuristic_multiplier(float(cost.min()))
    o.set_map(cost); o.set_start(*start); o.set_goal(*goal)
    out = {"cost": cost, "start": np.array(start, np.float32), "goal": np.array(goal, np.float32),
           "patch_pos": np.array([(t, l) for _, _, t, l, _ in script], np.int32),
           "patch_start": np.array([s for _, s, _, _, _ in script], np.float32),
           "patches": np.stack([p for *_, p in script])}
    steps = [None] + script
    for i, st in enumerate(steps):
        if st is not None:
            _, s, top, left, patch = st
            o.patch_map(patch, top, left); o.set_start(*s)
        assert o.step() == 0
        m = o.trusted_mask(below_start_key=True)
        pts, costs, tc, td = o.extract_path(max_steps=30, allow_indirect=(algo != "SG"))
        out["mask%d" % i] = np.packbits(m)
        out["g%d" % i] = o.g()[m]
        out["updated%d" % i] = np.array([o.num_updated], np.int64)
        out["path%d" % i] = pts
        out["pathcost%d" % i] = np.array([tc, td], np.float32)
    return out


if __name__ == "__main__":
    blob = {}
    for algo, lvl, heur in CASES:
        for k, v in run_case(algo, lvl, heur).items():
            blob["%s%d%s_%s" % (algo, lvl, "h" if heur else "", k)] = v
    np.savez_compressed(OUT, **blob)
    print("wrote", OUT, os.path.getsize(OUT), "bytes,", len(blob), "arrays")
