"""GPU box: throughput of a batch of independent maps on one GPU (BASELINE config 4 shape:
8 x 2048^2 MS-DFM maps per GPU).  Full plan of all maps in one set of launches."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ufm_amd
ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=2048)
ap.add_argument("--maps", default="1,2,4,8")
ap.add_argument("--algo", default="DFM")
ap.add_argument("--param", action="append", default=[], help="name=value for ufm_batch_set_param")
a = ap.parse_args()
algo = {"FD": 0, "SG": 1, "DFM": 2}[a.algo]
size = a.size
costs = [ufm_amd.synth.cost_map(1000 + m, size, size) for m in range(max(int(v) for v in a.maps.split(",")))]
start, goal = ufm_amd.synth.start_goal(size, size)
for n in [int(v) for v in a.maps.split(",")]:
    b = ufm_amd.BatchPlanner(n, algo, 1 if algo != 1 else 2)
    b.set_occupancy_threshold(1)
    for kv in a.param:
        b.set_param(kv.split("=")[0], float(kv.split("=")[1]))
    for i in range(n):
        b.set_map(i, costs[i]); b.set_start(i, *start); b.set_goal(i, *goal)
    best = None
    for rep in range(3):
        for i in range(n):
            b.reset(i)
        t0 = time.perf_counter()
        assert b.step() == 0
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    st = b.stats
    print("%s %d x %d^2: plan %.2f ms  cells %d  -> %.1f M cells/s  launches %d visits %d (%.0f per launch)" % (
        a.algo, n, size, best * 1e3, st.expanded, st.expanded / best / 1e6, st.launches, st.tile_visits, st.tile_visits / max(1, st.launches)), flush=True)
    b.close()
